"""Multi-GPU: worlds are independent, so they shard as contiguous ranges, one process per GPU.

There is no cross-world term anywhere in env.step (each reference env instance is its own OS
process, /root/reference/ga3c/GA3C/ProcessAgent.py:221), so the data path needs NO collective:
rank r owns worlds [offset_r, offset_r + count_r), its scenario RNG is keyed on GLOBAL world ids
(`world_offset`), and the sharded run reproduces the unsharded one bit for bit.

The one real exchange the north-star names is returning per-world (obs, reward, done) to a
trainer: `gather_step_outputs` packs them into one contiguous float32 buffer per rank and issues
ONE `all_gather_into_tensor` (RCCL over xGMI with backend "nccl"; gloo on CPU for tests).  For the
full GA3C loop keep a policy replica per GPU and skip this gather.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(total_worlds: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous partition; the first `total % size` ranks hold one extra world.  -> (offset, count)"""
    if not (0 <= rank < world_size) or total_worlds < 0:
        raise ValueError("bad shard request")
    base, extra = divmod(total_worlds, world_size)
    count = base + (1 if rank < extra else 0)
    offset = rank * base + min(rank, extra)
    return offset, count


def pack_step_outputs(obs: torch.Tensor, rewards: torch.Tensor, done: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[Wl,N,width] f32, [Wl,N] f32, [Wl,N] u8 -> one contiguous f32 [Wl, N, width+2] (obs | reward | done)."""
    Wl, N, width = obs.shape
    if out is None:
        out = torch.empty((Wl, N, width + 2), dtype=torch.float32, device=obs.device)
    out[..., :width] = obs
    out[..., width] = rewards
    out[..., width + 1] = done.to(torch.float32)
    return out


def unpack_step_outputs(packed: torch.Tensor):
    width = packed.shape[-1] - 2
    return packed[..., :width], packed[..., width], packed[..., width + 1].to(torch.uint8)


def gather_step_outputs(packed: torch.Tensor, total_worlds: int, group=None) -> torch.Tensor:
    """One all-gather of every rank's packed shard -> [total_worlds, N, width+2] in global world order.
    Shards of unequal size are padded to the largest one for the collective and trimmed after."""
    size = dist.get_world_size(group)
    counts = [shard_range(total_worlds, r, size)[1] for r in range(size)]
    biggest = max(counts)
    Wl = packed.shape[0]
    if Wl != counts[dist.get_rank(group)]:
        raise ValueError("this rank holds %d worlds, expected %d" % (Wl, counts[dist.get_rank(group)]))
    send = packed
    if Wl != biggest:
        send = torch.zeros((biggest,) + tuple(packed.shape[1:]), dtype=packed.dtype, device=packed.device)
        send[:Wl] = packed
    recv = torch.empty((size * biggest,) + tuple(packed.shape[1:]), dtype=packed.dtype, device=packed.device)
    dist.all_gather_into_tensor(recv, send.contiguous(), group=group)
    if all(c == biggest for c in counts):
        return recv
    recv = recv.view((size, biggest) + tuple(packed.shape[1:]))
    return torch.cat([recv[r, :counts[r]] for r in range(size)], dim=0)


class ShardedEnv(object):
    """This rank's shard of a `total_worlds`-world env (one process per GPU)."""

    def __init__(self, total_worlds: int, config=None, device=None, seed: int = 0, group=None, **cfg_overrides):
        from .batched_env import BatchedCollisionAvoidanceEnv
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.total_worlds = int(total_worlds)
        self.offset, self.count = shard_range(self.total_worlds, self.rank, self.size)
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.env = BatchedCollisionAvoidanceEnv(self.count, config, device=device, world_offset=self.offset, seed=seed,
                                                **cfg_overrides)
        self._packed = None

    def __getattr__(self, name):
        return getattr(self.env, name)

    def gather(self) -> torch.Tensor:
        """(obs | reward | done) of ALL worlds on every rank: [total_worlds, N, width+2] float32."""
        e = self.env
        self._packed = pack_step_outputs(e.obs, e.rewards, e.done, self._packed)
        if self.size == 1:
            return self._packed
        return gather_step_outputs(self._packed, self.total_worlds, self.group)
