"""Multi-GPU: worlds are independent, so they shard as contiguous ranges, one process per GPU.

There is no cross-world term anywhere in env.step (each reference env instance is its own OS
process, /root/reference/ga3c/GA3C/ProcessAgent.py:221), so the data path needs NO collective:
rank r owns worlds [offset_r, offset_r + count_r), its scenario RNG is keyed on GLOBAL world ids
(`world_offset`), and the sharded run reproduces the unsharded one bit for bit.

The one real exchange the north-star names is returning per-world (obs, reward, done) to a
trainer.  The env kernel writes the packed per-agent record (obs | reward | done) itself
(`cavoid_step*_packed`), and `NativeGather` issues ONE `ncclAllGather` per step through the C ABI
(`cavoid_gather_begin/wait`, RCCL over xGMI) on the communicator's own stream, double-buffered so
that gather(t) overlaps step(t+1) -- `ShardedEnv.step_and_gather`.  `gather_step_outputs` is the
same exchange through `torch.distributed` (`all_gather_into_tensor`; gloo on CPU for the tests,
ragged shards padded).  For the full GA3C loop keep a policy replica per GPU and skip this gather
(SURVEY.md section 8e).
"""
from __future__ import annotations

import ctypes as C
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


def gather_blocks(send: torch.Tensor, world_counts, root: int = -1, group=None):
    """The hand-over of `cavoid_gatherv_begin` through `torch.distributed` (any backend; gloo on CPU for the tests and for dry
    runs of the N > 1 path on one device): rank r contributes ``send`` [K, world_counts[r], N, width+2] (K steps of its shard's packed
    records); the receivers -- every rank, or only ``root`` -- get the list of all ranks' blocks in rank order (the wire layout of the
    native path: rank-major blocks of K steps each), the others None.  Ragged shards are padded to the largest one for the
    collective and trimmed after."""
    size, rank = dist.get_world_size(group), dist.get_rank(group)
    counts = [int(c) for c in world_counts]
    if len(counts) != size or send.dim() != 4 or send.shape[1] != counts[rank]:
        raise ValueError("send must be [K, %d worlds of this rank, N, width+2]" % (counts[rank] if len(counts) == size else -1))
    K, _, N, rec = send.shape
    biggest = max(counts)
    pad = send
    if counts[rank] != biggest:
        pad = torch.zeros((K, biggest, N, rec), dtype=send.dtype, device=send.device)
        pad[:, :counts[rank]] = send
    pad = pad.contiguous()
    if root < 0:
        recv = torch.empty((size,) + tuple(pad.shape), dtype=send.dtype, device=send.device)
        dist.all_gather_into_tensor(recv.view(size * K, biggest, N, rec), pad, group=group)
        return [recv[r, :, :counts[r]] for r in range(size)]
    dst = dist.get_global_rank(group, root) if group is not None else root
    if rank == root:
        parts = [torch.empty_like(pad) for _ in range(size)]
        dist.gather(pad, parts, dst=dst, group=group)
        return [parts[r][:, :counts[r]] for r in range(size)]
    dist.gather(pad, None, dst=dst, group=group)
    return None


class NativeGather(object):
    """The all-gather behind the C ABI: one RCCL communicator per process (one process per GPU), created from a
    128-byte id that rank 0 makes and the process group hands round (any backend: only 128 bytes travel that way)."""

    SLOTS = 2

    def __init__(self, device, group=None, force_rccl: Optional[bool] = None):
        """force_rccl: a ONE-rank communicator goes through RCCL too (`cavoid_comm_create_ex(..., CAVOID_COMM_FORCE_RCCL)`:
        ncclCommInitRank, ncclAllGather, grouped self send / recv) instead of the device copy; None = the environment variable
        CAVOID_COMM_FORCE_RCCL decides.  Multi-rank communicators always are RCCL communicators."""
        import os
        from . import _lib
        self._libmod = _lib
        self._lib = _lib.lib()
        self.device = torch.device(device)
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.size = dist.get_world_size(group) if dist.is_initialized() else 1
        if force_rccl is None:
            force_rccl = os.environ.get("CAVOID_COMM_FORCE_RCCL", "0") not in ("", "0")
        ident = torch.zeros(128, dtype=torch.uint8)
        if self.size > 1 or force_rccl:
            if self.rank == 0:
                buf = (C.c_ubyte * 128)()
                _lib.check(self._lib.cavoid_comm_unique_id(buf), "cavoid_comm_unique_id")
                ident = torch.tensor(list(buf), dtype=torch.uint8)
            if self.size > 1:
                backend = dist.get_backend(group)
                carrier = ident.to(self.device) if backend == "nccl" else ident
                dist.broadcast(carrier, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
                ident = carrier.cpu()
        raw = (C.c_ubyte * 128)(*ident.tolist())
        handle = C.c_void_p()
        _lib.check(self._lib.cavoid_comm_create_ex(raw, self.size, self.rank, self.device.index or 0,
                                                   _lib.COMM_FORCE_RCCL if force_rccl else 0, C.byref(handle)), "cavoid_comm_create_ex")
        self._h = handle
        vals = [C.c_int32() for _ in range(4)]
        _lib.check(self._lib.cavoid_comm_info(handle, *[C.byref(v) for v in vals]), "cavoid_comm_info")
        self.uses_rccl, self.rccl_version = bool(vals[2].value), int(vals[3].value)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def begin(self, slot: int, send: torch.Tensor, recv: torch.Tensor) -> None:
        """Enqueue gather `slot` of `send` (this rank's packed shard) into `recv` [size * send.numel()] behind the work
        already on the current stream; returns at once."""
        if send.dtype != torch.float32 or recv.dtype != torch.float32 or not send.is_contiguous() or not recv.is_contiguous():
            raise ValueError("send / recv must be contiguous float32 tensors")
        if recv.numel() != self.size * send.numel():
            raise ValueError("recv must hold %d x send" % self.size)
        self._libmod.check(self._lib.cavoid_gather_begin(self._h, slot, C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()),
                                                         send.numel(), self._stream()), "cavoid_gather_begin")

    def begin_v(self, slot: int, send: torch.Tensor, recv: Optional[torch.Tensor], counts, root: int = -1) -> None:
        """Ragged / rooted form: ``counts[r]`` floats from rank r (the same list on every rank), shards laid out in rank
        order in ``recv``; ``root >= 0``: only that rank receives (``recv`` may be None on the others)."""
        if send.dtype != torch.float32 or not send.is_contiguous() or len(counts) != self.size or send.numel() != counts[self.rank]:
            raise ValueError("send must be this rank's contiguous float32 shard of counts[rank] floats")
        receiver = root < 0 or root == self.rank
        if receiver and (recv is None or recv.dtype != torch.float32 or not recv.is_contiguous() or recv.numel() != sum(counts)):
            raise ValueError("recv must be a contiguous float32 tensor of sum(counts) floats")
        arr = (C.c_int64 * self.size)(*[int(c) for c in counts])
        self._libmod.check(self._lib.cavoid_gatherv_begin(self._h, slot, C.c_void_p(send.data_ptr()),
                                                          C.c_void_p(recv.data_ptr()) if receiver else None, arr, int(root),
                                                          self._stream()), "cavoid_gatherv_begin")

    def wait(self, slot: int) -> None:
        """Make the current stream wait for the last gather begun in `slot` (no host synchronisation)."""
        self._libmod.check(self._lib.cavoid_gather_wait(self._h, slot, self._stream()), "cavoid_gather_wait")

    def close(self) -> None:
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.cavoid_comm_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardedEnv(object):
    """This rank's shard of a `total_worlds`-world env (one process per GPU)."""

    def __init__(self, total_worlds: int, config=None, device=None, seed: int = 0, group=None, transport: Optional[str] = None,
                 force_rccl: Optional[bool] = None, **cfg_overrides):
        """transport: "native" = `cavoid_gather*` (RCCL behind the C ABI, own stream, overlapped) -- the default whenever the process
        group runs on nccl or there is one rank; "torch" = the same hand-over through `torch.distributed` (`gather_blocks`:
        synchronous; gloo dry runs of the N > 1 path on one device, and the only form a gloo group can carry)."""
        from .batched_env import BatchedCollisionAvoidanceEnv
        self.group = group
        if transport is None:
            transport = "torch" if (dist.is_initialized() and dist.get_world_size(group) > 1 and dist.get_backend(group) != "nccl") else "native"
        if transport not in ("native", "torch"):
            raise ValueError("transport must be 'native' or 'torch'")
        self.transport = transport
        self.force_rccl = force_rccl
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.total_worlds = int(total_worlds)
        self.offset, self.count = shard_range(self.total_worlds, self.rank, self.size)
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.env = BatchedCollisionAvoidanceEnv(self.count, config, device=device, world_offset=self.offset, seed=seed,
                                                **cfg_overrides)
        self._packed = None
        self._native = None
        self._send = self._recv = None
        self._t = 0

    def __getattr__(self, name):
        return getattr(self.env, name)

    def gather(self) -> torch.Tensor:
        """(obs | reward | done) of ALL worlds on every rank: [total_worlds, N, width+2] float32, from the env's plain
        outputs through torch.distributed (works for ragged shards and on any backend)."""
        e = self.env
        self._packed = pack_step_outputs(e.obs, e.rewards, e.done, self._packed)
        if self.size == 1:
            return self._packed
        return gather_step_outputs(self._packed, self.total_worlds, self.group)

    # -- the native path: packed records straight from the kernel, ncclAllGather behind the C ABI, overlapped ---------
    def _native_setup(self, steps: int = 1, root: int = -1):
        """Buffers of the native hand-over: per slot a send buffer of `steps` packed records of this shard and, on the
        receiving ranks, a recv buffer of rank-major blocks [rank][steps, count_r, N, width+2].  Equal shards to every rank go
        through ONE ncclAllGather; ragged shards or a single receiving rank (`root`) through the point-to-point gather
        (`cavoid_gatherv_begin`).  Called again with another (steps, root) it keeps the communicator and re-makes the buffers."""
        from .batched_env import StepSlots
        e = self.env
        if self._native is None:
            self._native = NativeGather(e.device, self.group, self.force_rccl) if self.transport == "native" else False
        self._steps, self._root = int(steps), int(root)
        self._counts = [shard_range(self.total_worlds, r, self.size)[1] for r in range(self.size)]
        self._even = all(c == self._counts[0] for c in self._counts)
        self._send = [StepSlots(e, steps, packed=True) for _ in range(NativeGather.SLOTS)]
        receiver = root < 0 or root == self.rank
        rec = e.max_agents * e.packed_width
        # wire layout: rank-major blocks, rank r's block = [steps, count_r, N, width+2]
        self._recv = [torch.zeros((steps * self.total_worlds * rec,), dtype=torch.float32, device=e.device) if (receiver and self._native)
                      else None for _ in range(NativeGather.SLOTS)]
        self._blocks = [None] * NativeGather.SLOTS          # transport "torch": the received blocks of each slot
        self._floats = [steps * c * rec for c in self._counts]

    def set_gather(self, steps: int = 1, root: int = -1) -> None:
        """Choose (or change) the hand-over form: `steps` env steps per launch-and-gather block, to every rank (root < 0) or to
        one.  Every gather in flight is waited for; the communicator is kept (one per process), only the buffers are re-made."""
        if self._native:
            torch.cuda.synchronize(self.env.device)
        self._native_setup(steps, root)
        self._t = 0

    @property
    def gather_form(self) -> str:
        """Which exchange `step_and_gather` issues (for logs and the bench line): decided by the shard sizes and the receiver set,
        not by the number of steps per launch."""
        if self._native is None:
            return "not set up"
        if self.transport == "torch":
            return "torch.distributed %s (%s)" % ("all_gather_into_tensor" if self._root < 0 else "gather to rank %d" % self._root,
                                                  dist.get_backend(self.group) if dist.is_initialized() else "single rank")
        if not self._native.uses_rccl:
            return "device copy on the communicator's stream (one rank; %s)" % ("cavoid_gather_begin" if self._even and self._root < 0 else "cavoid_gatherv_begin")
        if self._even and self._root < 0:
            return "ncclAllGather (cavoid_gather_begin: equal shards, every rank receives; blocks of %d step(s))" % self._steps
        return "point-to-point RCCL group (cavoid_gatherv_begin: %s shards, %s)" % (
            "equal" if self._even else "ragged", "every rank receives" if self._root < 0 else "rank %d receives" % self._root)

    def step_and_gather(self, actions: torch.Tensor, root: int = -1) -> int:
        """One auto-reset step (actions [Wl,N]) -- or K steps in ONE launch (actions [K,Wl,N], every step's records in its own
        slot) -- of this shard into a packed buffer + the gather of it, begun but not waited for: the NEXT call's launch
        runs while this gather is on the wire.  `root >= 0`: only that rank receives (the trainer rank).  Returns the slot;
        `gathered(slot)` makes the current stream wait for it and returns the records of ALL worlds."""
        steps = 1 if actions.dim() == 2 else int(actions.shape[0])
        if self._native is None or self._send is None:
            self._native_setup(steps, root)
        if steps != self._steps or root != self._root:
            raise ValueError("step_and_gather was set up for %d step(s) per launch, root %d" % (self._steps, self._root))
        slot = self._t % NativeGather.SLOTS
        self._t += 1
        if self._native:
            self._native.wait(slot)                 # gather(t-2) is done with send[slot] / recv[slot]
        sl = self._send[slot]
        if steps == 1:
            self.env.step_autoreset_packed(actions, sl.packed[0])
        else:
            self.env.step_autoreset_packed(actions, sl)
        if not self._native:                        # transport "torch": the same blocks through the process group, synchronously
            send = sl.packed if dist.get_backend(self.group) == "nccl" else sl.packed.cpu()
            blocks = gather_blocks(send, self._counts, root, self.group)
            self._blocks[slot] = None if blocks is None else [b.to(self.env.device) for b in blocks]
            return slot
        if self._even and root < 0:                 # equal shards to every rank: ONE ncclAllGather (rank-major blocks of K steps)
            self._native.begin(slot, sl.packed, self._recv[slot])
        else:
            self._native.begin_v(slot, sl.packed, self._recv[slot], self._floats, root)
        return slot

    def gathered_blocks(self, slot: int) -> Optional[list]:
        """The gather as it arrives: one view [K, worlds of rank r, N, width+2] per rank, in rank order (no copy); None on a rank
        that does not receive.  Valid until the slot's next use."""
        if not self._native:
            return self._blocks[slot]
        self._native.wait(slot)
        recv = self._recv[slot]
        if recv is None:
            return None
        e, K = self.env, self._steps
        blocks, off = [], 0
        for c, f in zip(self._counts, self._floats):
            blocks.append(recv[off:off + f].view(K, c, e.max_agents, e.packed_width))
            off += f
        return blocks

    def gathered(self, slot: int) -> Optional[torch.Tensor]:
        """[total_worlds, N, width+2] (one step per launch) or [K, total_worlds, N, width+2]; None on a rank that does not
        receive.  Valid until the slot's next use."""
        blocks = self.gathered_blocks(slot)
        if blocks is None:
            return None
        e, K = self.env, self._steps
        if K == 1 and self._native:
            return self._recv[slot].view(self.total_worlds, e.max_agents, e.packed_width)
        if K == 1:
            return torch.cat(blocks, dim=1)[0]
        # rank-major blocks -> [K, total_worlds, ...] (a view when there is one rank, else one gather-side copy per rank block)
        return blocks[0] if len(blocks) == 1 else torch.cat(blocks, dim=1)

    def close(self) -> None:
        if self._native:
            torch.cuda.synchronize(self.env.device)
            self._native.close()
        self._native = None
        self.env.close()
