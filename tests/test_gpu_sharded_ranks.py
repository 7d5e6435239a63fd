"""The N > 1 hand-over executed with 4 and 8 RANKS on one device (gloo process group, every rank on cuda:0 -- no multi-GPU node is
available to the builder): each rank steps ITS shard of a `total`-world env on the GPU through the C ABI (`ShardedEnv`, world
offsets, RNG keyed on global world ids), the packed records travel through the process group in the wire layout of the native
path (rank-major blocks of K steps; transport "torch" = `sharding.gather_blocks`), and every receiving rank must hold, bit for
bit, what ONE unsharded env of `total` worlds produces.  Even and ragged shards, to every rank and to one trainer rank, one step
and K steps per launch.  (The RCCL transport of the same layout: tests/test_gpu_packed.py, single rank; first multi-device run:
the driver's.)"""
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

N, SEED = 4, 13


def _cfg():
    from rl_collision_avoidance_amd.config import EnvConfig
    return EnvConfig()


def _actions(total, K, launches):
    g = torch.Generator().manual_seed(77)
    return torch.randint(0, 11, (launches, K, total, N), generator=g, dtype=torch.int32)


def _worker(rank, size, cases, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=size)
    try:
        from rl_collision_avoidance_amd.sharding import ShardedEnv
        torch.cuda.set_device(0)
        out = {}
        for total, K, root, launches in cases:
            sh = ShardedEnv(total, _cfg(), device="cuda:0", seed=SEED, gen_min_agents=2, gen_nonlearning_fraction=0.2, gen_pool_size=512)
            assert sh.transport == "torch" and (sh.offset, sh.count) == __import__("rl_collision_avoidance_amd.sharding", fromlist=["x"]).shard_range(total, rank, size)
            sh.reset()
            acts = _actions(total, K, launches)[:, :, sh.offset:sh.offset + sh.count].cuda()
            got = []
            for l in range(launches):
                slot = sh.step_and_gather(acts[l] if K > 1 else acts[l, 0], root=root)
                g = sh.gathered(slot)
                if root >= 0 and rank != root:
                    assert g is None
                else:
                    got.append(g.cpu().numpy().reshape(K, total, N, -1).copy())
            assert ("all_gather" in sh.gather_form) if root < 0 else ("gather to rank %d" % root in sh.gather_form)
            out[(total, K, root)] = np.stack(got) if got else None
            sh.close()
        ret[rank] = out
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("size", [4, 8])
def test_sharded_ranks_on_one_device_equal_the_unsharded_env(size):
    from rl_collision_avoidance_amd.batched_env import BatchedCollisionAvoidanceEnv
    # (total worlds, steps per launch, receiving rank or -1, launches)
    cases = [(64 * size, 1, -1, 12), (64 * size + 3, 8, -1, 6), (64 * size, 8, 1, 6), (40 * size + 5, 1, size - 1, 10)]
    port = 29300 + (os.getpid() % 600) + size
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(size, cases, port, ret), nprocs=size, join=True)
    for total, K, root, launches in cases:
        env = BatchedCollisionAvoidanceEnv(total, _cfg(), device="cuda:0", seed=SEED, gen_min_agents=2, gen_nonlearning_fraction=0.2,
                                           gen_pool_size=512)
        env.reset()
        acts = _actions(total, K, launches).cuda()
        slots = env.new_step_slots(K, packed=True)
        want = []
        for l in range(launches):
            if K > 1:
                env.step_autoreset_packed(acts[l], slots)
                want.append(slots.packed.cpu().numpy().copy())
            else:
                pk = env.new_packed()
                env.step_autoreset_packed(acts[l, 0], pk)
                want.append(pk.cpu().numpy()[None].copy())
        want = np.stack(want)
        assert env.episode.max().item() >= 0
        env.close()
        for rank in range(size):
            got = ret[rank][(total, K, root)]
            if root >= 0 and rank != root:
                assert got is None
            else:
                assert got.shape == want.shape and np.array_equal(got, want), (total, K, root, rank)
