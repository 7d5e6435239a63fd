"""Multi-GPU path on CPU: 2 processes over gloo.  Each rank steps ITS shard with the CPU oracle (test
infrastructure standing in for the GPU env), packs (obs|reward|done), all-gathers, and the result
must equal the unsharded run bit for bit -- partition offsets, global-world-id RNG keying and the
gather layout are what is under test."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from rl_collision_avoidance_amd import sharding


def test_shard_range_partitions():
    for total in (0, 1, 7, 8192, 65536, 65537):
        for size in (1, 2, 3, 8):
            spans = [sharding.shard_range(total, r, size) for r in range(size)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == total
            for (o0, c0), (o1, _) in zip(spans[:-1], spans[1:]):
                assert o0 + c0 == o1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def test_pack_unpack_roundtrip():
    obs = torch.randn(5, 4, 27)
    rew = torch.randn(5, 4)
    done = (torch.rand(5, 4) > 0.5).to(torch.uint8)
    o, r, d = sharding.unpack_step_outputs(sharding.pack_step_outputs(obs, rew, done))
    assert torch.equal(o, obs) and torch.equal(r, rew) and torch.equal(d, done)


def _oracle_run(total, offset, count, N, seed, steps):
    from oracle import c_oracle as co
    cfg, gen = co.default_cfg(N), co.default_gen(2, N, 0.2, pool_size=64)
    st = co.State.empty(count, N)
    ep = np.zeros(count, np.uint32)
    co.generate(cfg, gen, seed, st, ep, world_offset=offset)
    rng = np.random.default_rng(seed)
    all_actions = rng.integers(0, 11, size=(steps, total, N)).astype(np.int32)
    outs = []
    for t in range(steps):
        obs, rew, done, go = co.step_autoreset(cfg, gen, seed, st, ep, all_actions[t, offset:offset + count], world_offset=offset)
        outs.append((obs.astype(np.float32), rew.astype(np.float32), done))
    return outs


def _worker(rank, size, total, N, seed, steps, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=size)
    try:
        offset, count = sharding.shard_range(total, rank, size)
        outs = _oracle_run(total, offset, count, N, seed, steps)
        gathered = []
        for obs, rew, done in outs:
            packed = sharding.pack_step_outputs(torch.from_numpy(obs), torch.from_numpy(rew), torch.from_numpy(done))
            gathered.append(sharding.gather_step_outputs(packed, total).numpy().copy())
        ret[rank] = np.stack(gathered)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [64, 37])        # even split and a ragged one (padding path)
def test_two_rank_gloo_gather_equals_unsharded(total):
    N, seed, steps, size = 4, 5, 25, 2
    port = 29500 + (os.getpid() % 1000) + total
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(size, total, N, seed, steps, port, ret), nprocs=size, join=True)
    full = _oracle_run(total, 0, total, N, seed, steps)
    want = np.stack([sharding.pack_step_outputs(torch.from_numpy(o), torch.from_numpy(r), torch.from_numpy(d)).numpy()
                     for o, r, d in full])
    for rank in range(size):
        assert ret[rank].shape == want.shape
        assert np.array_equal(ret[rank], want), rank       # every rank holds the global result, bitwise


def _dp_worker(rank, size, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=size)
    try:
        from rl_collision_avoidance_amd.config import EnvConfig
        from rl_collision_avoidance_amd.ga3c.network import A3CTrainer, NetworkVP_rnn
        torch.manual_seed(0)
        net = NetworkVP_rnn(EnvConfig(), seed=4)
        tr = A3CTrainer(net, learning_rate=1e-3)
        assert tr.distributed
        g = torch.Generator().manual_seed(100)
        x = torch.randn(64, 26, generator=g); x[:, 0] = torch.randint(0, 4, (64,), generator=g).float()
        y = torch.randn(64, generator=g)
        a = torch.nn.functional.one_hot(torch.randint(0, 11, (64,), generator=g), 11).float()
        half = slice(rank * 32, (rank + 1) * 32)
        for _ in range(5):
            tr.train(x[half], y[half], a[half])
        ret[rank] = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).numpy()
    finally:
        dist.destroy_process_group()


def test_two_rank_data_parallel_trainer_matches_single_trainer():
    """Replica per rank + one flat gradient all-reduce == a single trainer on the concatenated batch."""
    from rl_collision_avoidance_amd.config import EnvConfig
    from rl_collision_avoidance_amd.ga3c.network import A3CTrainer, NetworkVP_rnn
    port = 29700 + (os.getpid() % 1000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_dp_worker, args=(2, port, ret), nprocs=2, join=True)
    assert np.array_equal(ret[0], ret[1])                       # replicas stay in lock step
    net = NetworkVP_rnn(EnvConfig(), seed=4)
    tr = A3CTrainer(net, learning_rate=1e-3, distributed=False)
    g = torch.Generator().manual_seed(100)
    x = torch.randn(64, 26, generator=g); x[:, 0] = torch.randint(0, 4, (64,), generator=g).float()
    y = torch.randn(64, generator=g)
    a = torch.nn.functional.one_hot(torch.randint(0, 11, (64,), generator=g), 11).float()
    for _ in range(5):
        tr.train(x, y, a)
    single = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).numpy()
    np.testing.assert_allclose(ret[0], single, rtol=0, atol=2e-5)


# ---- 4 and 8 ranks: every form of the hand-over (even / ragged shards, to every rank / to one trainer rank, one step or a
# ---- block of K steps per exchange), in the wire layout of the native path (rank-major blocks) --------------------------------
def _blocks_worker(rank, size, cases, N, seed, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=size)
    try:
        out = {}
        for total, K, root in cases:
            counts = [sharding.shard_range(total, r, size)[1] for r in range(size)]
            offset, count = sharding.shard_range(total, rank, size)
            outs = _oracle_run(total, offset, count, N, seed, K)
            send = torch.stack([sharding.pack_step_outputs(torch.from_numpy(o), torch.from_numpy(r), torch.from_numpy(d)) for o, r, d in outs])
            blocks = sharding.gather_blocks(send, counts, root)
            if root >= 0 and rank != root:
                assert blocks is None
                out[(total, K, root)] = None
            else:
                assert len(blocks) == size and all(b.shape[:2] == (K, c) for b, c in zip(blocks, counts))
                out[(total, K, root)] = torch.cat(blocks, dim=1).numpy().copy()           # [K, total, N, width+2] in global world order
            # the one-step all-gather of round 1 is the K = 1, root < 0 case of the same exchange
            if K == 1 and root < 0:
                assert np.array_equal(sharding.gather_step_outputs(send[0], total).numpy(), out[(total, K, root)][0])
        ret[rank] = out
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("size", [4, 8])
def test_four_and_eight_rank_gathers_equal_unsharded(size):
    N, seed = 4, 9
    # (total worlds, steps per exchange, receiving rank or -1): even and ragged shards, a rank with ZERO worlds (total < size)
    cases = [(8 * size, 1, -1), (8 * size + 3, 1, -1), (8 * size, 3, -1), (8 * size + 5, 4, -1), (8 * size, 1, 0), (8 * size + 1, 2, size - 1),
             (size - 1, 2, -1), (size - 1, 1, 1)]
    port = 29100 + (os.getpid() % 700) + size
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_blocks_worker, args=(size, cases, N, seed, port, ret), nprocs=size, join=True)
    for total, K, root in cases:
        full = _oracle_run(total, 0, total, N, seed, K)
        want = np.stack([sharding.pack_step_outputs(torch.from_numpy(o), torch.from_numpy(r), torch.from_numpy(d)).numpy() for o, r, d in full])
        for rank in range(size):
            got = ret[rank][(total, K, root)]
            if root >= 0 and rank != root:
                assert got is None
            else:
                assert got.shape == want.shape and np.array_equal(got, want), (total, K, root, rank)


def _dp_ragged_worker(rank, size, rows, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=size)
    try:
        from rl_collision_avoidance_amd.config import EnvConfig
        from rl_collision_avoidance_amd.ga3c.network import A3CTrainer, NetworkVP_rnn
        net = NetworkVP_rnn(EnvConfig(), seed=4)
        tr = A3CTrainer(net, learning_rate=1e-3)
        x, y, a = _dp_batch(sum(rows))
        lo = sum(rows[:rank])
        mine = slice(lo, lo + rows[rank])
        for _ in range(4):
            tr.train(x[mine], y[mine], a[mine])                 # (a rank with NO rows still joins the all-reduce and takes the step)
        ret[rank] = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).numpy()
    finally:
        dist.destroy_process_group()


def _dp_batch(n):
    g = torch.Generator().manual_seed(100)
    x = torch.randn(n, 26, generator=g); x[:, 0] = torch.randint(0, 4, (n,), generator=g).float()
    y = torch.randn(n, generator=g)
    a = torch.nn.functional.one_hot(torch.randint(0, 11, (n,), generator=g), 11).float()
    return x, y, a


def test_four_rank_data_parallel_trainer_with_unequal_row_counts():
    """Ranks drain different numbers of training rows (one of them none at all): the A3C loss is a SUM over rows, so the summed
    gradients -- and the replicas' Adam steps -- are those of a single trainer on the concatenated batch."""
    from rl_collision_avoidance_amd.config import EnvConfig
    from rl_collision_avoidance_amd.ga3c.network import A3CTrainer, NetworkVP_rnn
    rows = [40, 0, 17, 7]
    port = 29850 + (os.getpid() % 100)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_dp_ragged_worker, args=(4, rows, port, ret), nprocs=4, join=True)
    for r in range(1, 4):
        assert np.array_equal(ret[0], ret[r]), r                # replicas stay in lock step, the empty-handed one too
    net = NetworkVP_rnn(EnvConfig(), seed=4)
    tr = A3CTrainer(net, learning_rate=1e-3, distributed=False)
    x, y, a = _dp_batch(sum(rows))
    for _ in range(4):
        tr.train(x, y, a)
    single = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).numpy()
    np.testing.assert_allclose(ret[0], single, rtol=0, atol=2e-5)
