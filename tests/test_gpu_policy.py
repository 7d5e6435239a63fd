"""GPU parity of the fused policy kernel (HIP, through the C ABI: cavoid_policy_*) against the plain PyTorch
float32 ``NetworkVP_rnn.forward`` of the same weights -- a floating-point kernel, so the torch fp32 graph
is its reference (tolerances below); the action selection is integer work and is checked exactly."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle.cavoid_oracle import philox4x32

pytestmark = pytest.mark.gpu

P_TOL = 2e-5       # softmax probabilities (absolute; p <= 1): f32 MFMA vs rocBLAS f32, different summation order
V_TOL = 2e-4       # value head (|v| = O(1)); absolute + relative


def _net(M, seed=0, min_policy=0.0, normalize=True, A=11):
    from rl_collision_avoidance_amd.config import EnvConfig
    from rl_collision_avoidance_amd.ga3c.network import NetworkVP_rnn

    class Cfg(EnvConfig):
        def __init__(self):
            self.MAX_NUM_AGENTS_IN_ENVIRONMENT = M + 1
            EnvConfig.__init__(self)
    cfg = Cfg()
    cfg.MIN_POLICY = min_policy
    cfg.NORMALIZE_INPUT = normalize
    net = NetworkVP_rnn(cfg, num_actions=A, seed=seed).cuda()
    g = torch.Generator().manual_seed(seed + 100)
    with torch.no_grad():                      # non-zero biases so that every bias path is exercised
        for name, prm in net.named_parameters():
            if name.endswith("_bias"):
                prm.copy_((torch.rand(prm.shape, generator=g) - 0.5).to(prm.device))
    return net


def _inputs(net, B, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    M = net.max_others
    x = torch.randn((B, net.input_size), generator=g) * scale * net.std.cpu() + net.avg.cpu()
    x[:, 0] = torch.randint(0, M + 1, (B,), generator=g).to(torch.float32)
    return x.cuda()


@pytest.mark.parametrize("M,B", [(3, 1), (3, 63), (3, 64), (3, 130), (3, 32768), (9, 1000), (19, 517), (1, 200)])
def test_forward_matches_torch_fp32(M, B):
    from rl_collision_avoidance_amd.ga3c.policy_kernel import FusedPolicy
    net = _net(M, seed=M)
    pol = FusedPolicy(net)
    x = _inputs(net, B, seed=B)
    p, v = pol(x)
    with torch.no_grad():
        _, p_ref, v_ref = net.forward(x)
    assert p.shape == (B, 11) and v.shape == (B,)
    assert torch.isfinite(p).all() and torch.isfinite(v).all()
    assert (p - p_ref).abs().max().item() <= P_TOL
    assert ((v - v_ref).abs() <= V_TOL + V_TOL * v_ref.abs()).all()
    assert (p.sum(dim=1) - 1.0).abs().max().item() <= 1e-5


@pytest.mark.parametrize("min_policy,normalize", [(1e-3, True), (0.0, False)])
def test_min_policy_and_unnormalised_input(min_policy, normalize):
    from rl_collision_avoidance_amd.ga3c.policy_kernel import FusedPolicy
    net = _net(3, seed=5, min_policy=min_policy, normalize=normalize)
    pol = FusedPolicy(net)
    x = _inputs(net, 777, seed=1, scale=0.3 if not normalize else 1.0)
    p, v = pol(x)
    with torch.no_grad():
        _, p_ref, v_ref = net.forward(x)
    assert (p - p_ref).abs().max().item() <= P_TOL
    assert ((v - v_ref).abs() <= V_TOL + V_TOL * v_ref.abs()).all()


def test_runs_on_the_env_observation_tensor_in_place():
    """The kernel reads the env's obs tensor through a row stride (column 0 'is_learning' skipped): no slice copy."""
    from rl_collision_avoidance_amd.batched_env import BatchedCollisionAvoidanceEnv
    from rl_collision_avoidance_amd.ga3c.policy_kernel import FusedPolicy
    env = BatchedCollisionAvoidanceEnv(300, seed=3, gen_min_agents=2)
    net = _net(3, seed=9)
    pol = FusedPolicy(net)
    obs = env.reset()
    rng = np.random.default_rng(0)
    for _ in range(25):
        obs, _, _, _ = env.step_autoreset(torch.from_numpy(rng.integers(0, 11, size=(300, 4)).astype(np.int32)).cuda())
    view = obs.view(300 * 4, -1)[:, 1:]
    assert not view.is_contiguous()
    p, v = pol(view)
    with torch.no_grad():
        _, p_ref, v_ref = net.forward(view.contiguous())
    assert len(torch.unique(view[:, 0])) >= 2            # rows with different neighbour counts
    assert (p - p_ref).abs().max().item() <= P_TOL
    assert ((v - v_ref).abs() <= V_TOL + V_TOL * v_ref.abs()).all()


def test_refresh_picks_up_new_weights():
    from rl_collision_avoidance_amd.ga3c.policy_kernel import FusedPolicy
    net = _net(3, seed=2)
    pol = FusedPolicy(net)
    x = _inputs(net, 256, seed=4)
    p0, _ = pol(x)
    with torch.no_grad():
        for prm in net.parameters():
            prm.mul_(0.5)
    p_stale, _ = pol(x)
    assert torch.equal(p_stale, p0)                         # the handle holds its own packed copy
    pol.refresh()
    p1, v1 = pol(x)
    with torch.no_grad():
        _, p_ref, v_ref = net.forward(x)
    assert (p1 - p_ref).abs().max().item() <= P_TOL and not torch.allclose(p1, p0)


def test_greedy_is_argmax_and_sampling_is_the_inverse_cdf():
    from rl_collision_avoidance_amd.ga3c.policy_kernel import FusedPolicy
    net = _net(3, seed=7)
    SEED = 0x1234567890AB
    pol = FusedPolicy(net, seed=SEED)
    B = 5000
    x = _inputs(net, B, seed=8)
    a_g, p, _ = pol.act(x, greedy=True)
    assert torch.equal(a_g.long(), p.argmax(dim=1))
    draws = []
    for _ in range(3):
        a, p, _ = pol.act(x)
        draws.append(a.cpu().numpy())
    pn = p.cpu().numpy().astype(np.float64)
    cdf = np.cumsum(pn, axis=1)
    # every launch that selects actions advances the device-side counter: greedy was launch 0
    for k, a in enumerate(draws):
        step = 1 + k
        bad = 0
        for row in range(0, B, 7):
            bits = philox4x32(row, 0, step, 0x504F4C, SEED & 0xFFFFFFFF, SEED >> 32)[0]
            u = (bits >> 8) / 16777216.0
            expect = min(int(np.sum(cdf[row] <= u * cdf[row, -1])), 10)
            if expect != a[row]:
                # only a draw within float32 rounding of a CDF boundary may land on the neighbour
                assert np.min(np.abs(cdf[row] - u * cdf[row, -1])) < 1e-6, (row, step, expect, a[row])
                bad += 1
        assert bad <= 2
    assert not np.array_equal(draws[0], draws[1])           # a fresh stream per launch
    pol.seed(SEED)                                          # re-seeding rewinds the counter
    pol.act(x, greedy=True)
    again, _, _ = pol.act(x)
    assert np.array_equal(again.cpu().numpy(), draws[0])


def test_sampling_frequencies_follow_p():
    from rl_collision_avoidance_amd.ga3c.policy_kernel import FusedPolicy
    net = _net(3, seed=11)
    pol = FusedPolicy(net, seed=99)
    x = _inputs(net, 1, seed=3).repeat(200000, 1)
    a, p, _ = pol.act(x)
    freq = torch.bincount(a.long(), minlength=11).double() / a.numel()
    assert (freq - p[0].double()).abs().max().item() < 5e-3


def test_rollout_with_fused_policy_in_a_graph():
    """BatchedRollout accepts the fused policy (strided obs, in-kernel action selection) and captures it."""
    from rl_collision_avoidance_amd.batched_env import BatchedCollisionAvoidanceEnv
    from rl_collision_avoidance_amd.ga3c.policy_kernel import FusedPolicy
    from rl_collision_avoidance_amd.ga3c.rollout import BatchedRollout
    env = BatchedCollisionAvoidanceEnv(512, seed=5)
    net = _net(3, seed=13)
    roll = BatchedRollout(env, FusedPolicy(net, seed=1), reflush_done=False)
    roll.reset()
    roll.capture(steps_per_graph=4)
    roll.replay(30)
    batch = roll.drain(flush_all=True)
    assert len(batch) > 20000 and batch.dropped == 0
    assert int(batch.a_index.min()) >= 0 and int(batch.a_index.max()) <= 10
    assert len(torch.unique(batch.a_index)) == 11
    eps = roll.drain_episodes()
    assert eps.shape[0] > 100


def test_train_cli_saves_resumes_and_plays(tmp_path, capsys):
    """The training loop end to end on one GPU: trains (every drained row), writes network_%08d.pt, resumes from it and
    runs PLAY_MODE (argmax, trainers off) on the loaded weights."""
    import glob
    from rl_collision_avoidance_amd.ga3c import train
    ck = str(tmp_path / "ck")
    train.main(["--worlds", "256", "--episodes", "600", "--print-every", "0", "--checkpoint-dir", ck, "--save-every", "300",
                "--train-rows", "4096"])
    out = capsys.readouterr().out
    assert "finished" in out and "training steps" in out
    files = sorted(glob.glob(ck + "/network_*.pt"))
    assert len(files) >= 2
    state = torch.load(files[-1], map_location="cpu")
    assert state["episode"] >= 600 and state["training_step"] > 0 and "lstm_kernel" in state["model"]
    train.main(["--worlds", "256", "--episodes", "200", "--print-every", "0", "--load", files[-1], "--lr-end", "1e-5"])
    assert "finished" in capsys.readouterr().out
    train.main(["--worlds", "256", "--episodes", "200", "--print-every", "0", "--load", files[-1], "--play"])
    out = capsys.readouterr().out
    assert "finished" in out and " 0 training steps" in out


@pytest.mark.parametrize("M,B,min_policy", [(3, 64, 0.0), (3, 1000, 1e-3), (3, 32768, 0.0), (9, 700, 0.0), (1, 130, 0.0)])
def test_fused_trainer_gradients_match_autograd(M, B, min_policy):
    """cavoid_policy_train + the weight-gradient GEMMs vs PyTorch autograd on NetworkVP_rnn.loss (same weights, same batch)."""
    from rl_collision_avoidance_amd.ga3c.policy_kernel import FusedA3CTrainer
    net = _net(M, seed=20 + M, min_policy=min_policy)
    x = _inputs(net, B, seed=B + 1)
    g = torch.Generator().manual_seed(B)
    y = torch.randn(B, generator=g).cuda()
    a = torch.randint(0, 11, (B,), generator=g).cuda()
    onehot = torch.nn.functional.one_hot(a, 11).float()
    import copy
    ref_net = copy.deepcopy(net).double()                    # float64 autograd = the yardstick for both float32 paths
    total, cost_p, cost_v = ref_net.loss(x.double(), y.double(), onehot.double())
    total.backward()
    want = {k: v.grad.clone() for k, v in ref_net.named_parameters()}
    net.zero_grad()
    net.loss(x, y, onehot)[0].backward()
    torch32 = {k: v.grad.clone() for k, v in net.named_parameters()}
    tr = FusedA3CTrainer(net, learning_rate=0.0)             # lr 0: the optimiser step leaves the weights alone
    loss = float(tr.train(x, y, a))
    assert abs(loss - float(total.detach())) <= 2e-4 * max(1.0, abs(float(total.detach())))
    for k, v in net.named_parameters():
        ref = want[k]
        scale = ref.abs().max().item() + 1e-6
        err = (v.grad.double() - ref).abs().max().item()
        err32 = (torch32[k].double() - ref).abs().max().item()
        # as close to the float64 gradient as PyTorch's own float32 autograd is (x3), or 1e-4 of the largest entry ...
        tight = max(3.0 * err32, 1e-4 * scale)
        if err > tight:
            # ... except for the gradient paths behind a relu whose pre-activation is 0 to float32 rounding: among
            # 25 M units (B = 32768) a handful sit there, float32 and float64 then take different sub-gradients, and
            # one unit's contribution to the weight gradients flips.  The heads never see that.
            assert B >= 8192 and not k.startswith(("p_", "v_")), (k, err, err32, scale)
            bad = ((v.grad.double() - ref).abs() > tight).float().mean().item()
            assert err <= 5e-3 * scale and bad <= 5e-3, (k, err, bad, scale)


def test_fused_trainer_learns_like_the_autograd_trainer():
    """Same batches, same Adam: the two trainers end at the same weights (to float32 accumulation differences)."""
    import copy
    from rl_collision_avoidance_amd.ga3c.network import A3CTrainer
    from rl_collision_avoidance_amd.ga3c.policy_kernel import FusedA3CTrainer
    net_a = _net(3, seed=31)
    net_b = copy.deepcopy(net_a)
    ta, tb = A3CTrainer(net_a, learning_rate=1e-4), FusedA3CTrainer(net_b, learning_rate=1e-4)
    for step in range(5):
        x = _inputs(net_a, 4096, seed=100 + step)
        g = torch.Generator().manual_seed(step)
        y = torch.randn(4096, generator=g).cuda()
        a = torch.randint(0, 11, (4096,), generator=g).cuda()
        la = ta.train(x, y, torch.nn.functional.one_hot(a, 11).float())
        lb = float(tb.train(x, y, a))
        assert abs(la - lb) <= 1e-3 * max(1.0, abs(la))
    for (k, pa), (_, pb) in zip(net_a.named_parameters(), net_b.named_parameters()):
        assert (pa - pb).abs().max().item() <= 2e-5, k


def test_trainers_accept_an_empty_batch():
    """Multi-GPU runs give every rank the same number of optimiser steps; a rank whose drain came up short trains on
    zero rows (it still has to join the gradient all-reduce)."""
    from rl_collision_avoidance_amd.ga3c.network import A3CTrainer
    from rl_collision_avoidance_amd.ga3c.policy_kernel import FusedA3CTrainer
    net = _net(3, seed=41)
    x = _inputs(net, 128, seed=5)
    y = torch.zeros(128).cuda()
    a = torch.zeros(128, dtype=torch.int64).cuda()
    for tr in (A3CTrainer(net), FusedA3CTrainer(net)):
        loss = tr.train(x[:0], y[:0], a[:0] if isinstance(tr, FusedA3CTrainer) else torch.nn.functional.one_hot(a[:0], 11).float())
        assert float(loss) == 0.0 and tr.training_step == 1


def test_row_list_pass_equals_the_full_pass_on_the_listed_rows():
    from rl_collision_avoidance_amd.ga3c.policy_kernel import FusedPolicy
    net = _net(3, seed=51)
    B = 5000
    x = _inputs(net, B, seed=6)
    g = torch.Generator().manual_seed(1)
    listed = torch.randperm(B, generator=g)[:3210].to(torch.int32).cuda()          # unordered, not a multiple of the tile
    index = torch.zeros(B, dtype=torch.int32, device="cuda")
    index[:listed.numel()] = listed
    count = torch.tensor([listed.numel()], dtype=torch.int32, device="cuda")
    pol = FusedPolicy(net, seed=77)
    a_full, p_full, v_full = pol.act(x)
    pol.seed(77)                                             # same launch counter -> same draws per row
    a_rows, p_rows, v_rows = pol.act(x, rows=(index, count))
    sel = listed.long()
    assert torch.equal(p_rows[sel], p_full[sel]) and torch.equal(v_rows[sel], v_full[sel]) and torch.equal(a_rows[sel], a_full[sel])
    rest = torch.ones(B, dtype=torch.bool, device="cuda")
    rest[sel] = False
    assert float(p_rows[rest].abs().sum()) == 0.0 and float(v_rows[rest].abs().sum()) == 0.0
    count.zero_()                                            # an empty list is fine (every workgroup leaves at once)
    a0, p0, v0 = pol.act(x, rows=(index, count))
    assert float(p0.abs().sum()) == 0.0


def test_skipping_finished_agents_does_not_change_the_rollout():
    """No policy pass for agents that are done and wait for their world to end: same trajectories, same training rows."""
    from rl_collision_avoidance_amd.batched_env import BatchedCollisionAvoidanceEnv
    from rl_collision_avoidance_amd.ga3c.policy_kernel import FusedPolicy
    from rl_collision_avoidance_amd.ga3c.rollout import BatchedRollout
    net = _net(3, seed=61)
    out = []
    for skip in (False, True):
        env = BatchedCollisionAvoidanceEnv(256, seed=9, gen_min_agents=2)
        roll = BatchedRollout(env, FusedPolicy(net, seed=3), reflush_done=False, skip_finished=skip, ring_len=200)
        roll.reset()
        for _ in range(150):
            roll.step()
        b = roll.drain(flush_all=True)
        key = b.src[:, 0].long() * 100000 + b.src[:, 1].long() * 10000 + b.src[:, 2].long()
        order = torch.argsort(key)
        out.append((key[order], b.x[order], b.r[order], b.a_index[order], roll.drain_episodes()))
        if skip:
            assert int(roll.row_count.item()) < 256 * 4     # some rows really were skipped in the last step
        roll.close(); env.close()
    (k0, x0, r0, a0, e0), (k1, x1, r1, a1, e1) = out
    assert torch.equal(k0, k1) and torch.equal(x0, x1) and torch.equal(r0, r1) and torch.equal(a0, a1) and k0.numel() > 20000
    assert e0.shape == e1.shape and e0.shape[0] > 100


def test_regression_pretraining_clones_the_teacher():
    """Supervised initialisation (the role of Regression.py): after a few hundred rows per step the network picks the
    teacher's action on fresh observations."""
    from rl_collision_avoidance_amd.batched_env import BatchedCollisionAvoidanceEnv
    from rl_collision_avoidance_amd.ga3c.regression import pretrain, teacher_actions
    env = BatchedCollisionAvoidanceEnv(512, seed=4)
    net = _net(3, seed=71)
    info = pretrain(net, env, steps=60, learning_rate=2e-3, rows_per_step=8192)
    assert info["steps"] == 60 and info["p_loss_per_row"] < 0.2
    obs = env.reset()
    table = torch.as_tensor(__import__("rl_collision_avoidance_amd.actions", fromlist=["Actions"]).Actions().actions,
                            dtype=torch.float32, device="cuda")
    with torch.no_grad():
        _, p, _ = net.forward(obs.view(512 * 4, -1)[:, 1:].contiguous())
    agree = (p.argmax(dim=1).view(512, 4) == teacher_actions(obs, table).long()).float().mean().item()
    assert agree > 0.9


@pytest.mark.parametrize("A", [5, 15])
def test_other_action_counts(A):
    """NUM_ACTIONS is a Config value (Config.py:79): heads of 5 and 15 logits through the inference kernel, the action draw
    and the trainer pass."""
    from rl_collision_avoidance_amd.ga3c.policy_kernel import FusedA3CTrainer, FusedPolicy
    net = _net(3, seed=80 + A, A=A)
    B = 777
    x = _inputs(net, B, seed=A)
    pol = FusedPolicy(net, seed=5)
    a_s, p, v = pol.act(x)
    with torch.no_grad():
        _, p_ref, v_ref = net.forward(x)
    assert p.shape == (B, A) and (p - p_ref).abs().max().item() <= P_TOL and int(a_s.max()) < A and int(a_s.min()) >= 0
    g = torch.Generator().manual_seed(A)
    y = torch.randn(B, generator=g).cuda()
    a = torch.randint(0, A, (B,), generator=g).cuda()
    net.zero_grad()
    net.loss(x, y, torch.nn.functional.one_hot(a, A).float())[0].backward()
    want = {k: t.grad.clone() for k, t in net.named_parameters()}
    FusedA3CTrainer(net, pol, learning_rate=0.0).train(x, y, a)
    for k, t in net.named_parameters():
        scale = want[k].abs().max().item() + 1e-6
        assert (t.grad - want[k]).abs().max().item() <= 2e-4 * scale, k


def test_evaluate_reports_consistent_outcome_rates():
    """EVALUATE_MODE runs: every learning agent ends in exactly one of goal / collision / timeout, a go-to-goal policy
    on 2-agent worlds mostly arrives, and a stand-still policy always times out."""
    from rl_collision_avoidance_amd.actions import Actions
    from rl_collision_avoidance_amd.batched_env import BatchedCollisionAvoidanceEnv
    from rl_collision_avoidance_amd.ga3c.evaluate import evaluate
    from rl_collision_avoidance_amd.ga3c.regression import teacher_actions
    table = torch.as_tensor(Actions().actions, dtype=torch.float32, device="cuda")

    def go_to_goal(x):                                       # x = obs[:, 1:], so the ego heading is column 2
        obs = torch.cat([torch.ones_like(x[:, :1]), x], dim=1).unsqueeze(0)
        a = teacher_actions(obs, table).view(-1).long()
        return torch.nn.functional.one_hot(a, 11).float(), torch.zeros(x.shape[0], device=x.device)

    def stand_still(x):
        p = torch.zeros((x.shape[0], 11), device=x.device)
        p[:, 9] = 1.0                                        # (speed 0, no turn)
        return p, torch.zeros(x.shape[0], device=x.device)

    env = BatchedCollisionAvoidanceEnv(512, seed=2, gen_min_agents=2, gen_max_agents=2, evaluate_mode=1)
    r = evaluate(env, go_to_goal, rounds=2)
    assert r["agents"] == 2 * 512 * 2
    assert abs(r["success_rate"] + r["collision_rate"] + r["timeout_rate"] + r["unfinished_rate"] - 1.0) < 1e-9
    assert r["unfinished_rate"] == 0.0
    # two agents swapping places head-on meet in the middle: mostly collisions, the reward says the same
    assert r["collision_rate"] > 0.5 and r["mean_reward"] < 0.5
    r0 = evaluate(env, stand_still, rounds=1)
    assert r0["timeout_rate"] == 1.0 and r0["success_rate"] == 0.0 and abs(r0["mean_reward"]) < 1e-6
    env.close()


@pytest.mark.parametrize("kernel", ["f16", "split3", "split4", "split5", "f32"])
def test_both_inference_kernels_against_a_float64_yardstick(kernel, monkeypatch):
    """The inference kernel computes its float32 GEMMs by operand splitting on the 16-bit matrix pipe, float32 accumulate.  Default
    (round 4): float16 pieces -- weights and activations in two pieces each (22 bits), three partial products: a float32-GRADE form,
    held to the bar a float32 predictor is held to (|dp| <= 1e-6, |dv| <= 5e-6 against the SAME network evaluated in float64 -- the
    reference's predictor is TensorFlow float32, ThreadPredictor.py:46,67), also with large inputs (scale 4: saturating gates).
    CAVOID_POLICY_PRODUCTS = 3 / 4 / 5: bf16 pieces (round 3's forms; their 16-bit activation pieces set an error of ~2^-17 per
    product: a quarter of the 2e-5 / 2e-4 bar); CAVOID_POLICY_F32 = 1: the float32-MFMA kernel."""
    import copy
    from rl_collision_avoidance_amd.ga3c.policy_kernel import FusedPolicy
    monkeypatch.setenv("CAVOID_POLICY_F32", "1" if kernel == "f32" else "0")
    monkeypatch.setenv("CAVOID_POLICY_PRODUCTS", {"split3": "3", "split4": "4", "split5": "5"}.get(kernel, "16"))
    f32_grade = kernel in ("f16", "f32")
    for M, B, scale in ((3, 4096, 1.0), (9, 2048, 1.0), (3, 4096, 4.0)):
        net = _net(M, seed=40 + M)
        pol = FusedPolicy(net)
        x = _inputs(net, B, seed=7, scale=scale)
        p, v = pol(x)
        with torch.no_grad():
            _, p32, v32 = net.forward(x)
            net64 = copy.deepcopy(net).double()
            _, p64, v64 = net64.forward(x.double())
        e_kernel_p, e_torch_p = (p.double() - p64).abs().max().item(), (p32.double() - p64).abs().max().item()
        e_kernel_v, e_torch_v = (v.double() - v64).abs().max().item(), (v32.double() - v64).abs().max().item()
        assert e_kernel_p <= P_TOL and e_kernel_v <= V_TOL * (1.0 + v64.abs().max().item()), (kernel, M, scale, e_kernel_p, e_kernel_v)
        if f32_grade:
            # float32 grade: the bar of the round-3 verdict (measured: f16 4.4e-8 / 3.7e-7, x4 inputs 1.9e-7 / 1.4e-6; the float32-MFMA
            # kernel 3.9e-8 / 2.2e-7, x4 1.8e-7 / 8.7e-7; profiles/r04_split_f16_vs_bf16.txt) ...
            assert e_kernel_p <= 1e-6 and e_kernel_v <= 5e-6, (kernel, M, scale, e_kernel_p, e_torch_p, e_kernel_v, e_torch_v)
            # ... and within a small factor of what the float32 PyTorch graph itself differs from float64 by
            assert e_kernel_p <= 6.0 * e_torch_p + 5e-8 and e_kernel_v <= 6.0 * e_torch_v + 5e-7, (kernel, M, scale, e_kernel_p, e_torch_p, e_kernel_v, e_torch_v)
        else:
            # the bf16 activation pieces carry 16 significant bits, so these forms sit above float32 rounding (measured at scale 4:
            # p 3.4e-6 / 2.8e-6 / 2.1e-6 with 3 / 4 / 5 products against 1e-7 for the float32 graph) -- a factor >= 4 inside the bar
            assert e_kernel_p <= P_TOL / 4 and e_kernel_v <= V_TOL / 4 * (1.0 + v64.abs().max().item()), \
                (kernel, M, scale, e_kernel_p, e_torch_p, e_kernel_v, e_torch_v)
        print("policy kernel %s M=%d scale=%g: |dp| %.2e (torch f32 %.2e)  |dv| %.2e (torch f32 %.2e)"
              % (kernel, M, scale, e_kernel_p, e_torch_p, e_kernel_v, e_torch_v))


def test_float16_pieces_saturate_instead_of_overflowing():
    """float16's range is the price of the float32-grade split: a piece saturates at +-65504 (inputs and relu outputs are clamped
    there) -- an absurd input must degrade, never turn into inf / NaN; and up to that range the kernel stays on the float32 graph."""
    from rl_collision_avoidance_amd.ga3c.policy_kernel import FusedPolicy
    net = _net(3, seed=3)
    pol = FusedPolicy(net)
    x = _inputs(net, 512, seed=1)
    big = x.clone()
    big[:, 1:] *= 1e7                                        # far beyond float16
    p, v = pol(big)
    assert torch.isfinite(p).all() and torch.isfinite(v).all()
    assert (p.sum(dim=1) - 1.0).abs().max().item() <= 1e-5 and (p >= 0).all()
    mid = x.clone()
    mid[:, 1:] *= 300.0                                      # large but representable: |x| up to a few thousand
    p, v = pol(mid)
    with torch.no_grad():
        _, p_ref, v_ref = net.forward(mid)
    assert (p - p_ref).abs().max().item() <= P_TOL and ((v - v_ref).abs() <= V_TOL + V_TOL * v_ref.abs()).all()


def test_weights_beyond_float16_range_are_reported_not_silently_clamped(monkeypatch):
    """the default float16 split clamps a weight at +-65504.  The load counts what it clamped (cavoid_policy_info) and the host
    side refuses such a network loudly; the bf16-piece form (float32's range) takes it; the handle knows its own inference form."""
    from rl_collision_avoidance_amd.ga3c.policy_kernel import FusedPolicy
    net = _net(3, seed=5)
    pol = FusedPolicy(net)
    assert pol.inference_form == ("split", 16) and pol.clamped_weights() == 0
    with torch.no_grad():
        net.layer2_kernel[3, 7] = 1.0e5
        net.lstm_kernel[0, 1] = -5.0e4            # inside +-65504 as stored, beyond it once the gate column's log2 e scale is applied
    pol.refresh()                                   # the hot path does not check (no synchronisation) ...
    assert pol.clamped_weights() == 2             # ... but the count is there
    with pytest.raises(ValueError, match="65504"):
        pol.refresh(check_range=True)
    with pytest.raises(ValueError, match="65504"):
        FusedPolicy(net)
    monkeypatch.setenv("CAVOID_POLICY_PRODUCTS", "3")
    wide = FusedPolicy(net)                         # bf16 pieces: no range limit, no complaint
    assert wide.inference_form == ("split", 3) and wide.clamped_weights() == 0
    monkeypatch.delenv("CAVOID_POLICY_PRODUCTS")
    assert wide.inference_form == ("split", 3)      # fixed at creation: the environment no longer matters


def test_fused_actor_availability_follows_the_handles_not_the_environment(monkeypatch):
    """`fused_available` must say no (with the reason) where cavoid_actor_run would return CAVOID_EUNSUPPORTED -- a policy
    handle created under another inference form, or an env that generates frozen-network agents without a frozen policy -- and must not
    be fooled by environment variables set AFTER the handle was made."""
    from rl_collision_avoidance_amd.batched_env import BatchedCollisionAvoidanceEnv
    from rl_collision_avoidance_amd.config import EnvConfig
    from rl_collision_avoidance_amd.ga3c.policy_kernel import FusedPolicy
    from rl_collision_avoidance_amd.ga3c.rollout import BatchedRollout
    net = _net(3, seed=9)
    env = BatchedCollisionAvoidanceEnv(128, EnvConfig(), device="cuda:0", seed=1)
    pol = FusedPolicy(net)
    roll = BatchedRollout(env, pol)
    assert roll.fused_available
    monkeypatch.setenv("CAVOID_POLICY_PRODUCTS", "5")                # too late for `pol`: it stays on the default form
    assert roll.fused_available
    other = FusedPolicy(net)                                         # created under the switch
    roll5 = BatchedRollout(env, other)
    assert not roll5.fused_available and "non-default inference form" in roll5.fused_unavailable_reason
    monkeypatch.delenv("CAVOID_POLICY_PRODUCTS")
    assert not roll5.fused_available                                 # still the handle's form
    roll.close(); roll5.close(); env.close()
    # an env that generates frozen-network agents cannot even get a rollout without their network (nothing can fall through to the
    # fused kernel handing those agents the learner's sample); should the env's cfg change under a live rollout, the reason says so
    env2 = BatchedCollisionAvoidanceEnv(128, EnvConfig(), device="cuda:0", seed=1, gen_nonlearning_fraction=0.5, gen_frozen_fraction=0.5)
    with pytest.raises(ValueError, match="frozen_policy"):
        BatchedRollout(env2, pol)
    env3 = BatchedCollisionAvoidanceEnv(128, EnvConfig(), device="cuda:0", seed=1)
    roll3 = BatchedRollout(env3, pol)
    env3.cfg.gen_nonlearning_fraction, env3.cfg.gen_frozen_fraction = 0.5, 0.5
    assert not roll3.fused_available and "frozen-network agents" in roll3.fused_unavailable_reason and "one launch per phase" in roll3.actor_path
    roll3.close(); env3.close(); env2.close()


@pytest.mark.parametrize("form", ["oct", "duo", "pipe"])
@pytest.mark.parametrize("M,B,scale", [(3, 5000, 1.0), (9, 2111, 1.0), (3, 4096, 4.0), (1, 300, 1.0), (3, 64, 1.0), (5, 129, 1.0)])
def test_other_tile_to_wavefront_forms_are_bit_identical(M, B, scale, form, monkeypatch):
    """Two other mappings of the same pass (CAVOID_POLICY_FORM, read at cavoid_policy_create):
    oct -- eight wavefronts per 64-row tile, each owning 32 output columns (cavoid_policy_split8.hpp; the LSTM's gate columns re-ordered so
           that a lane still holds all four gates of its hidden units);
    duo -- two tiles per workgroup of eight wavefronts, the second tile one barrier behind the first, so that one tile's matrix phases run
           beside the other's vector phases (policy_forward_split_duo_kernel); a tile with fewer LSTM steps idles through the difference.
    pipe -- the LSTM steps and layer1 as a software pipeline over row halves: one half's cell update / epilogue inside the other half's matrix
           instructions (cavoid_policy_pipe.hpp).
    Every output element is the same float32 sum in the same order: probabilities, values and the drawn actions are BIT for bit the default
    form's -- full pass and row-list pass, ragged observed-agent counts, odd tile counts, a last tile with one row."""
    from rl_collision_avoidance_amd.ga3c.policy_kernel import FusedPolicy
    net = _net(M, seed=60 + M)
    x = _inputs(net, B, seed=11, scale=scale)
    g = torch.Generator().manual_seed(3)
    x[:, 0] = torch.randint(0, M + 1, (B,), generator=g).float().cuda()
    x[: B // 3, 0] = float(M)                               # tiles in which every row is live at every LSTM step
    x[B // 3: B // 3 + 70, 0] = 1.0                         # ... and a tile with fewer steps than its neighbour (duo: the idle barrier pairs)
    monkeypatch.setenv("CAVOID_POLICY_FORM", "quad")
    pol4 = FusedPolicy(net, seed=77)
    monkeypatch.setenv("CAVOID_POLICY_FORM", form)
    pol8 = FusedPolicy(net, seed=77)
    a4, p4, v4 = pol4.act(x)
    a8, p8, v8 = pol8.act(x)
    assert torch.equal(p4, p8) and torch.equal(v4, v8) and torch.equal(a4, a8)
    assert float((p4.sum(dim=1) - 1.0).abs().max()) < 1e-5 and (B < 100 or a4.float().std().item() > 0)
    idx = torch.zeros(B, dtype=torch.int32, device="cuda")
    listed = torch.randperm(B, generator=torch.Generator().manual_seed(1))[: B // 2 + 3].to(torch.int32).cuda()
    idx[: listed.numel()] = listed
    count = torch.tensor([listed.numel()], dtype=torch.int32, device="cuda")
    a4r, p4r, v4r = pol4.act(x, rows=(idx, count))
    a8r, p8r, v8r = pol8.act(x, rows=(idx, count))
    assert torch.equal(p4r, p8r) and torch.equal(v4r, v8r) and torch.equal(a4r, a8r)
    count.zero_()                                            # an empty list: every workgroup leaves at once
    a0, p0, v0 = pol8.act(x, rows=(idx, count))
    assert float(p0.abs().sum()) == 0.0
