import sys, torch
sys.path.insert(0, "/root/repo")
from rl_collision_avoidance_amd.batched_env import BatchedCollisionAvoidanceEnv
from rl_collision_avoidance_amd.config import EnvConfig
from rl_collision_avoidance_amd.ga3c.network import NetworkVP_rnn
from rl_collision_avoidance_amd.ga3c.policy_kernel import FusedPolicy
from rl_collision_avoidance_amd.ga3c.rollout import BatchedRollout
W, N, K = 8192, 4, 16
cfg = EnvConfig()
env = BatchedCollisionAvoidanceEnv(W, cfg, seed=3)
torch.manual_seed(0)
pol = FusedPolicy(NetworkVP_rnn(cfg).cuda(), seed=5)
roll = BatchedRollout(env, pol, reflush_done=False, ring_len=4 * K + 64)
roll.reset()
for it in range(40):
    roll.run_fused(K); roll.drain(provenance=False)
    if it % 5 == 4:
        obs = roll.obs.view(W, N, -1)
        live = (obs[..., 0] > 0.5) & ((env.game_over.view(W, 1) != 0) | (env.done.view(W, N) == 0))
        per_tile = live.view(W // 16, 64).sum(1)
        nrt = (per_tile + 15) // 16
        print(it * K + K, "live fraction %.3f" % live.float().mean().item(), "n_rt histogram", torch.bincount(nrt, minlength=5).tolist())
