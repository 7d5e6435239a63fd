import sys, torch
sys.path.insert(0, '/root/repo')
from rl_collision_avoidance_amd.batched_env import BatchedCollisionAvoidanceEnv
from rl_collision_avoidance_amd.config import EnvConfig
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
env = BatchedCollisionAvoidanceEnv(8192, EnvConfig(), device="cuda:0", seed=7, gen_pool_size=0, gen_lookahead=128, gen_mode=mode)
env.reset()
g = torch.Generator(device="cuda").manual_seed(1)
acts = torch.randint(0, 11, (64, 8192, 4), generator=g, device="cuda", dtype=torch.int32)
slots = env.new_step_slots(64)
for _ in range(40):
    env.step_autoreset_n(acts, 64, slots=slots)
torch.cuda.synchronize()
