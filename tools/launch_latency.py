import sys, time, os
sys.path.insert(0, os.getcwd())
import torch
from rl_collision_avoidance_amd.batched_env import BatchedCollisionAvoidanceEnv
from rl_collision_avoidance_amd.config import EnvConfig
env = BatchedCollisionAvoidanceEnv(8192, EnvConfig(), seed=1)
env.reset()
acts = torch.randint(0, 11, (20, 8192, 4), device="cuda", dtype=torch.int32)
slots = env.new_step_slots(20)
x = torch.zeros(8, device="cuda")
def t(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    a = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        a.append((t1 - t0, t2 - t1))
    a.sort(key=lambda p: p[0] + p[1])
    m = a[len(a) // 2]
    return round(m[0] * 1e6, 1), round(m[1] * 1e6, 1)
print("empty torch op (x.add_): call, sync us", t(lambda: x.add_(1)))
print("20-step launch: call, sync us", t(lambda: env.step_autoreset_n(acts, 20, slots=slots)))
print("1-step launch: call, sync us", t(lambda: env.step_autoreset(acts[0])))
print("kernel-only us per 20-step launch", env.kernel_time_ms(acts, 20, 20, slots=slots) * 1e3 if hasattr(env, "kernel_time_ms") else None)
ev = torch.cuda.Event()
def spin():
    env.step_autoreset_n(acts, 20, slots=slots); ev.record()
    while not ev.query(): pass
print("20-step launch + event spin: call(total), sync us", t(spin))
# the launch's fixed cost: kernel time of a K-step launch (per-step output slots) for several K -> slope (per step) and intercept
acts64 = torch.randint(0, 11, (64, 8192, 4), device="cuda", dtype=torch.int32)
slots64 = env.new_step_slots(64)
ks = [1, 2, 4, 8, 12, 16, 20, 24, 32, 48, 64]
us = [env.kernel_time_ms(acts64, 40 * k, k, slots=slots64) * 1e3 for k in ks]
print("K-step launch, kernel us:", {k: round(u, 2) for k, u in zip(ks, us)})
print("  per-step slope 32 -> 64: %.3f us; intercept at K = 20: %.2f us" % ((us[-1] - us[-3]) / 32, us[6] - 20 * (us[-1] - us[-3]) / 32))
print("env:", {k: os.environ.get(k) for k in ("ROC_ACTIVE_WAIT_TIMEOUT", "HSA_ENABLE_INTERRUPT", "HIP_FORCE_SPIN")})
# an ISOLATED launch against the same launch in a train: HIP-event time of ONE 20-step launch (a) behind a host synchronisation, (b) behind another
# kernel on the stream (a torch op of ~20 us on a large tensor: other code, other data), (c) behind another 20-step launch; and the wall clock of
# sync -> launch -> sync after an idle gap of g microseconds of host spinning
big = torch.zeros(64 << 20, device="cuda")
def one(pre):
    v = []
    for _ in range(60):
        torch.cuda.synchronize()
        pre()
        v.append(env.kernel_time_ms(acts, 20, 20, slots=slots) * 1e3)
    v.sort()
    return round(v[len(v) // 2], 2)
print("one 20-step launch, kernel us: behind a host sync %s, behind a torch kernel %s, behind another 20-step launch %s" % (
    one(lambda: None), one(lambda: big.add_(1.0)), one(lambda: env.step_autoreset_n(acts, 20, slots=slots))))
def gap(g_us):
    v = []
    for _ in range(80):
        torch.cuda.synchronize()
        t_end = time.perf_counter() + g_us * 1e-6
        while time.perf_counter() < t_end: pass
        t0 = time.perf_counter(); env.step_autoreset_n(acts, 20, slots=slots); torch.cuda.synchronize(); v.append(time.perf_counter() - t0)
    v.sort()
    return round(v[len(v) // 2] * 1e6, 1)
print("sync -> 20-step launch -> sync, wall us, after an idle gap of 0 / 20 / 100 / 1000 / 10000 us:", [gap(g) for g in (0, 20, 100, 1000, 10000)])
# the same loop on an env made the way bench.py makes its headline env (fresh scenarios from the look-ahead rings), with bench.py's own bracket
env2 = BatchedCollisionAvoidanceEnv(8192, EnvConfig(), seed=7, gen_pool_size=0, gen_lookahead=128)
env2.reset()
slots2 = env2.new_step_slots(20)
dev = torch.device("cuda:0")
for _ in range(20): env2.step_autoreset_n(acts, 20, slots=slots2)
def bench_like(e, sl, sync):
    v = []
    for _ in range(120):
        int(e.episode.to(torch.int64).sum().item())
        sync(); t0 = time.perf_counter(); e.step_autoreset_n(acts, 20, slots=sl); sync(); v.append(time.perf_counter() - t0)
    v.sort()
    return round(v[len(v) // 2] * 1e6, 1), round(v[len(v) // 10] * 1e6, 1), round(v[-len(v) // 10] * 1e6, 1)
print("bench.py's bracket (episode sum .item(); sync; launch; sync), wall us median / p10 / p90:")
print("   pool env,       torch.cuda.synchronize(device):", bench_like(env, slots, lambda: torch.cuda.synchronize(dev)))
print("   pool env,       torch.cuda.synchronize():      ", bench_like(env, slots, lambda: torch.cuda.synchronize()))
print("   look-ahead env, torch.cuda.synchronize(device):", bench_like(env2, slots2, lambda: torch.cuda.synchronize(dev)))
print("   look-ahead env, torch.cuda.synchronize():      ", bench_like(env2, slots2, lambda: torch.cuda.synchronize()))
print("   look-ahead env, kernel us per 20-step launch (events, back to back):", round(env2.kernel_time_ms(acts, 20 * 40, 20, slots=slots2) * 1e3, 2))
