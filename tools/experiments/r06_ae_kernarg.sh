# where the kernel arguments live: HIP_FORCE_DEV_KERNARG = 0 / 1 / unset (the runtime's default) -- a launch's first scalar loads read them.  One box, kbench (HIP events).
o=$PWD/gpurun_out/r06_ae; mkdir -p $o
kb() { echo -n "$1 [$2]: "; env $2 timeout 300 python tools/kbench.py --worlds 8192 --agents 4 --spl 1 20 64 2>&1 | grep us_per | sed 's/"Gagent.*//' | tr '\n' ' '; echo; }
{
for rep in 1 2 3; do
  kb default "CAVOID_X=0"
  kb dev0 "HIP_FORCE_DEV_KERNARG=0"
  kb dev1 "HIP_FORCE_DEV_KERNARG=1"
done
for v in "CAVOID_X=0" "HIP_FORCE_DEV_KERNARG=0" "HIP_FORCE_DEV_KERNARG=1"; do
  echo -n "bench K=20 [$v]: "; env $v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-full-loop --no-configs3 --no-pmc --no-fresh-scenarios --evidence off 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('value %.4e wall_us_per_step %.4f kernel_us %.3f one-step %.3f null_roundtrip %.1f' % (d['value'], d['ms_per_step'] * 1e3, r['kernel_us'], r['one_step_launch']['kernel_us'], r['wall_clock']['null_launch_roundtrip_us']))"
done
} > $o/kernarg.txt 2>&1
cat $o/kernarg.txt
