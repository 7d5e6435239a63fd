# same-box A/B: D, P and L at priority 3 while they make the last step's observation together (tail3) against as they were (tail_old: the loader at 0, below the
# consumers that are still at their last steps).  bench.py's K = 20 form (kernel us per launch, HIP events; wall clock) and kbench, four interleaved repetitions.
o=$PWD/gpurun_out/r06_ad; mkdir -p $o
bn() { echo -n "$1 bench $2: "; CAVOID_LIB=$PWD/.ab/lib$1.so timeout 300 python bench.py $2 --no-cpu-baseline --no-full-loop --no-configs3 --no-pmc --no-fresh-scenarios --evidence off 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('value %.4e wall_us_per_step %.4f kernel_us %.3f frac %.4f' % (d['value'], d['ms_per_step'] * 1e3, r['kernel_us'], r['frac']))"; }
kb() { echo -n "$1: "; CAVOID_LIB=$PWD/.ab/lib$1.so timeout 300 python tools/kbench.py --worlds 8192 --agents 4 --spl 8 20 64 2>&1 | grep us_per | sed 's/"Gagent.*//' | tr '\n' ' '; echo; }
{
for rep in 1 2 3 4; do for v in tail_old tail3; do bn $v "--steps 20 --warmup 5"; done; done
for rep in 1 2; do for v in tail_old tail3; do kb $v; done; done
for v in tail_old tail3; do bn $v ""; done
( CAVOID_LIB=$PWD/.ab/libtail3.so timeout 200 python tools/relay_soak.py 40 2>&1 | grep -a "soak\|MISMATCH" )
} > $o/tail_prio.txt 2>&1
cat $o/tail_prio.txt
