# same-box: (1) the relay-carried GPU tests on the product (write-through slot stores with their hazard distance) -- and on old_nt if they fail;
# (2) depth2 (CAVOID_RELAY_DEPTH=2: D two steps ahead of P's verdicts): relay soak + the same tests; (3) kbench and the bench lines, product / depth2 / old_nt.
o=$PWD/gpurun_out/r06_z; mkdir -p $o
T="tests/test_gpu_packed.py tests/test_gpu_parity.py tests/test_gpu_relay_fault.py tests/test_gpu_lookahead.py"
flt() { grep -v "amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl"; }
( timeout 900 python -m pytest $T -x -q --tb=short 2>&1 | flt | tail -40 ) > $o/tests_product.txt
if ! grep -q " passed" $o/tests_product.txt || grep -q "failed" $o/tests_product.txt; then
  ( CAVOID_LIB=$PWD/.ab/libold_nt.so timeout 900 python -m pytest $T -x -q --tb=short 2>&1 | flt | tail -40 ) > $o/tests_old_nt.txt
fi
( CAVOID_LIB=$PWD/.ab/libdepth2.so timeout 200 python tools/relay_soak.py 60 2>&1 | grep "soak\|MISMATCH" ) > $o/soak_depth2.txt
( CAVOID_LIB=$PWD/.ab/libdepth2.so timeout 900 python -m pytest tests/test_gpu_packed.py tests/test_gpu_parity.py tests/test_gpu_lookahead.py -x -q --tb=short 2>&1 | flt | tail -40 ) > $o/tests_depth2.txt
kb() { echo -n "$1: "; if [ "$1" = product ]; then L=""; else L=$PWD/.ab/lib$1.so; fi; env ${L:+CAVOID_LIB=$L} timeout 300 python tools/kbench.py --worlds 8192 --agents $2 --spl 20 64 2>&1 | grep us_per | sed 's/"Gagent.*//' | tr '\n' ' '; echo; }
bn() { echo -n "$1 bench $2: "; if [ "$1" = product ]; then L=""; else L=$PWD/.ab/lib$1.so; fi; env ${L:+CAVOID_LIB=$L} timeout 300 python bench.py $2 --no-cpu-baseline --no-full-loop --no-configs3 --no-pmc --evidence off 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('value %.4e wall_us_per_step %.4f kernel_us %.3f frac %.4f' % (d['value'], d['ms_per_step'] * 1e3, r['kernel_us'], r['frac']))"; }
{
for rep in 1 2 3; do for v in product depth2; do kb $v 4; done; done
for v in product depth2; do kb $v 2; kb $v 3; done
for rep in 1 2 3; do for v in product depth2 old_nt; do bn $v "--steps 20 --warmup 5"; done; done
for v in product depth2; do bn $v ""; done
} > $o/relay_depth2.txt 2>&1
tail -5 $o/tests_product.txt; cat $o/tests_old_nt.txt 2>/dev/null | tail -5; cat $o/soak_depth2.txt; tail -5 $o/tests_depth2.txt; cat $o/relay_depth2.txt
