# same-box: (1) the relay-carried GPU tests + a relay soak on the library with the new defaults (consumers priority 1 / loader 0, write-through slot stores),
# (2) kbench + the K = 20 bench line: product vs old_prio (consumers 0 / loader 1) vs old_nt (nt slot stores), (3) timing-only ablations of the loop-carried cycle:
# abl_free (D never waits for a verdict), abl_depth2 (D waits for the verdict one step further back).
o=$PWD/gpurun_out/r06_y; mkdir -p $o
( timeout 600 python -m pytest tests/test_gpu_packed.py tests/test_gpu_parity.py tests/test_gpu_relay_fault.py tests/test_gpu_lookahead.py -x -q 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -4 ) > $o/tests.txt
( timeout 200 python tools/relay_soak.py 45 2>&1 | grep "soak\|MISMATCH" ) > $o/relay_soak.txt
kb() { echo -n "$1: "; if [ "$1" = product ]; then L=""; else L=$PWD/.ab/lib$1.so; fi; env ${L:+CAVOID_LIB=$L} timeout 300 python tools/kbench.py --worlds 8192 --agents 4 --spl 20 64 2>&1 | grep us_per | sed 's/"Gagent.*//' | tr '\n' ' '; echo; }
bn() { echo -n "$1 bench $2: "; if [ "$1" = product ]; then L=""; else L=$PWD/.ab/lib$1.so; fi; env ${L:+CAVOID_LIB=$L} timeout 300 python bench.py $2 --no-cpu-baseline --no-full-loop --no-configs3 --no-pmc --evidence off 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('value %.4e wall_us_per_step %.4f kernel_us %.3f frac %.4f' % (d['value'], d['ms_per_step'] * 1e3, r['kernel_us'], r['frac']))"; }
{
for rep in 1 2; do for v in product old_prio abl_free abl_depth2; do kb $v; done; done
for rep in 1 2; do for v in product old_prio old_nt; do bn $v "--steps 20 --warmup 5"; done; done
for v in product old_prio; do bn $v ""; done
} > $o/relay_depth.txt 2>&1
cat $o/tests.txt $o/relay_soak.txt $o/relay_depth.txt
