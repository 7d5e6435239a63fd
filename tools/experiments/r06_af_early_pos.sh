# same-box A/B: D posts the positions of the state it is making as soon as they exist (CAVOID_RELAY_EARLY_POS=1: P's distance loop starts under the rest of D's advance).
# correctness on the variant (relay soak + the relay-carried tests), then kbench and bench.py's K = 20 form, interleaved.
o=$PWD/gpurun_out/r06_af; mkdir -p $o
flt() { grep -av "amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl"; }
{
echo "== correctness on early1 =="
CAVOID_LIB=$PWD/.ab/libearly1.so timeout 200 python tools/relay_soak.py 60 2>&1 | grep -a "soak\|MISMATCH"
CAVOID_LIB=$PWD/.ab/libearly1.so timeout 600 python -m pytest tests/test_gpu_packed.py tests/test_gpu_parity.py tests/test_gpu_lookahead.py -x -q 2>&1 | flt | tail -3
echo "== timing =="
kb() { echo -n "$1: "; CAVOID_LIB=$PWD/.ab/lib$1.so timeout 300 python tools/kbench.py --worlds 8192 --agents $2 --spl 20 64 2>&1 | grep us_per | sed 's/"Gagent.*//' | tr '\n' ' '; echo; }
bn() { echo -n "$1 bench $2: "; CAVOID_LIB=$PWD/.ab/lib$1.so timeout 300 python bench.py $2 --no-cpu-baseline --no-full-loop --no-configs3 --no-pmc --no-fresh-scenarios --evidence off 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('value %.4e wall_us_per_step %.4f kernel_us %.3f frac %.4f' % (d['value'], d['ms_per_step'] * 1e3, r['kernel_us'], r['frac']))"; }
for rep in 1 2 3; do for v in base0 early1; do kb $v 4; done; done
for v in base0 early1; do kb $v 2; kb $v 3; done
for rep in 1 2 3; do for v in base0 early1; do bn $v "--steps 20 --warmup 5"; done; done
for v in base0 early1; do bn $v ""; done
} > $o/early_pos.txt 2>&1
cat $o/early_pos.txt
