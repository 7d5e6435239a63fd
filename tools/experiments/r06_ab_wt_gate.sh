# same-box: the library with the write-through slot stores gated by tile geometry (whole 128-byte lines), on in every translation unit:
# the whole -m gpu suite, tools/launch_latency.py (incl. bench.py's own bracket on a look-ahead env), and the bench lines of the shapes r06_aa measured.
o=$PWD/gpurun_out/r06_ab; mkdir -p $o
( timeout 1200 python -m pytest tests -m gpu -x -q --tb=short 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -15 ) > $o/pytest_gpu.txt
( timeout 300 python tools/launch_latency.py 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" ) > $o/launch_latency.txt
bn() { echo -n "bench $1: "; timeout 400 python bench.py $1 --no-cpu-baseline --no-full-loop --no-configs3 --no-pmc --evidence off 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('value %.4e wall_us_per_step %.4f kernel_us %.3f frac %.4f %s' % (d['value'], d['ms_per_step'] * 1e3, r['kernel_us'], r['frac'], r['kernel'][:60]))"; }
{
for rep in 1 2; do bn "--steps 20 --warmup 5"; bn "--steps 20 --warmup 5 --scenarios pool"; done
bn ""; bn "--agents 10"; bn "--worlds 65536"; bn "--agents 10 --worlds 262144 --slices 16 --steps 128 --warmup 32"
} > $o/bench_lines.txt 2>&1
cat $o/pytest_gpu.txt $o/launch_latency.txt $o/bench_lines.txt
