# same-box: (1) tools/launch_latency.py (an isolated launch against the same launch in a train; idle gaps); (2) write-through (sc1) slot stores in the OTHER K-step
# forms (cavoid_multistep.hip: env_kernel<N, MODE_STEP_AUTORESET_N / _PF>, env_pipe_kernel): configs[3] 10 x 8192 (64-step launches), 4 x 65536 (1024+ tiles:
# the pipeline / single-wavefront loops) and the saturated shapes, bench.py's slot form.   ms_sc1 = TU=cavoid_multistep tools/mkrelayvariant.sh ms_sc1 -DCAVOID_STREAM_POLICY=1
o=$PWD/gpurun_out/r06_aa; mkdir -p $o
( timeout 300 python tools/launch_latency.py 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" ) > $o/launch_latency.txt
bn() { echo -n "$1 bench $2: "; if [ "$1" = product ]; then L=""; else L=$PWD/.ab/lib$1.so; fi; env ${L:+CAVOID_LIB=$L} timeout 400 python bench.py $2 --no-cpu-baseline --no-full-loop --no-configs3 --no-pmc --evidence off 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('value %.4e wall_us_per_step %.4f kernel_us %.3f frac %.4f %s' % (d['value'], d['ms_per_step'] * 1e3, r['kernel_us'], r['frac'], r['kernel'][:60]))"; }
{
for rep in 1 2; do
  for v in product ms_sc1; do
    bn $v "--agents 10"
    bn $v "--worlds 65536"
    bn $v "--worlds 1048576 --slices 16 --steps 128 --warmup 32"
    bn $v "--agents 10 --worlds 262144 --slices 16 --steps 128 --warmup 32"
  done
done
} > $o/store_policy_other_forms.txt 2>&1
cat $o/launch_latency.txt $o/store_policy_other_forms.txt
