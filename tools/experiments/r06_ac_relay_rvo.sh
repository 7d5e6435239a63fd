# env_relay_kernel<N, true> (ORCA agents in the role-split K-step loop): the new test, the relay-carried suites, the soak with ORCA cases, and the training-mix rate
o=$PWD/gpurun_out/r06_ac; mkdir -p $o
flt() { grep -v "amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl"; }
( timeout 900 python -m pytest tests/test_gpu_packed.py -x -q --tb=short -k "orca" 2>&1 | flt | tail -30 ) > $o/test_orca.txt
( timeout 900 python -m pytest tests/test_gpu_packed.py tests/test_gpu_parity.py tests/test_gpu_relay_fault.py tests/test_gpu_lookahead.py -x -q --tb=short 2>&1 | flt | tail -15 ) > $o/tests.txt
( RELAY_SOAK_RVO_P=0.6 timeout 300 python tools/relay_soak.py 90 2>&1 | grep "soak\|MISMATCH" ) > $o/relay_soak.txt
( timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-full-loop --no-configs3 --no-pmc 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('K=20 value %.4e wall_us_per_step %.4f kernel_us %.3f frac %.4f' % (d['value'], d['ms_per_step'] * 1e3, r['kernel_us'], r['frac']))
print(json.dumps(d['extra'].get('scenario_sources'), indent=1)[:3000])" ) > $o/bench_k20.txt 2>&1
( timeout 600 python bench.py --no-cpu-baseline --no-full-loop --no-configs3 --no-pmc 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('default value %.4e wall_us_per_step %.4f kernel_us %.3f frac %.4f' % (d['value'], d['ms_per_step'] * 1e3, r['kernel_us'], r['frac']))
print(json.dumps(d['extra'].get('scenario_sources'), indent=1)[:3000])" ) > $o/bench_default.txt 2>&1
cat $o/test_orca.txt; tail -4 $o/tests.txt; cat $o/relay_soak.txt $o/bench_k20.txt $o/bench_default.txt
