# env_relay_kernel<N, true> (ORCA agents in the role-split K-step loop): the new test (under rocprofv3: which kernels it really launches), the relay-carried suites,
# the soak with ORCA cases, and the training-mix rate
o=$PWD/gpurun_out/r06_ac; mkdir -p $o; export TMPDIR=/tmp
flt() { grep -av "amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl"; }
repo=$PWD
rm -rf /tmp/rp_orca; mkdir -p /tmp/rp_orca
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/rp_orca -o orca -- python -m pytest $repo/tests/test_gpu_packed.py -x -q --tb=short -k "orca" -p no:cacheprovider 2>&1 | flt | tail -30 ) > $o/test_orca.txt
db=$(find /tmp/rp_orca -name "*.db" | head -1); [ -n "$db" ] && python tools/rocprof_summary.py $db $o/test_orca_kernels.csv "rocprofv3 --kernel-trace --stats -- python -m pytest tests/test_gpu_packed.py -k orca" > /dev/null
( timeout 900 python -m pytest tests/test_gpu_packed.py tests/test_gpu_parity.py tests/test_gpu_relay_fault.py tests/test_gpu_lookahead.py tests/test_gpu_actor.py -x -q --tb=short 2>&1 | flt | tail -15 ) > $o/tests.txt
( RELAY_SOAK_RVO_P=0.6 timeout 300 python tools/relay_soak.py 90 2>&1 | grep -a "soak\|MISMATCH" ) > $o/relay_soak.txt
for a in "--steps 20 --warmup 5" ""; do
( timeout 600 python bench.py $a --no-cpu-baseline --no-full-loop --no-configs3 --no-pmc 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('bench $a: value %.4e wall_us_per_step %.4f kernel_us %.3f frac %.4f' % (d['value'], d['ms_per_step'] * 1e3, r['kernel_us'], r['frac']))
for k, v in d['extra'].get('scenario_sources', {}).items():
    if isinstance(v, dict) and 'value' in v: print('   %-32s %.3e agent-steps/s  %.3f us per step by wall clock, kernel %.3f us per step, frac %.3f  %s' % (k, v['value'], v['ms_per_step'] * 1e3, v['roofline']['kernel_us_per_step'], v['roofline']['frac'], v['roofline']['kernel'][:48]))
    elif isinstance(v, dict): print('   ', k, v)" ) >> $o/bench_lines.txt 2>&1
done
cat $o/test_orca.txt; cut -c1-120 $o/test_orca_kernels.csv | head -12; tail -4 $o/tests.txt; cat $o/relay_soak.txt $o/bench_lines.txt
