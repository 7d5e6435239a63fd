# Round 6, final library: 30 parity-stress passes (seed offsets 60..89) + long soaks, one box.  usage (GPU box): bash tools/experiments/r06_stress_long.sh
o=gpurun_out/r06_stress; mkdir -p $o
timeout 2400 python tests/parity_stress.py $(seq 60 89) > $o/parity_stress_full.log 2>&1; echo "rc=$?" >> $o/parity_stress_full.log
{
echo "Round 6, final library (env_relay_kernel with the early LDS reads): tests/parity_stress.py 60 .. 89 (30 passes), total line:"; tail -2 $o/parity_stress_full.log
echo "worlds classified (ties / unexplained) over the run:"; grep -c "left the oracle" $o/parity_stress_full.log
echo; echo "tools/actor_soak.py 300:"; timeout 700 python tools/actor_soak.py 300 2>&1 | grep -v amdgpu.ids | tail -1
echo; echo "tools/relay_soak.py 300:"; timeout 700 python tools/relay_soak.py 300 2>&1 | grep -v amdgpu.ids | tail -1
} > $o/r06_stress_long.txt 2>&1
cat $o/r06_stress_long.txt
