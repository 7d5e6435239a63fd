# Round 6, the FINAL library (write-through slot stores, 8-slot rings, role order / priorities): 14 parity-stress passes against the float64 oracle on new seeds (offsets 90..103)
# + the randomised bitwise soaks, one box.  usage (GPU box): bash tools/experiments/r06_stress_final.sh
o=gpurun_out/r06_stress_final; mkdir -p $o
timeout 1500 python tests/parity_stress.py $(seq 90 103) > $o/parity_stress_full.log 2>&1; echo "rc=$?" >> $o/parity_stress_full.log
{
echo "Round 6, FINAL library: tests/parity_stress.py 90 .. 103 (14 passes), total line:"; tail -2 $o/parity_stress_full.log
echo "worlds classified (ties / unexplained) over the run:"; grep -c "left the oracle" $o/parity_stress_full.log; grep "left the oracle" $o/parity_stress_full.log | head -12
echo; echo "tools/actor_soak.py 150:"; timeout 500 python tools/actor_soak.py 150 2>&1 | grep -v amdgpu.ids | tail -1
echo; echo "tools/relay_soak.py 150 (ORCA world sets in 30 % of the cases: the pipeline's ORCA instantiation):"; timeout 500 python tools/relay_soak.py 150 2>&1 | grep -v amdgpu.ids | tail -1
echo; echo "tools/policy_soak.py 60:"; timeout 300 python tools/policy_soak.py 60 2>&1 | grep -v amdgpu.ids | tail -1
} > $o/r06_stress_final.txt 2>&1
cat $o/r06_stress_final.txt
