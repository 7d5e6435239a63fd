# Round 6, the FINAL library, more of tools/experiments/r06_stress_final.sh: 30 parity-stress passes against the float64 oracle on seeds 104..133 + a relay soak.
o=gpurun_out/r06_stress_final2; mkdir -p $o
timeout 2000 python tests/parity_stress.py $(seq 104 133) > $o/parity_stress_full.log 2>&1; echo "rc=$?" >> $o/parity_stress_full.log
{
echo "Round 6, FINAL library: tests/parity_stress.py 104 .. 133 (30 passes), total line:"; tail -2 $o/parity_stress_full.log
echo "worlds classified (ties / unexplained) over the run:"; grep -c "left the oracle" $o/parity_stress_full.log; grep "left the oracle" $o/parity_stress_full.log | sort | uniq -c | head -20
echo; echo "tools/relay_soak.py 120:"; timeout 400 python tools/relay_soak.py 120 2>&1 | grep -v amdgpu.ids | tail -1
} > $o/r06_stress_final2.txt 2>&1
cat $o/r06_stress_final2.txt
