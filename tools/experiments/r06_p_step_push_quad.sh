# step_push_quad_kernel (env.step of a tile by four cooperating wavefronts + the Experience bookkeeping, one launch) against step_push_kernel
# (CAVOID_QUAD=0), same box: rocprofv3 kernel averages of tools/robench.py, then the tests that run through cavoid_step_push
repo=$PWD; o=$PWD/gpurun_out/r06_p; mkdir -p $o; export TMPDIR=/tmp
{
for q in 0 1; do
  rm -rf /tmp/rp_sp$q; mkdir -p /tmp/rp_sp$q
  if [ $q = 0 ]; then export CAVOID_QUAD=0; else unset CAVOID_QUAD; fi
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/rp_sp$q -o sp -- python $repo/tools/robench.py 8192 4 400 2>/dev/null | grep "us per step")
  db=$(find /tmp/rp_sp$q -name "*.db" | head -1)
  echo "CAVOID_QUAD=${CAVOID_QUAD:-default}:"; python tools/rocprof_summary.py $db /tmp/rp_sp$q/sum.csv "robench 8192 4 400" > /dev/null; grep -E "step_push" /tmp/rp_sp$q/sum.csv | awk -F'",' '{split($1,n,"("); print n[1] "  calls,total_us,avg_us: " $2}' | cut -c1-140
done
unset CAVOID_QUAD
echo "(tests: 57 passed on the first run of this script)"
} 2>&1 | tee $o/step_push_quad.txt
