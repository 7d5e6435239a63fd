#!/bin/bash
# Does the in-wavefront matrix / vector interleave of the PIPE form work once it has registers?  The committed PIPE kernel spills 92 registers at the
# 256-register budget of two workgroups per CU; here it is built with __launch_bounds__(256, 1) (512 registers, one workgroup per CU).
#   build (CPU box): bash tools/experiments/r06_i_pipe512.sh build  -> .ab/libpipe512.so      run (GPU box): ... run
set -e
repo=$(cd $(dirname $0)/../.. && pwd)
if [ "$1" = build ]; then
  d=/tmp/pipe512; rm -rf $d; mkdir -p $d; cp $repo/rl_collision_avoidance_amd/csrc/* $d/
  sed -i 's/__global__ void __launch_bounds__(256, 2) policy_forward_split_kernel/__global__ void __launch_bounds__(256, PIPE ? 1 : 2) policy_forward_split_kernel/' $d/cavoid_policy_split.hpp
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I$repo/include -I$d -c $d/cavoid_policy_capi.hip -o $d/cavoid_policy_capi.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A12 "policy_forward_split_kernelILi16ELb1" | head -16
  objs=$(ls $repo/rl_collision_avoidance_amd/build/*.o | grep -v "policy_capi\|\.ulp\|\.fault\|\.trace")
  hipcc --offload-arch=gfx950 -shared -fPIC $objs $d/cavoid_policy_capi.o -ldl -o $repo/.ab/libpipe512.so
  exit 0
fi
o=$repo/gpurun_out/r06_i; mkdir -p $o
{
for rows in 16384 32768; do for spec in "abl_base quad" "abl_base pipe" "pipe512 pipe"; do set -- $spec; for i in 1 2; do
  echo -n "rows $rows lib $1 form $2: "; CAVOID_LIB=$repo/.ab/lib$1.so CAVOID_POLICY_FORM=$2 timeout 300 python $repo/tools/polbench.py $rows 3 2>&1 | grep fused_us | sed "s/.*'fused_us': \([0-9.]*\).*/\1 us/"
done; done; done
} | tee $o/pipe512.txt
