#!/bin/bash
# Timing-only ablations of the policy pass's GEMM loop (wrong results on purpose): which operand stream the matrix instructions wait for.
#   noW   the weight-fragment buffer loads inside split_gemm3's chunk loop are not issued (stale registers)
#   noA   the activation-fragment LDS reads inside the loop are not issued
#   noWA  neither: the loop is matrix instructions only
# Build (here, CPU box): bash tools/experiments/r06_h_gemm_ablation.sh build    -> .ab/libabl_{base,noW,noA,noWA}.so
# Run (GPU box):         bash tools/experiments/r06_h_gemm_ablation.sh run      -> gpurun_out/r06_h/gemm_ablation.txt
set -e
repo=$(cd $(dirname $0)/../.. && pwd)
if [ "$1" = build ]; then
  mkdir -p $repo/.ab
  for v in base noW noA noWA; do
    d=/tmp/abl_$v; rm -rf $d; mkdir -p $d; cp $repo/rl_collision_avoidance_amd/csrc/* $d/
    python3 - $d/cavoid_policy_split.hpp $v <<'PY'
import sys
p, v = sys.argv[1], sys.argv[2]
s = open(p).read()
a = s.index("__device__ __forceinline__ void split_gemm3(")
b = s.index("template <int P, class RT = SpAllRows>\n__device__ __forceinline__ void split_gemm(")
body = s[a:b]
k = body.index("#pragma unroll 1\n    for (;;) {")
head, loop = body[:k], body[k:]
if "W" in v: loop = loop.replace("split_load_w1(", "ABL_NOP(")
if "A" in v: loop = loop.replace("split_load_a(", "ABL_NOP(").replace("split_load_a_mix(", "ABL_NOP(")
s = s[:a] + "#define ABL_NOP(...) ((void)0)\n" + head + loop + s[b:]
open(p, "w").write(s)
PY
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I$repo/include -I$d -c $d/cavoid_policy_capi.hip -o $d/cavoid_policy_capi.o &
  done
  wait
  for v in base noW noA noWA; do
    objs=$(ls $repo/rl_collision_avoidance_amd/build/*.o | grep -v "policy_capi\|\.ulp\|\.fault\|\.trace")
    hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/abl_$v/cavoid_policy_capi.o -ldl -o $repo/.ab/libabl_$v.so
  done
  ls -la $repo/.ab/
  exit 0
fi
o=$repo/gpurun_out/r06_h; mkdir -p $o
{
for rows in 16384 32768; do for form in quad duo; do for v in base noW noA noWA; do for i in 1 2; do
  echo -n "rows $rows form $form $v: "; CAVOID_LIB=$repo/.ab/libabl_$v.so CAVOID_POLICY_FORM=$form timeout 300 python $repo/tools/polbench.py $rows 3 2>&1 | grep fused_us | sed "s/.*'fused_us': \([0-9.]*\).*/\1 us/"
done; done; done; done
} | tee $o/gemm_ablation.txt
