#!/bin/bash
# The packed-float32 failure of DESIGN.md 3.7 (d), bisected IN THE FAILING KERNEL (the fused actor kernel at 4 x 8192, two workgroups per
# CU); results: profiles/r05_b_pk_bisect.txt.  `build`: the variant library .ab/libpk_c.so (dev-only N = 4, 10) = the round-4 source left to
# the vectoriser (-DCAVOID_DEV_PKFORM=0) and .ab/libpk_base.so = the product source; the other source-level variants (pk_a, pk_n*, pk_d, pk_m,
# pk_s, pk_sn, pk_w*, pk_i*) need tools/experiments/pk_forms_snippet.hpp dropped back into neighbour_features(); the ISA-level ones are made
# from pk_c's objects by tools/experiments/pk_isa_patch.py.  `run` (on the GPU box): tools/repro_actor_case.py against each library
# (VARIANTS="pk_c pk_isa_mul3_scalar ..."), mismatch counts into gpurun_out/pk_bisect/.
set -e
cd "$(dirname "$0")/../.."
case "$1" in
build)
  XFLAGS="-DCAVOID_DEV_PKFORM=0" tools/mkvariant.sh pk_c ''
  XFLAGS="" tools/mkvariant.sh pk_base ''
  ;;
run)
  out=gpurun_out/pk_bisect; mkdir -p $out
  for v in ${VARIANTS:-pk_base pk_c}; do
    [ -f .ab/lib$v.so ] || continue
    for rep in 1 2 3; do
      echo "== $v rep $rep" | tee -a $out/log.txt
      CAVOID_LIB=$PWD/.ab/lib$v.so timeout 300 python tools/repro_actor_case.py 2 1 1 3 5 8 16 16 16 2>&1 | grep -v amdgpu.ids | tee -a $out/log.txt
    done
  done
  ;;
esac
