#!/bin/bash
# The packed-float32 operand-swap hazard of DESIGN.md 3.7 (d), bisected IN THE FAILING KERNEL (the fused actor kernel at 4 x 8192,
# two workgroups per CU).  `build`: variant libraries .ab/libpk_*.so (dev-only N = 4, 10) from -DCAVOID_DEV_PKFORM switches in
# neighbour_features() (csrc/cavoid_kernels.hpp); `run` (on the GPU box): tools/repro_actor_case.py against each, mismatch counts
# into gpurun_out/pk_bisect/.
#   pk_c      the round-4 source left to the vectoriser (v_pk_mul_f32 + v_pk_fma_f32 op_sel:[0,0,1])
#   pk_a      the same pair spelled out in one asm block, nothing between the two instructions
#   pk_n0/1/3/7  ... with s_nop 0 / 1 / 3 / 7 between them
#   pk_d      ... with s_waitcnt lgkmcnt(0) in front (LDS reads drained)
#   pk_m      the same arithmetic with the swap made by two v_mov_b32: a packed fma WITHOUT op_sel
set -e
cd "$(dirname "$0")/../.."
case "$1" in
build)
  XFLAGS="-DCAVOID_DEV_PKFORM=0" tools/mkvariant.sh pk_c ''
  XFLAGS="-DCAVOID_DEV_PKFORM=1 -DCAVOID_DEV_PKNOPS=-1" tools/mkvariant.sh pk_a ''
  for n in 0 1 3 7; do XFLAGS="-DCAVOID_DEV_PKFORM=1 -DCAVOID_DEV_PKNOPS=$n" tools/mkvariant.sh pk_n$n ''; done
  XFLAGS="-DCAVOID_DEV_PKFORM=1 -DCAVOID_DEV_PKNOPS=-1 -DCAVOID_DEV_PKDRAIN" tools/mkvariant.sh pk_d ''
  XFLAGS="-DCAVOID_DEV_PKFORM=2" tools/mkvariant.sh pk_m ''
  XFLAGS="" tools/mkvariant.sh pk_base ''
  ;;
run)
  out=gpurun_out/pk_bisect; mkdir -p $out
  for v in ${VARIANTS:-pk_base pk_c pk_a pk_n0 pk_n1 pk_n3 pk_n7 pk_d pk_m pk_s pk_sn pk_cz}; do
    [ -f .ab/lib$v.so ] || continue
    for rep in 1 2 3; do
      echo "== $v rep $rep" | tee -a $out/log.txt
      CAVOID_LIB=$PWD/.ab/lib$v.so timeout 300 python tools/repro_actor_case.py 2 1 1 3 5 8 16 16 16 2>&1 | grep -v amdgpu.ids | tee -a $out/log.txt
    done
  done
  ;;
esac
