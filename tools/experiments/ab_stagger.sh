#!/bin/bash
# same-box A/B: the CU's second workgroup delayed by n x s_sleep(127) (~4 us each) at the start of the fused actor kernel
for rep in 1 2; do for v in ${VARIANTS:-stag0 stag2 stag4 stag6 stag9}; do echo -n "$v "; CAVOID_LIB=$PWD/.ab/lib$v.so python tools/actbench.py 8192 4 ${STEPS:-32} 12 2>&1 | grep -v amdgpu.ids; done; done
