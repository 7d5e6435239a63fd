#!/bin/bash
# same-box A/B: the env step's wavefront on SIMD 0 of both of a CU's workgroups (envw0, rounds 1-4) vs on SIMD 0 / SIMD 2 by arrival parity (envrot)
for rep in 1 2 3; do for v in envw0 envrot; do echo -n "$v "; CAVOID_LIB=$PWD/.ab/lib$v.so python tools/actbench.py 8192 4 16 20 2>&1 | grep -v amdgpu.ids; done; done
for v in envw0 envrot; do echo -n "$v N=10 "; CAVOID_LIB=$PWD/.ab/lib$v.so python tools/actbench.py 8192 10 16 10 2>&1 | grep -v amdgpu.ids; done
