#!/bin/bash
# what the parity harness (tests/parity_stress.py + tests/replay.py::classify_divergence) says about ONE ulp-scale arithmetic fault in the kernels
# (python -m rl_collision_avoidance_amd.build --ulp-faults): usage: ulp_fault_probe.sh "<kinds>" ; results: profiles/r05_d_ulp_fault_probe.txt
for k in ${1:-1 2 3 4}; do for c in '[4, 2048, 300, 700, 0.6, 0, 1, 0.5, 1]' '[10, 512, 256, 800, 0.5, 1, 1, 0.5, 8]' '[4, 2048, 320, 1011, 0.5, 0, 1, 0.5, 16, true, 0]' '[4, 4096, 300, 3, 0.0, 0]'; do echo "== fault $k case $c"; CAVOID_LIB=$PWD/tests/_variants/libcavoid_hip_ulp$k.so timeout 600 python tests/parity_stress.py --one "$c" 2>&1 | grep -v amdgpu.ids | tail -6 | cut -c1-400; done; done
