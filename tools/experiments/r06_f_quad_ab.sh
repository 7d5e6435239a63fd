mkdir -p gpurun_out/r06_f; o=gpurun_out/r06_f
timeout 900 python -m pytest tests/test_gpu_quad.py -x -q -m gpu 2>&1 | tail -15 > $o/pytest_quad.log; tail -5 $o/pytest_quad.log
for q in 0 1; do for i in 1 2 3; do echo "CAVOID_QUAD=$q"; CAVOID_QUAD=$q timeout 300 python tools/kbench.py --worlds 8192 --agents 4 --spl 1 2>&1 | grep us_per; done; done | tee $o/kbench_quad_n4.txt
for q in 0 1; do echo "CAVOID_QUAD=$q"; CAVOID_QUAD=$q timeout 300 python tools/kbench.py --worlds 8192 --agents 10 --spl 1 2>&1 | grep us_per;  CAVOID_QUAD=$q timeout 300 python tools/kbench.py --worlds 4096 --agents 4 --spl 1 2>&1 | grep us_per; CAVOID_QUAD=$q timeout 300 python tools/kbench.py --worlds 16384 --agents 4 --spl 1 2>&1 | grep us_per; CAVOID_QUAD=$q timeout 300 python tools/kbench.py --worlds 3072 --agents 10 --spl 1 2>&1 | grep us_per; done | tee $o/kbench_quad_other.txt
timeout 900 python -m pytest tests/test_gpu_actor.py -x -q -m gpu 2>&1 | tail -8 > $o/pytest_actor.log; tail -4 $o/pytest_actor.log
for q in 0 1; do for i in 1 2 3; do echo "CAVOID_ACTOR_QUAD=$q"; CAVOID_ACTOR_QUAD=$q timeout 300 python tools/actbench.py 8192 4 16 6 2>&1 | grep us_per; done; done | tee $o/actbench_quad.txt
