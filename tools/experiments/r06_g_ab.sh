# same-box A/Bs of round 6's restructurings: the actor kernel with / without the cooperative (four-wavefront) env step, the stand-alone
# policy pass in its four forms.  usage (GPU box): bash tools/experiments/r06_g_ab.sh
o=gpurun_out/r06_g; mkdir -p $o
for q in 0 1; do for i in 1 2 3; do echo "CAVOID_ACTOR_QUAD=$q"; CAVOID_ACTOR_QUAD=$q timeout 300 python tools/actbench.py 8192 4 16 6 2>&1 | grep us_per; done; done | tee $o/actbench_quad.txt
for q in 0 1; do echo "CAVOID_ACTOR_QUAD=$q (32-step launches)"; CAVOID_ACTOR_QUAD=$q timeout 300 python tools/actbench.py 8192 4 32 6 2>&1 | grep us_per; done | tee -a $o/actbench_quad.txt
for f in quad duo oct pipe; do for i in 1 2; do echo "CAVOID_POLICY_FORM=$f"; CAVOID_POLICY_FORM=$f timeout 300 python tools/polbench.py 32768 3 2>&1 | grep -v amdgpu.ids | tail -2; done; done | tee $o/polbench_forms.txt
