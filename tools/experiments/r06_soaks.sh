# Round 6, final library: the randomised soaks + one full parity-stress pass (seed offset 6) on one box.  usage (GPU box): bash tools/experiments/r06_soaks.sh
o=gpurun_out/r06_soaks; mkdir -p $o
{
echo "Round 6, the round's library (U5 = +0.5 default, env_quad_kernel for one-step launches, the duo policy pass, the cooperative env step inside the actor kernel,"
echo "continuous / holonomic actions in every launch form): the randomised soaks and one full parity-stress pass (seed offset 6), one box"
echo; echo "tools/actor_soak.py 240:"; timeout 600 python tools/actor_soak.py 240 2>&1 | grep -v amdgpu.ids | tail -2
echo; echo "tools/relay_soak.py 120:"; timeout 400 python tools/relay_soak.py 120 2>&1 | grep -v amdgpu.ids | tail -2
echo; echo "tools/policy_soak.py 120:"; timeout 400 python tools/policy_soak.py 120 2>&1 | grep -v amdgpu.ids | tail -2
echo; echo "tests/parity_stress.py 6 (last case + total):"; timeout 1500 python tests/parity_stress.py 6 2>&1 | grep -v amdgpu.ids | tail -2
} > $o/r06_soaks.txt 2>&1
tail -20 $o/r06_soaks.txt
