bash tools/gpu_round.sh r06_f > gpurun_out/r06_f_round.log 2>&1
bash tools/profile_relay.sh r06_f > gpurun_out/r06_f_profile.log 2>&1
tail -30 gpurun_out/r06_f_round.log; tail -40 gpurun_out/r06_f_profile.log
