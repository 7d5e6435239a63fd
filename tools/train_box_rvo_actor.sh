# the fused actor kernel on worlds with scripted agents: box scenarios generated INSIDE the step (no pool) + a static / non-cooperative /
# ORCA mix -> cavoid_actor_run over the env step's ORCA instantiation (actor_kernel<N, true>)
mkdir -p gpurun_out/r03_train /tmp/ck3
python -m rl_collision_avoidance_amd.ga3c.train --worlds 4096 --scenario box --scripted-fraction 0.4 --static-fraction 0.3 --rvo-fraction 0.4 \
   --pretrain-steps 300 --lr 1e-4 --beta 3e-3 --train-rows 16384 --episodes 20000000 --print-every 500000 --steps-per-graph 8 \
   --checkpoint-dir /tmp/ck3 --save-every 100000000 > gpurun_out/r03_train/train_box_rvo_actor_kernel.txt 2>&1
tail -3 gpurun_out/r03_train/train_box_rvo_actor_kernel.txt
ck=$(ls /tmp/ck3/*.pt | tail -1)
python -m rl_collision_avoidance_amd.ga3c.train --worlds 4096 --scenario box --scripted-fraction 0.4 --static-fraction 0.3 --rvo-fraction 0.4 \
   --load $ck --evaluate 4 >> gpurun_out/r03_train/train_box_rvo_actor_kernel.txt 2>&1
tail -1 gpurun_out/r03_train/train_box_rvo_actor_kernel.txt
