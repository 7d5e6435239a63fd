# training evidence for SURVEY section 8f-N3 reached from the loop: box scenarios + a scripted mix (static / RVO / frozen-network / non-cooperative)
mkdir -p gpurun_out/r03_train /tmp/ck
python -m rl_collision_avoidance_amd.ga3c.train --worlds 4096 --scenario box --scripted-fraction 0.4 --static-fraction 0.3 --rvo-fraction 0.3 \
   --frozen-fraction 0.2 --pretrain-steps 300 --lr 1e-4 --beta 3e-3 --train-rows 16384 --episodes 20000000 --print-every 500000 \
   --checkpoint-dir /tmp/ck --save-every 100000000 > gpurun_out/r03_train/train_box_mix.txt 2>&1
tail -3 gpurun_out/r03_train/train_box_mix.txt
ck=$(ls /tmp/ck/*.pt | tail -1)
python -m rl_collision_avoidance_amd.ga3c.train --worlds 4096 --scenario box --scripted-fraction 0.4 --static-fraction 0.3 --rvo-fraction 0.3 \
   --frozen-fraction 0.2 --load $ck --evaluate 4 >> gpurun_out/r03_train/train_box_mix.txt 2>&1
tail -1 gpurun_out/r03_train/train_box_mix.txt
# the fused actor kernel in the training loop (all-learner box scenarios from a pool -> cavoid_actor_run)
python -m rl_collision_avoidance_amd.ga3c.train --worlds 4096 --scenario box --scenario-pool 65536 --pretrain-steps 300 --lr 1e-4 --beta 3e-3 \
   --train-rows 16384 --episodes 20000000 --print-every 500000 --steps-per-graph 8 --checkpoint-dir /tmp/ck2 --save-every 100000000 > gpurun_out/r03_train/train_box_actor_kernel.txt 2>&1
tail -3 gpurun_out/r03_train/train_box_actor_kernel.txt
ck=$(ls /tmp/ck2/*.pt | tail -1)
python -m rl_collision_avoidance_amd.ga3c.train --worlds 4096 --scenario box --scenario-pool 65536 --load $ck --evaluate 4 >> gpurun_out/r03_train/train_box_actor_kernel.txt 2>&1
tail -1 gpurun_out/r03_train/train_box_actor_kernel.txt
