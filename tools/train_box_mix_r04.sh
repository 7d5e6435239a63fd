# round 4: the reference's training mix -- box scenarios generated inside the step, static / RVO / frozen-network / non-cooperative agents
# around the learners -- through the FUSED actor kernel (cavoid_actor_run_mix; round 3 ran this mix as one launch per phase), float16-split
# inference.  usage (GPU box): bash tools/train_box_mix_r04.sh
mkdir -p gpurun_out/r04_train /tmp/ck4
python -m rl_collision_avoidance_amd.ga3c.train --worlds 4096 --scenario box --scripted-fraction 0.4 --static-fraction 0.3 --rvo-fraction 0.3 \
   --frozen-fraction 0.2 --pretrain-steps 300 --lr 1e-4 --beta 3e-3 --train-rows 16384 --episodes 20000000 --print-every 500000 --steps-per-graph 8 \
   --checkpoint-dir /tmp/ck4 --save-every 100000000 > gpurun_out/r04_train/train_box_mix.txt 2>&1
grep -m1 "^actors:" gpurun_out/r04_train/train_box_mix.txt; tail -2 gpurun_out/r04_train/train_box_mix.txt
ck=$(ls /tmp/ck4/*.pt | tail -1)
python -m rl_collision_avoidance_amd.ga3c.train --worlds 4096 --scenario box --scripted-fraction 0.4 --static-fraction 0.3 --rvo-fraction 0.3 \
   --frozen-fraction 0.2 --load $ck --evaluate 4 >> gpurun_out/r04_train/train_box_mix.txt 2>&1
tail -1 gpurun_out/r04_train/train_box_mix.txt
