mkdir -p gpurun_out/r03_train
common="--worlds 2048 --agents 10 --pretrain-steps 300 --lr 1e-4 --beta 3e-3 --train-rows 16384 --episodes 1500000 --print-every 250000 --save-every 100000000 --steps-per-graph 4"
for seed in 1 2 3; do
  timeout 900 python -m rl_collision_avoidance_amd.ga3c.train $common --seed $seed > gpurun_out/r03_train/n10_actor_s$seed.txt 2>&1
  echo "actor kernel seed $seed: $(grep -o 'RScore: *[-0-9.]*' gpurun_out/r03_train/n10_actor_s$seed.txt | tr -s ' ' | cut -d' ' -f2 | tr '\n' ' ') $(grep -c 'fused actor' gpurun_out/r03_train/n10_actor_s$seed.txt)"
done
