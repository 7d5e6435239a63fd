# does N = 10 training learn equally well with 3 and with 5 partial products in the inference kernel?  (round 1, float32 inference:
# rolling reward 0.89 after 1 M episodes with these options; the take-off is abrupt and its timing varies from run to run)
mkdir -p gpurun_out/r03_train
common="--worlds 2048 --agents 10 --pretrain-steps 300 --lr 1e-4 --beta 3e-3 --train-rows 16384 --episodes 1500000 --print-every 250000 --save-every 100000000 --steps-per-graph 4 --no-actor-kernel"
for seed in 1 2 3; do for p in 3 5; do
  CAVOID_POLICY_PRODUCTS=$p timeout 900 python -m rl_collision_avoidance_amd.ga3c.train $common --seed $seed > gpurun_out/r03_train/n10_p${p}_s$seed.txt 2>&1
  echo "products $p seed $seed: $(grep -o 'RScore: *[-0-9.]*' gpurun_out/r03_train/n10_p${p}_s$seed.txt | tr -s ' ' | cut -d' ' -f2 | tr '\n' ' ')"
done; done
