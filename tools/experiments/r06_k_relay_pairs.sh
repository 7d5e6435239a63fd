# same-box A/B of the pair-pass wavefront's loop in env_relay_kernel: base (abl_base = HEAD), A = every neighbour record read before the first chain,
# B = A + the N-1 square-root chains in lockstep (stage by stage).  usage (GPU box): bash tools/experiments/r06_k_relay_pairs.sh
o=$PWD/gpurun_out/r06_k; mkdir -p $o
{
for rep in 1 2 3; do for v in abl_base relayA relayB; do
  echo -n "$v: "; CAVOID_LIB=$PWD/.ab/lib$v.so timeout 300 python tools/kbench.py --worlds 8192 --agents 4 --spl 20 64 2>&1 | grep us_per | tr '\n' ' '; echo
done; done
for v in relayA relayB; do echo "== $v: bit-identity tests"; CAVOID_LIB=$PWD/.ab/lib$v.so timeout 900 python -m pytest tests/test_gpu_packed.py tests/test_gpu_lookahead.py -x -q -m gpu 2>&1 | tail -2; done
} | tee $o/relay_pairs_ab.txt
