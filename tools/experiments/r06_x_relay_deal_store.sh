# same-box A/B of env_relay_kernel variants (tools/mkrelayvariant.sh; .ab/lib<name>.so):
#   ord_plain   role order D P C0 C1 C2 L (round-robin deal)            base = the product's source (order D L P C0 C1 C2, round-robin)
#   dealA / B   plain order + an UNEVEN deal of the observations: C1 (the consumer beside the other tile's P and loader) takes 1/2 (A) or 4/10 (B)
#   dealC / D   the product's order + C1 (the consumer beside D and the other tile's P) takes 1/5 (C) or 1/4 (D)
#   dealA_prio  dealA + consumers at priority 1, loader 0;  prio_c1l0: the same priorities on the product's order and deal
#   share_dist  P hands the N-1 distances to the consumers
#   st_*        cache policy of the per-step-slot output stores: sc1 / sc0 sc1 / sc0 sc1 nt write-through, plain (bench.py's slot form; kbench overwrites one buffer)
o=$PWD/gpurun_out/r06_x; mkdir -p $o
kb() { echo -n "$1: "; CAVOID_LIB=$PWD/.ab/lib$1.so timeout 300 python tools/kbench.py --worlds 8192 --agents 4 --spl 20 64 2>&1 | grep us_per | sed 's/"Gagent.*//' | tr '\n' ' '; echo; }
bn() { echo -n "$1 bench $2: "; CAVOID_LIB=$PWD/.ab/lib$1.so timeout 300 python bench.py $2 --no-cpu-baseline --no-full-loop --no-configs3 --no-pmc --evidence off 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('value %.4e wall_us_per_step %.4f kernel_us %.3f frac %.4f' % (d['value'], d['ms_per_step'] * 1e3, r['kernel_us'], r['frac']))"; }
{
for rep in 1 2; do
  for v in base ord_plain dealA dealB dealC dealD dealA_prio prio_c1l0 share_dist; do kb $v; done
done
for rep in 1 2; do
  for v in base st_sc1 st_sc0sc1 st_sc0sc1nt st_plain share_dist dealA; do bn $v "--steps 20 --warmup 5"; done
done
for v in base st_sc1 st_sc0sc1 dealA; do bn $v ""; done
} > $o/relay_deal_store.txt 2>&1
cat $o/relay_deal_store.txt
