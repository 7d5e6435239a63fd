# same-box A/B: which wavefront index of env_relay_kernel's workgroup carries which role (wavefront w of a workgroup lands on SIMD w % 4 -- tools/ubench/wave_placement.hip):
#   product  D, P, C0, C1, C2, L       ord1  D, P, C0, C1, L, C2 (the light loader beside D)       ord2  P, D, C0, ... (D and P swapped)
o=$PWD/gpurun_out/r06_n; mkdir -p $o
{
for rep in 1 2 3; do
  echo -n "product: "; timeout 300 python tools/kbench.py --worlds 8192 --agents 4 --spl 20 64 2>&1 | grep us_per | tr '\n' ' '; echo
  for v in relay_ord1 relay_ord2; do echo -n "$v: "; CAVOID_LIB=$PWD/.ab/lib$v.so timeout 300 python tools/kbench.py --worlds 8192 --agents 4 --spl 20 64 2>&1 | grep us_per | tr '\n' ' '; echo; done
done
} | tee $o/relay_role_order.txt
