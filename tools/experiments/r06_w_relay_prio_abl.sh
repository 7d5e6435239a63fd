# same-box session: (1) pytest -m gpu + relay soak on HEAD's product library, (2) host side of the driver's 20-step launch, (3) A/B of env_relay_kernel's
# role priorities (CAVOID_RELAY_PRIO_*) and timing-only ablations of its roles (CAVOID_RELAY_ABL: wrong results, what bounds the period).
# variants: tools/mkrelayvariant.sh <name> -D...   (built on the build box into .ab/)
o=$PWD/gpurun_out/r06_w; mkdir -p $o
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 ) > $o/pytest_gpu.txt
( timeout 200 python tools/relay_soak.py 60 2>&1 | tail -3 ) > $o/relay_soak.txt
( timeout 200 python tools/launch_latency.py 2>&1 | grep -v amdgpu.ids ) > $o/launch_latency.txt
{
for rep in 1 2; do
  echo -n "product: "; timeout 300 python tools/kbench.py --worlds 8192 --agents 4 --spl 20 64 2>&1 | grep us_per | tr '\n' ' '; echo
  for v in prio_p2 prio_d2 prio_c1l0 prio_flat abl_c abl_p abl_d abl_cp abl_cpd; do
    echo -n "$v: "; CAVOID_LIB=$PWD/.ab/lib$v.so timeout 300 python tools/kbench.py --worlds 8192 --agents 4 --spl 20 64 2>&1 | grep us_per | tr '\n' ' '; echo
  done
done
} > $o/relay_prio_abl.txt 2>&1
cat $o/pytest_gpu.txt $o/relay_soak.txt $o/launch_latency.txt $o/relay_prio_abl.txt
