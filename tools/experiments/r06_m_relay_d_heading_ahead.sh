# same-box A/B of env_relay_kernel: base = before round 6's changes of the state-owner wavefront (.ab/libabl_base.so), early = its LDS reads issued early
# (.ab/librelay_early.so), product = + the heading half of the advance one step ahead
o=$PWD/gpurun_out/r06_m; mkdir -p $o
{
for rep in 1 2 3; do
  for v in abl_base relay_early; do echo -n "$v: "; CAVOID_LIB=$PWD/.ab/lib$v.so timeout 300 python tools/kbench.py --worlds 8192 --agents 4 --spl 20 64 2>&1 | grep us_per | tr '\n' ' '; echo; done
  echo -n "product: "; timeout 300 python tools/kbench.py --worlds 8192 --agents 4 --spl 20 64 2>&1 | grep us_per | tr '\n' ' '; echo
done
for n in 2 3 5 6; do
  echo -n "N=$n early:   "; CAVOID_LIB=$PWD/.ab/librelay_early.so timeout 300 python tools/kbench.py --worlds 8192 --agents $n --spl 20 64 2>&1 | grep us_per | tr '\n' ' '; echo
  echo -n "N=$n product: "; timeout 300 python tools/kbench.py --worlds 8192 --agents $n --spl 20 64 2>&1 | grep us_per | tr '\n' ' '; echo
done
echo "== bit-identity / protocol tests on the product library"
timeout 1200 python -m pytest tests/test_gpu_packed.py tests/test_gpu_lookahead.py tests/test_gpu_relay_fault.py tests/test_gpu_parity.py tests/test_gpu_cfg_fields.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python tools/relay_soak.py 120 2>&1 | grep -v amdgpu.ids | tail -2
} | tee $o/relay_d_heading_ahead.txt
