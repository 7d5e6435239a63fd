# same-box A/B of env_relay_kernel: HEAD before the H wavefront (.ab/librelay_early.so) against the product with it (the heading half of D's advance made one
# step ahead on a wavefront of its own), then the bit-identity / protocol tests and a soak on the product.  usage (GPU box): bash tools/experiments/r06_o_relay_heading_wavefront.sh
o=$PWD/gpurun_out/r06_o; mkdir -p $o
{
echo "== quick protocol check first (a hang here must not cost the box: everything under timeout)"
timeout 120 python tools/kbench.py --worlds 8192 --agents 4 --spl 20 2>&1 | grep us_per || echo "KBENCH FAILED rc=$?"
timeout 600 python -m pytest tests/test_gpu_packed.py tests/test_gpu_lookahead.py tests/test_gpu_relay_fault.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2 3; do
  echo -n "before:  "; CAVOID_LIB=$PWD/.ab/librelay_early.so timeout 300 python tools/kbench.py --worlds 8192 --agents 4 --spl 20 64 2>&1 | grep us_per | tr '\n' ' '; echo
  echo -n "product: "; timeout 300 python tools/kbench.py --worlds 8192 --agents 4 --spl 20 64 2>&1 | grep us_per | tr '\n' ' '; echo
done
for n in 2 3 5 6; do
  echo -n "N=$n before:  "; CAVOID_LIB=$PWD/.ab/librelay_early.so timeout 300 python tools/kbench.py --worlds 8192 --agents $n --spl 20 64 2>&1 | grep us_per | tr '\n' ' '; echo
  echo -n "N=$n product: "; timeout 300 python tools/kbench.py --worlds 8192 --agents $n --spl 20 64 2>&1 | grep us_per | tr '\n' ' '; echo
done
echo "== more tests + soak on the product library"
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cfg_fields.py tests/test_gpu_kat.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python tools/relay_soak.py 120 2>&1 | grep -v amdgpu.ids | tail -2
} 2>&1 | tee $o/relay_heading_wavefront.txt
