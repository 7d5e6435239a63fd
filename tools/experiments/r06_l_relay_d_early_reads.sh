# same-box A/B: env_relay_kernel with D's LDS reads issued early (the product library) against the library before that change (.ab/libabl_base.so)
o=$PWD/gpurun_out/r06_l; mkdir -p $o
{
for rep in 1 2 3; do
  echo -n "before: "; CAVOID_LIB=$PWD/.ab/libabl_base.so timeout 300 python tools/kbench.py --worlds 8192 --agents 4 --spl 20 64 2>&1 | grep us_per | tr '\n' ' '; echo
  echo -n "after:  "; timeout 300 python tools/kbench.py --worlds 8192 --agents 4 --spl 20 64 2>&1 | grep us_per | tr '\n' ' '; echo
done
for n in 2 3 5 6; do
  echo -n "N=$n before: "; CAVOID_LIB=$PWD/.ab/libabl_base.so timeout 300 python tools/kbench.py --worlds 8192 --agents $n --spl 20 64 2>&1 | grep us_per | tr '\n' ' '; echo
  echo -n "N=$n after:  "; timeout 300 python tools/kbench.py --worlds 8192 --agents $n --spl 20 64 2>&1 | grep us_per | tr '\n' ' '; echo
done
echo "== bit-identity / protocol tests on the new library"
timeout 1200 python -m pytest tests/test_gpu_packed.py tests/test_gpu_lookahead.py tests/test_gpu_relay_fault.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python tools/relay_soak.py 90 2>&1 | grep -v amdgpu.ids | tail -2
} | tee $o/relay_d_early_reads.txt
