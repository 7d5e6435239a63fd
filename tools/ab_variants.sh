mkdir -p gpurun_out/r03e
for rep in 1 2; do
  (cd .ab/r02 && python tools/kbench.py --worlds 8192 --agents 4 --spl 1) 2>&1 | grep W | sed 's/^/r02 /'
  for v in v0 v4 v5 v6; do
    CAVOID_LIB=$PWD/.ab/lib$v.so python tools/kbench.py --worlds 8192 --agents 4 --spl 1 2>&1 | grep W | sed "s/^/$v /"
  done
done | tee gpurun_out/r03e/variants.log
for v in v0 v6; do CAVOID_LIB=$PWD/.ab/lib$v.so python tools/kbench.py --worlds 262144 --agents 10 --spl 1 --steps 100 2>&1 | grep W | sed "s/^/$v sat /"; done | tee -a gpurun_out/r03e/variants.log
(cd .ab/r02 && python tools/kbench.py --worlds 262144 --agents 10 --spl 1 --steps 100) 2>&1 | grep W | sed 's/^/r02 sat /' | tee -a gpurun_out/r03e/variants.log
timeout 2000 python -m pytest tests -x -q -m gpu > gpurun_out/r03e/pytest.log 2>&1; tail -15 gpurun_out/r03e/pytest.log
