mkdir -p gpurun_out/r03g
for rep in 1 2; do
  (cd .ab/r02 && python tools/kbench.py --worlds 8192 --agents 4 --spl 1) 2>&1 | grep W | sed 's/^/r02 /'
  for v in v7 v8 v10; do
    CAVOID_LIB=$PWD/.ab/lib$v.so python tools/kbench.py --worlds 8192 --agents 4 --spl 1 2>&1 | grep W | sed "s/^/$v /"
  done
  python tools/kbench.py --worlds 8192 --agents 4 --spl 1 32 2>&1 | grep W | sed "s/^/main /"
  python tools/kbench.py --worlds 8192 --agents 10 --spl 1 32 2>&1 | grep W | sed "s/^/main /"
  (cd .ab/r02 && python tools/kbench.py --worlds 8192 --agents 10 --spl 1 32) 2>&1 | grep W | sed 's/^/r02 /'
done | tee gpurun_out/r03g/variants.log
timeout 2000 python -m pytest tests -x -q -m gpu > gpurun_out/r03g/pytest.log 2>&1; tail -5 gpurun_out/r03g/pytest.log
