mkdir -p gpurun_out/r03c
(cd .ab/r02 && python tools/kbench.py --worlds 8192 --agents 4 --spl 1 64 && python tools/kbench.py --worlds 8192 --agents 10 --spl 1 32) > gpurun_out/r03c/kbench_r02.log 2>&1
(python tools/kbench.py --worlds 8192 --agents 4 --spl 1 64 && python tools/kbench.py --worlds 8192 --agents 10 --spl 1 32) > gpurun_out/r03c/kbench_new.log 2>&1
(cd .ab/r02 && python tools/kbench.py --worlds 8192 --agents 4 --spl 1 64) >> gpurun_out/r03c/kbench_r02.log 2>&1
python tools/kbench.py --worlds 8192 --agents 4 --spl 1 64 >> gpurun_out/r03c/kbench_new.log 2>&1
grep -h W gpurun_out/r03c/kbench_r02.log; echo ---; grep -h W gpurun_out/r03c/kbench_new.log
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r03c/pytest.log 2>&1; tail -15 gpurun_out/r03c/pytest.log
