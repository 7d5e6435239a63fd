# same-box A/B of variant libraries (.ab/lib<name>.so, tools/mkvariant.sh): policy kernel and actor loop.  usage: bash tools/ab_policy.sh v1 v2 ...
for rep in 1 2 3; do
  for v in "$@"; do
    CAVOID_LIB=$PWD/.ab/lib$v.so python tools/polbench.py 32768 3 2>&1 | grep rows | cut -c1-80 | sed "s/^/$v /"
    CAVOID_LIB=$PWD/.ab/lib$v.so python tools/actbench.py 8192 4 16 2>&1 | grep steps_per | sed "s/^/$v /"
  done
done
for v in "$@"; do CAVOID_LIB=$PWD/.ab/lib$v.so timeout 600 python -m pytest tests/test_gpu_policy.py tests/test_gpu_actor.py -q -k "not 3-130" 2>&1 | tail -3; done
