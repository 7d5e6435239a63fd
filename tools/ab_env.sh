# same-box A/B of variant libraries on the env kernels.  usage: bash tools/ab_env.sh <agents> v1 v2 ...
n=$1; shift
for rep in 1 2 3; do
  for v in "$@"; do
    CAVOID_LIB=$PWD/.ab/lib$v.so python tools/kbench.py --worlds 8192 --agents $n --spl 1 32 2>&1 | grep '"W"' | sed "s/^/$v /"
  done
done
for v in "$@"; do CAVOID_LIB=$PWD/.ab/lib$v.so python tools/kbench.py --worlds 262144 --agents $n --spl 1 2>&1 | grep '"W"' | sed "s/^/$v /"; done
