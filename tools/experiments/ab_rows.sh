# same-box A/B of key-parking one-step variants (.ab/lib<name>.so): usage  bash tools/experiments/ab_rows.sh v1 v2 ...
for rep in 1 2 3; do
  for v in "$@"; do
    CAVOID_LIB=$PWD/.ab/lib$v.so python tools/kbench.py --worlds 8192 --agents 10 --spl 1 2>&1 | grep '"W"' | sed "s/^/$v full /"
    CAVOID_LIB=$PWD/.ab/lib$v.so python tools/kbench.py --worlds 8192 --agents 10 --spl 1 --gen-min 2 2>&1 | grep '"W"' | sed "s/^/$v mix  /"
  done
done
for v in "$@"; do
  CAVOID_LIB=$PWD/.ab/lib$v.so python tools/kbench.py --worlds 262144 --agents 10 --spl 1 2>&1 | grep '"W"' | sed "s/^/$v full /"
  CAVOID_LIB=$PWD/.ab/lib$v.so python tools/kbench.py --worlds 262144 --agents 10 --spl 1 --gen-min 2 2>&1 | grep '"W"' | sed "s/^/$v mix  /"
done
