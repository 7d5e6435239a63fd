# same-box A/B: the bookkeeping's first trip to memory issued in front of the env step (.ab/libbase.so = before, .ab/libpf.so = with it)
repo=$PWD; export TMPDIR=/tmp
if [ -z "$SKIP_ACTOR" ]; then
for rep in 1 2 3; do
  for v in pf2 cu32; do
    CAVOID_LIB=$repo/.ab/lib$v.so python tools/actbench.py 8192 4 16 2>&1 | grep steps_per | sed "s/^/$v N=4  /"
    CAVOID_LIB=$repo/.ab/lib$v.so python tools/actbench.py 8192 10 16 2>&1 | grep steps_per | sed "s/^/$v N=10 /"
  done
done
fi
for rep in 1 2; do
for v in pf2 cu32; do
  for n in 4 10; do
    rm -rf /tmp/rp_$v$n; mkdir -p /tmp/rp_$v$n
    (cd /tmp && CAVOID_LIB=$repo/.ab/lib$v.so timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/rp_$v$n -o x -- python $repo/tools/robench.py 8192 $n 600 > /dev/null 2>&1)
    db=$(find /tmp/rp_$v$n -name "*.db" | head -1)
    python tools/rocprof_summary.py $db /tmp/rp_$v$n/s.csv "robench" > /dev/null
    python - <<PY
import csv
for r in csv.reader(l for l in open("/tmp/rp_$v$n/s.csv") if not l.startswith("#")):
    if r and "step_push" in r[0]: print("$v N=$n step_push_kernel calls", r[1], "avg us", r[3])
PY
  done
done
done
