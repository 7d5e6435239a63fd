export TMPDIR=/tmp; repo=$PWD; mkdir -p gpurun_out/loopprof
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/rp_loop -o loop -- python $repo/bench.py --no-cpu-baseline --no-configs3 --no-pmc > $repo/gpurun_out/loopprof/bench.json 2> $repo/gpurun_out/loopprof/err.txt
db=$(find /tmp/rp_loop -name "*.db" | head -1)
python $repo/tools/rocprof_summary.py $db $repo/gpurun_out/loopprof/stats.csv "loop" > /dev/null
head -14 $repo/gpurun_out/loopprof/stats.csv | cut -c1-150
python - <<PY
import sqlite3
db = sqlite3.connect("$db")
names = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
print([n for n in names if 'kernel' in n.lower()][:20])
PY
