#!/bin/bash
# The part of tools/profile_round.sh that concerns the K-step relay kernel (4 x 8192): rocprofv3 kernel-trace stats of the default line and of the
# driver's K = 20 form, PMC traffic of the 64-step and one-step launches, the driver's command line itself, and the role timeline.
# usage: bash tools/profile_relay.sh <tag>   (on the GPU box)
tag=${1:-r06_e}; out=$PWD/gpurun_out/prof_$tag; mkdir -p $out; repo=$PWD
export TMPDIR=/tmp
prof() {   # prof <name> <bench args...>
  name=$1; shift
  rm -rf /tmp/rp_$name; mkdir -p /tmp/rp_$name
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/rp_$name -o $name -- python $repo/bench.py "$@" > $out/$name.bench.json 2> $out/$name.err)
  db=$(find /tmp/rp_$name -name "*.db" | head -1)
  if [ -n "$db" ]; then python $repo/tools/rocprof_summary.py $db $out/${tag}_${name}.csv "rocprofv3 --kernel-trace --stats -- python bench.py $*" > /dev/null; else echo "no db for $name" >> $out/errors.txt; find /tmp/rp_$name | head >> $out/errors.txt; fi
}
prof kernel_trace_stats --no-cpu-baseline --no-full-loop --no-configs3 --no-pmc
prof kernel_trace_stats_k20 --steps 20 --warmup 5 --no-cpu-baseline --no-full-loop --no-configs3 --no-pmc
for spec in "4 8192 64" "4 8192 1"; do set -- $spec
  python - <<PY > $out/${tag}_pmc_traffic_n$1_w$2_k$3.json 2>> $out/errors.txt
import json, sys
sys.path.insert(0, "$repo")
import bench
r = bench.measure_traffic($1, $2, $3, max(4 * $3, 64), timeout_s=400.0, min_agents=0)
r = (r or {}).get("one_step" if $3 == 1 else "k_step")
M = $1 - 1
if r is not None:
    per_step = r["traffic"] / r["steps_per_launch"]
    moved = bench.moved_bytes_per_agent_step(M, $1, $3 == 1) * $1 * $2
    r.update({"round": 6, "agents": $1, "worlds": $2, "steps_per_launch": $3, "outputs": "per-step slots [K,W,N,.]" if $3 > 1 else "one step per launch",
              "traffic_bytes_per_step": per_step, "moved_bytes_per_step_expected": moved, "traffic_over_moved": per_step / moved,
              "contract_bytes_per_step": bench.algorithmic_bytes_per_agent_step(M) * $1 * $2})
print(json.dumps(r, indent=1))
PY
done
timeout 300 python bench.py --steps 20 --warmup 5 > $out/${tag}_bench_k20.json 2>> $out/errors.txt   # the driver's command line
timeout 300 python tools/trace_relay.py 8192 4 20 --launch 2>&1 | grep -v amdgpu.ids > $out/${tag}_relay_timeline.txt
ls -la $out; cat $out/errors.txt 2>/dev/null | tail -5
head -c 1200 $out/${tag}_kernel_trace_stats.csv; head -c 1200 $out/${tag}_kernel_trace_stats_k20.csv; cat $out/${tag}_relay_timeline.txt
