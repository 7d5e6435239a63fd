# what profile_round.sh r06_a did not deliver on its first run: the PMC traffic files (script fixed since) and the phase traces (the trace build was missing)
tag=r06_a; out=$PWD/gpurun_out/prof_$tag; mkdir -p $out; repo=$PWD; export TMPDIR=/tmp
for spec in "4 8192 64" "4 8192 1" "10 8192 64" "10 8192 1"; do set -- $spec
  python - <<PY > $out/${tag}_pmc_traffic_n$1_w$2_k$3.json 2>> $out/errors2.txt
import json, sys
sys.path.insert(0, "$repo")
import bench
r = bench.measure_traffic($1, $2, $3, max(4 * $3, 64), timeout_s=400.0, min_agents=2 if $1 == 10 else 0)
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
T=tests/_variants/libcavoid_hip_trace.so
for n in 16384 32768; do for f in quad duo; do echo "== CAVOID_POLICY_FORM=$f rows $n"; CAVOID_POLICY_FORM=$f CAVOID_LIB=$T python tools/trace_policy.py $n 2>&1 | grep -v "amdgpu.ids\|pair "; done; done > $out/${tag}_policy_phase_trace.txt
for q in 0 1; do echo "== CAVOID_ACTOR_QUAD=$q"; CAVOID_ACTOR_QUAD=$q CAVOID_LIB=$T python tools/trace_actor.py 8192 4 16 2>&1 | grep -v amdgpu.ids; done > $out/${tag}_actor_phase_trace.txt
for q in 0 1; do echo "== CAVOID_QUAD=$q"; CAVOID_QUAD=$q CAVOID_LIB=$T python tools/trace_step.py 8192 4 1 2>&1 | grep -v amdgpu.ids; done > $out/${tag}_step_phase_trace.txt
ls -la $out | tail -12; tail -3 $out/errors2.txt
