#!/bin/bash
# rocprofv3 evidence for profiles/: kernel-trace --stats of the bench commands (N = 4 and N = 10, default and the driver's
# K = 20 form, the full loop), PMC traffic, and the sweep.  usage: bash tools/profile_round.sh <tag>   (on the GPU box)
tag=${1:-r06_c}; out=$PWD/gpurun_out/prof_$tag; mkdir -p $out; repo=$PWD
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
prof kernel_trace_stats_n10 --agents 10 --no-cpu-baseline --no-full-loop --no-pmc
prof kernel_trace_stats_w1048576 --worlds 1048576 --slices 16 --steps 128 --warmup 32 --no-cpu-baseline --no-full-loop --no-configs3 --no-pmc
prof kernel_trace_stats_n10_w262144 --agents 10 --worlds 262144 --slices 16 --steps 128 --warmup 32 --no-cpu-baseline --no-full-loop --no-pmc
prof full_loop_kernel_trace_stats --steps 256 --warmup 64 --no-cpu-baseline --no-configs3 --no-pmc
# PMC traffic (separate passes per counter, --kernel-trace only beside --pmc)
for spec in "4 8192 64" "4 8192 1" "10 8192 64" "10 8192 1" "4 1048576 16" "10 262144 16"; do set -- $spec
  python - <<PY > $out/${tag}_pmc_traffic_n$1_w$2_k$3.json 2>> $out/errors.txt
import json, sys
sys.path.insert(0, "$repo")
import bench
r = bench.measure_traffic($1, $2, $3, max(4 * $3, 64), timeout_s=400.0, min_agents=2 if $1 == 10 else 0)
r = (r or {}).get("one_step" if $3 == 1 else "k_step")      # (both launch forms come back from one pass: this file is the named one's)
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
# the policy kernel's issue counters (own passes, --kernel-trace only beside --pmc) and the SIMD-sharing micro-benchmark
timeout 900 python tools/pmc.py policy_forward_split $out/${tag}_policy_pmc.json "SQ_BUSY_CYCLES SQ_WAVE_CYCLES,SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY,SQ_INSTS_MFMA SQ_INSTS_VALU,SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU,GRBM_GUI_ACTIVE SQ_WAIT_ANY,SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT,SQ_INSTS_SALU SQ_INSTS_VMEM_RD" -- python tools/polbench.py 32768 3 > /dev/null 2>> $out/errors.txt
timeout 900 python tools/pmc.py actor_kernel $out/${tag}_actor_pmc.json "SQ_BUSY_CYCLES SQ_WAVE_CYCLES,SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY,SQ_INSTS_MFMA SQ_INSTS_VALU,SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU,GRBM_GUI_ACTIVE SQ_WAIT_ANY,SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT,SQ_INSTS_SALU SQ_INSTS_VMEM_RD" -- python tools/actbench.py 8192 4 16 6 > /dev/null 2>> $out/errors.txt
(hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/mfma_valu_overlap.hip -o /tmp/mvo 2>/dev/null && /tmp/mvo > $out/${tag}_mfma_valu_overlap.txt) 2>> $out/errors.txt
(hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/cu_scaling.hip -o /tmp/cus 2>/dev/null && /tmp/cus > $out/${tag}_cu_scaling.txt) 2>> $out/errors.txt
(hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/wave_placement.hip -o /tmp/wp 2>/dev/null && /tmp/wp > $out/${tag}_wave_placement.txt) 2>> $out/errors.txt
for n in 16384 32768; do CAVOID_LIB=tests/_variants/libcavoid_hip_trace.so python tools/trace_policy.py $n 2>&1 | grep -v "amdgpu.ids\|pair " ; done > $out/${tag}_policy_phase_trace.txt
timeout 1500 python bench.py --sweep --full-loop > $out/${tag}_bench.json 2>> $out/errors.txt
timeout 900 python bench.py --agents 10 --sweep --no-full-loop > $out/${tag}_bench_n10.json 2>> $out/errors.txt
timeout 300 python bench.py --steps 20 --warmup 5 > $out/${tag}_bench_k20.json 2>> $out/errors.txt   # the driver's command line
ls -la $out; cat $out/errors.txt 2>/dev/null | tail -5
head -c 1500 $out/${tag}_kernel_trace_stats.csv; head -c 900 $out/${tag}_pmc_traffic_n4_w8192_k64.json
