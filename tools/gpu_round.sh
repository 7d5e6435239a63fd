#!/bin/bash
# One GPU session: the whole -m gpu suite, the default bench line, kernel sweeps.  usage: bash tools/gpu_round.sh <tag>
tag=${1:-r03x}; out=gpurun_out/$tag; mkdir -p $out
timeout 1500 python -m pytest tests -x -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" >> $out/bench.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-full-loop --no-configs3 --no-pmc > $out/bench_k20.json 2>> $out/bench.err
timeout 300 python tools/kbench.py --worlds 8192 --agents 4 --spl 1 20 32 > $out/kbench.log 2>&1
timeout 300 python tools/kbench.py --worlds 8192 --agents 10 --spl 1 20 32 >> $out/kbench.log 2>&1
tail -3 $out/pytest.log; tail -2 $out/bench.err; grep -v amdgpu.ids $out/kbench.log
python - <<PY
import json
for f in ("$out/bench.json", "$out/bench_k20.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "value %.3e ms/step %.5f" % (d["value"], d["ms_per_step"]), {k: d["roofline"].get(k) for k in ("bound", "frac", "frac_contract", "kernel_us_per_step", "steps_per_launch", "traffic", "traffic_over_moved")},
              "one-step:", {k: d["roofline"]["one_step_launch"].get(k) for k in ("bound", "frac", "frac_contract", "kernel_us", "traffic_over_moved")})
        ex = d.get("extra", {})
        if "configs3_n10" in ex: print(" n10:", {k: v for k, v in ex["configs3_n10"].items() if k not in ("roofline", "cpu_baseline")}, {k: ex["configs3_n10"].get("roofline", {}).get(k) for k in ("bound", "frac", "frac_contract", "kernel_us_per_step", "traffic_over_moved")})
        if "full_ga3c_loop" in ex: print(" loop:", json.dumps(ex["full_ga3c_loop"])[:600])
    except Exception as e:
        print(f, "unreadable", e)
PY
