#!/usr/bin/env python
"""Run bench.py's self-launched N-rank dry run many times in a row and keep EVERYTHING a failing attempt printed.

    python tools/launch_flake_hunt.py [--runs 20] [--gpus 8] [--out gpurun_out/flake_hunt] [--cpu]

Round 4 carried a retry in tests/test_bench_launch.py for a failure seen "once in about ten runs", whose message was never kept.
This tool is the hunt: consecutive launches of exactly the command the test runs (`--cpu`: the GPU-free rendezvous-only form),
no retry, per-attempt wall clock, and for every non-zero exit the full stdout / stderr under <out>/attempt_<i>.{out,err}."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=20)
    ap.add_argument("--gpus", type=int, default=8)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "flake_hunt"))
    ap.add_argument("--cpu", action="store_true")
    ap.add_argument("--timeout", type=float, default=600.0)
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    if a.cpu:
        args = ["--gpus", str(a.gpus), "--backend", "gloo", "--rendezvous-only"]
    else:
        args = ["--gpus", str(a.gpus), "--backend", "gloo", "--share-device", "--steps", "20", "--warmup", "5", "--worlds", "1024", "--reps", "3",
                "--no-cpu-baseline", "--no-full-loop", "--no-configs3", "--no-pmc", "--no-fresh-scenarios"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    rows = []
    for i in range(a.runs):
        t0 = time.time()
        try:
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT, timeout=a.timeout,
                                 stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            rc, so, se = out.returncode, out.stdout, out.stderr
        except subprocess.TimeoutExpired as exc:
            rc, so, se = -999, (exc.stdout or b"").decode(errors="replace") if isinstance(exc.stdout, bytes) else (exc.stdout or ""), \
                (exc.stderr or b"").decode(errors="replace") if isinstance(exc.stderr, bytes) else (exc.stderr or "")
        lines = [ln for ln in so.splitlines() if ln.startswith("{")]
        ok = rc == 0 and len(lines) == 1
        rows.append({"attempt": i + 1, "rc": rc, "json_lines": len(lines), "seconds": round(time.time() - t0, 1), "ok": ok})
        if not ok:
            with open(os.path.join(a.out, "attempt_%d.out" % (i + 1)), "w") as f:
                f.write(so)
            with open(os.path.join(a.out, "attempt_%d.err" % (i + 1)), "w") as f:
                f.write(se)
        print(rows[-1], flush=True)
    summary = {"command": "python bench.py " + " ".join(args), "runs": a.runs, "failures": sum(not r["ok"] for r in rows), "attempts": rows}
    with open(os.path.join(a.out, "summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    print(json.dumps({k: summary[k] for k in ("command", "runs", "failures")}))


if __name__ == "__main__":
    main()
