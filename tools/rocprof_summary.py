#!/usr/bin/env python
"""Export the per-kernel statistics of a rocprofv3 run (its rocpd SQLite database, `top_kernels`
view = what `--stats` prints) as a small CSV for profiles/.  Usage:
    python tools/rocprof_summary.py gpurun_out/prof_r1/r1_results.db profiles/r01_xxx.csv [note]"""
import csv
import sqlite3
import sys


def main():
    db_path, out_path = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ""
    cur = sqlite3.connect(db_path).cursor()
    rows = cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    extra = {}
    for name, vg, sg, lds, gx, wx in cur.execute(
            "select name, max(vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) from kernels group by name"):
        extra[name] = (vg, sg, lds, gx, wx)
    with open(out_path, "w", newline="") as f:
        if note:
            f.write("# %s\n" % note)
        f.write("# source: rocprofv3 --kernel-trace --stats (rocpd db view top_kernels); durations in microseconds\n")
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "percent", "vgpr", "sgpr", "lds_bytes", "grid_x", "workgroup_x"])
        for name, calls, total, avg, pct in rows:
            short = name if len(name) < 160 else name[:157] + "..."
            w.writerow([short, calls, "%.3f" % total, "%.3f" % avg, "%.4f" % pct] + list(extra.get(name, ("",) * 5)))
    print(open(out_path).read()[:1500])


if __name__ == "__main__":
    main()
