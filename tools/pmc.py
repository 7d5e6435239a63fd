#!/usr/bin/env python
"""Collect rocprofv3 PMC counters for one kernel, one `--pmc` pass per counter group (never mixed with the
trace domains gpurun refuses), and print / save the per-dispatch means as JSON.
usage: python tools/pmc.py <kernel-substring> <out.json> "<CTR1 CTR2,CTR3 CTR4,...>" -- <command...>"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys


def main():
    sep = sys.argv.index("--")
    kernel, out_path, groups = sys.argv[1], sys.argv[2], [g.split() for g in sys.argv[3].split(",")]
    cmd = sys.argv[sep + 1:]
    res = {"kernel_substring": kernel, "command": " ".join(cmd), "counters": {}}
    env = dict(os.environ, TMPDIR="/tmp")
    for gi, grp in enumerate(groups):
        d = "/tmp/pmc_pass_%d" % gi
        shutil.rmtree(d, ignore_errors=True)
        subprocess.run(["rocprofv3", "--pmc"] + grp + ["--kernel-trace", "--output-format", "csv", "-d", d, "--"] + cmd,
                       check=True, cwd="/tmp" if not os.path.isabs(cmd[-1]) and False else None, env=env,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        sums, n = {}, {}
        for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    if kernel not in row["Kernel_Name"]:
                        continue
                    c = row["Counter_Name"]
                    sums[c] = sums.get(c, 0.0) + float(row["Counter_Value"])
                    n[c] = n.get(c, 0) + 1
        for c in sums:
            res["counters"][c] = {"mean_per_dispatch": sums[c] / n[c], "dispatches": n[c]}
        shutil.rmtree(d, ignore_errors=True)
    with open(out_path, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
