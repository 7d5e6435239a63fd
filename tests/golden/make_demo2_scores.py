#!/usr/bin/env python
"""Extract the recorded episode scores of the reference's own terminal recording into a small JSON fixture (data, not source).

    python tests/golden/make_demo2_scores.py      # run in the build container, where /root/reference exists

/root/reference/docs/_static/demo2.yml:161-269 is a terminal cast of a `ga3c` training run ("Loading Regression Model then training
RL"); its `[Episode: k Score: s]` lines (ProcessStats.py:98-109 prints them; the score is ProcessAgent.run's `total_reward`,
ProcessAgent.py:230-243) are the only outputs of the reference ENV that exist in-tree.  Stored: episode number, score (4 decimals as
printed), the rolling score printed beside it, and the NT / NP / NA fields of the line."""
import json
import os
import re

SRC = "/root/reference/docs/_static/demo2.yml"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "demo2_scores.json")
LINE = re.compile(r"\[Time:\s*(\d+)\] \[Episode:\s*(\d+) Score:\s*(-?\d+\.\d+)\] \[RScore:\s*(-?\d+\.\d+) RPPS:\s*(\d+)\] "
                  r"\[PPS:\s*(\d+) TPS:\s*(\d+)\] \[NT:\s*(\d+) NP:\s*(\d+) NA:\s*(\d+)\]")


def main():
    with open(SRC) as f:
        text = f.read()
    rows, seen = [], set()
    for m in LINE.finditer(text):
        t, ep, score, rscore, rpps, pps, tps, nt, np_, na = m.groups()
        if int(ep) in seen:
            continue
        seen.add(int(ep))
        rows.append({"time_s": int(t), "episode": int(ep), "score": float(score), "rolling_score": float(rscore), "pps": int(pps),
                     "trainers": int(nt), "predictors": int(np_), "agents": int(na)})
    rows.sort(key=lambda r: r["episode"])
    with open(OUT, "w") as f:
        json.dump({"source": "docs/_static/demo2.yml (terminal recording of a GA3C training run), lines 161-269",
                   "printed_by": "ga3c/GA3C/ProcessStats.py:98-109; score = ProcessAgent.run total_reward (ProcessAgent.py:230-243)",
                   "episodes": rows}, f, indent=1)
    print(len(rows), "episodes ->", OUT)


if __name__ == "__main__":
    main()
