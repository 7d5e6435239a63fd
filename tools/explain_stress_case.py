#!/usr/bin/env python
"""Re-run single cases of tests/parity_stress.py with the classifier's diagnostics on (CAVOID_STRESS_EXPLAIN=1): for a world the
classifier calls "real", print per agent how far HIP is from the oracle, the oracle's un-cast action and its distance from a float32
rounding boundary, and the programme's conditioning.  usage: python tools/explain_stress_case.py <seed> [<seed> ...]
(the N = 10 box-scenario / ORCA case of the stress: run(10, 512, 256, seed, 0.5, 1, 1, 0.5, 8))"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["CAVOID_STRESS_EXPLAIN"] = "1"
import parity_stress  # noqa: E402

for seed in [int(x) for x in sys.argv[1:]]:
    print("seed", seed, parity_stress.run(10, 512, 256, seed, 0.5, 1, 1, 0.5, 8), flush=True)
