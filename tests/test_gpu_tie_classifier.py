"""`tests/replay.py::classify_divergence` -- the code that decides whether a world that left the float64 oracle is a TIE of the ORCA
linear programme or a real difference -- tested on both sides: the two known ties of the round-3 parity stress (N = 10, box
scenarios generated inside the step, ORCA agents; seeds 21012 / 41012, `profiles/r03_parity_stress_5pass_analysis.txt`) must come
out as ties and the run must otherwise be clean; an injected REAL fault (the oracle's ORCA reciprocity coefficient changed, or its
close-range penalty) must never be excused as one."""
import os
import sys

import pytest

torch = pytest.importorskip("torch")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [21012, 41012])
def test_known_orca_ties_are_classified_as_ties(seed):
    import parity_stress as ps
    r = ps.run(10, 512, 256, seed, 0.4, 1, 1, 0.3, 8, True, 0)
    assert r["ties"] >= 1 and r["unexplained"] == 0, r
    assert r["flag_mismatch"] == 0 and r["episode_mismatch"] == 0, r
    assert r["obs"] <= 1e-5 and r["rew"] <= 1e-5, r


def test_an_ill_conditioned_orca_programme_is_classified_as_a_tie():
    """parity stress pass 5 (round 4; the round-3 kernels give the same): one world of 2048 in which an ORCA agent's linear
    programme is ill-conditioned at one step -- HIP's speed and heading each land one float32 ulp from the oracle's (6e-8 rad,
    1.6e-8 m).  No perturbed oracle answer IS HIP's, but HIP's lies inside the spread of the oracle's own answers under +-1e-13 m."""
    import parity_stress as ps
    r = ps.run(4, 2048, 300, 706, 0.6, 0, 1, 0.5, 1)
    assert r["ties"] >= 1 and r["unexplained"] == 0, r
    assert r["flag_mismatch"] == 0 and r["episode_mismatch"] == 0 and r["obs"] <= 1e-5, r


@pytest.mark.parametrize("seed", [844, 851])
def test_a_drift_over_several_ill_conditioned_steps_is_classified_as_a_tie(seed):
    """parity stress passes 30 .. 59 (round 4, 5.9 G agent-steps): two worlds in which HIP and the oracle part by 7e-9 / 8e-8 at an
    ORCA agent without any single step being a jump -- at the step that crosses the 1e-9 bar the two pre-step states already differ
    by 1e-10, and that step itself amplifies +-1e-13 only to 2e-12.  The oracle's own trajectories, started 1e-13 apart a few
    launches earlier and rolled through the same actions, spread further than HIP is from the oracle; every flag bit agrees."""
    import parity_stress as ps
    r = ps.run(10, 512, 256, seed, 0.5, 1, 1, 0.5, 8)
    assert r["ties"] >= 1 and r["unexplained"] == 0, r
    assert r["flag_mismatch"] == 0 and r["episode_mismatch"] == 0 and r["obs"] <= 1e-5, r


def test_a_clean_orca_run_has_no_ties_to_excuse():
    import parity_stress as ps
    r = ps.run(4, 1024, 200, 700, 0.6, 0, 1, 0.5, 1)
    assert r["unexplained"] == 0 and r["flag_mismatch"] == 0 and r["episode_mismatch"] == 0, r


@pytest.mark.parametrize("fault,rvo", [(dict(rvo_collab_coeff=0.45), 0.5), (dict(reward_getting_close=-0.11), 0.5), (dict(getting_close_range=0.25), 0.0)])
def test_an_injected_fault_is_never_excused_as_a_tie(fault, rvo):
    import parity_stress as ps
    r = ps.run(4, 512, 120, 7, 0.6 if rvo else 0.3, 0, 1 if rvo else 0, rvo, 8, True, 0 if rvo else 4096, oracle_over=fault)
    assert r["unexplained"] >= 1, r
