"""The one first-party datum of the env half: the 62 episode scores of the reference's terminal recording
(/root/reference/docs/_static/demo2.yml:161-269 -> tests/golden/demo2_scores.json, extracted by tests/golden/make_demo2_scores.py).

It cannot pin env.step (no states, no seeds), but every score must be EXPRESSIBLE under the oracle's reward algebra -- goals x 1.0 +
collisions x (-0.25) + a sum of getting-close terms, divided by the number of learning agents (ProcessAgent.py:168,195) -- and the set
says something about Appendix A's U5 (the sign of the getting-close slope).  What it says is pinned here so that DESIGN.md section 0 quotes
a test, not an impression."""
import json
import os

from oracle.cavoid_oracle import OracleConfig
from oracle.score_algebra import RewardAlgebra, explanations, fewest_close_steps

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "demo2_scores.json")) as f:
    EPISODES = json.load(f)["episodes"]
SCORES = [e["score"] for e in EPISODES]


def _alg(slope):
    c = OracleConfig()
    return RewardAlgebra(c.reward_at_goal, c.reward_collision, c.reward_getting_close, c.getting_close_range, slope)


def test_fixture_is_the_recording():
    assert len(EPISODES) == 62 and EPISODES[0]["episode"] == 1 and EPISODES[-1]["episode"] == 62
    assert SCORES[0] == -0.3135 and EPISODES[-1]["pps"] == 563 and {e["agents"] for e in EPISODES} == {32}
    assert -0.4248 in SCORES and 0.1667 in SCORES and 0.375 in SCORES and 0.6875 in SCORES


def test_rolling_score_is_the_mean_of_the_scores_so_far():
    """ProcessStats keeps a rolling mean over the last STAT_ROLLING_MEAN_WINDOW episodes (ProcessStats.py:76-96); the recording skips a few
    episode lines (two printed at once), so the check is made where the prefix is complete"""
    total = 0.0
    for k, e in enumerate(EPISODES):
        if e["episode"] != k + 1:
            break
        total += e["score"]
        assert abs(total / (k + 1) - e["rolling_score"]) < 2e-4 * (k + 1) / (k + 1) + 1e-4, e


def test_pure_outcome_scores_are_exact_fractions_of_the_reward_constants():
    """scores without any getting-close term identify (learning agents, goals, collisions) and fix the divisor as the CONSTANT number of
    learning agents (U6: an agent stays `learning` after it is done) -- 0.375 = (1 - 0.25) / 2, 0.1667 = (1 - 0.25 - 0.25) / 3,
    0.6875 = (3 - 0.25) / 4, -0.1667 = (-0.25 - 0.25 + 0) / 3"""
    want = {0.375: (2, 1, 1), 0.1667: (3, 1, 2), 0.6875: (4, 3, 1), -0.1667: (3, 0, 2), 0.5: (2, 1, 0), 1.0: (1, 1, 0), -0.25: (1, 0, 1), 0.0: (1, 0, 0)}
    for s, (n, g, c) in want.items():
        assert s in SCORES
        pure = [(e.n, e.goals, e.collisions) for e in explanations(s, _alg(-0.5)) if e.close_steps == 0]
        assert (n, g, c) in pure, (s, pure)
    # a divisor that shrank as agents finished (learning -> False at done) would make 0.375 and 0.6875 unreachable without close terms:
    # 2 agents: -0.25/2 + 1/1 = 0.875 or 1/2 - 0.25/1 = 0.25; neither is in the recording
    assert 0.875 not in SCORES and 0.25 not in SCORES


def test_every_score_is_expressible_under_both_signs_of_u5():
    """neither sign of the getting-close slope is EXCLUDED by the recording: every score has an explanation with n <= 4 under both"""
    for s in SCORES:
        for slope in (-0.5, +0.5):
            assert fewest_close_steps(s, _alg(slope)) is not None, (s, slope)


def test_what_the_recording_says_about_u5():
    """... but the two signs are not equally parsimonious.  A collision is approached through the (0, 0.2] m band in about one step
    (closing speeds of 1..4 m/s x 0.2 s), so a collision episode scores  -0.25 + t  per colliding agent with t ONE getting-close term:
    under slope +0.5 (paper) t is in (-0.1, 0], scores in (-0.35, -0.25]; under -0.5 (upstream code as recalled) t is in [-0.2, -0.1),
    scores in [-0.45, -0.35).  The recording: 18 scores in (-0.35, -0.25), 4 in [-0.45, -0.35), 11 exactly -0.25."""
    band_plus = [s for s in SCORES if -0.35 < s < -0.25]
    band_minus = [s for s in SCORES if -0.45 <= s <= -0.35]
    assert (len(band_plus), len(band_minus), SCORES.count(-0.25)) == (18, 4, 11)
    # one learning agent, one collision, at most one getting-close step: admitted by +0.5 for all 18, by -0.5 for none of them
    def one_step(s, slope):
        return any(e.n == 1 and e.collisions == 1 and e.goals == 0 and e.close_steps <= 1 for e in explanations(s, _alg(slope), max_agents=1))
    assert all(one_step(s, +0.5) for s in band_plus) and not any(one_step(s, -0.5) for s in band_plus)
    # -0.3135 (the first score, SURVEY App. A's example): with ONE learning agent -0.5 has no explanation that contains the collision
    # (only "no collision, two getting-close steps, timed out"); +0.5 has the plain one: collision after one step at a gap of 0.073 m
    n1_minus = [e for e in explanations(-0.3135, _alg(-0.5), max_agents=1)]
    n1_plus = [e for e in explanations(-0.3135, _alg(+0.5), max_agents=1)]
    assert n1_minus and all(e.collisions == 0 for e in n1_minus) and min(e.close_steps for e in n1_minus) == 2
    assert (0, 1, 1) in [(e.goals, e.collisions, e.close_steps) for e in n1_plus]
    # over the whole recording: episodes explained by ONE learning agent with at most one getting-close step, and the learning agents
    # the most parsimonious explanation needs in total
    simple = {slope: sum(any(e.close_steps <= 1 for e in explanations(s, _alg(slope), max_agents=1)) for s in SCORES) for slope in (-0.5, +0.5)}
    agents = {slope: sum(fewest_close_steps(s, _alg(slope)).n for s in SCORES) for slope in (-0.5, +0.5)}
    assert simple == {-0.5: 24, +0.5: 36} and agents == {-0.5: 125, +0.5: 103}, (simple, agents)
