"""The activation planes' placement inside an LDS row (cavoid_policy_split.hpp: sp_phys) against gfx950's bank model for ds_read_b128
(/opt/skills/guides/MI355X_MICROARCH.md, LDS: 64 banks of 4 bytes, a wave64 read serviced in four NON-contiguous 16-lane groups, lanes of one
group conflict when two of them address different 16-byte slots on the same banks).  The formula is read out of the header, so the test
follows the code; the model says why k-groups g and g ^ 1 sit 256 bytes apart (PMC: profiles/r05_i_slot_mix_ab.txt)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "rl_collision_avoidance_amd", "csrc", "cavoid_policy_split.hpp")

GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
          list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
GROUPS += [[l + 32 for l in g] for g in GROUPS]
STRIDE = 528


def _sp_phys():
    src = open(HDR).read()
    m = re.search(r"constexpr int sp_phys\(int col\) \{\s*return (.*?);\s*\}", src, re.S)
    assert m, "sp_phys not found"
    expr = m.group(1)
    cond, rest = expr.split("?", 1)
    a, b = rest.split(":", 1)
    return lambda col: int(eval(a, {"col": col})) if eval(cond, {"col": col}) else int(eval(b, {"col": col}))


def _worst_way(addr_of_lane):
    worst = 1
    for grp in GROUPS:
        slots = {}
        for l in grp:
            a = addr_of_lane(l)
            assert a % 16 == 0
            slots.setdefault((a // 16) % 16, set()).add(a)      # 16 slots of 16 bytes = the 64 banks
        worst = max(worst, max(len(v) for v in slots.values()))
    return worst


def test_sp_phys_is_a_bijection_on_the_row():
    phys = _sp_phys()
    offs = sorted(phys(c) for c in range(0, 256))
    assert offs == list(range(0, 512, 2))                                # 256 two-byte elements fill bytes 0 .. 511 exactly
    assert [phys(c) for c in range(256, 264)] == list(range(512, 528, 2))  # the zero column behind them
    for c in range(0, 256, 8):                                           # a k-group's 8 columns are one aligned 16-byte slot
        assert phys(c) % 16 == 0 and [phys(c + e) - phys(c) for e in range(8)] == list(range(0, 16, 2))


def test_fragment_reads_are_conflict_free_under_the_gfx950_lane_groups():
    phys = _sp_phys()
    for chunk in range(8):
        # lane l of an activation fragment read: row l % 16 (+ 16 per row tile: 16 x 528 bytes = a multiple of 256), k-group l // 16
        assert _worst_way(lambda l: (l % 16) * STRIDE + phys(32 * chunk + 8 * (l // 16))) == 1
        # rounds 2-4: the four k-groups of a chunk side by side -> a two-way conflict in every group
        assert _worst_way(lambda l: (l % 16) * STRIDE + 64 * chunk + 16 * (l // 16)) == 2
    # the mixed input-slot fragments: k-groups 0, 2, 3 read the first plane's slot, k-group 1 the second plane's (33 792 bytes further)
    for slot in range(0, 24):
        col = 64 + 8 * slot
        assert _worst_way(lambda l: (l % 16) * STRIDE + phys(col) + (64 * STRIDE if l // 16 == 1 else 0)) == 1
