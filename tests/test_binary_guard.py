"""Guards on the SHIPPED binary (CPU suite: llvm-objdump needs no GPU).

The packed-float32 operand-swap hazard (profiles/r05_b_pk_bisect.txt): on gfx950 a `v_pk_mul_f32` /
`v_pk_add_f32` whose LOW result takes the HIGH half of its SECOND source (`op_sel:[0,1]`) reads that operand as 0 in lanes 48..63 now
and then while another wavefront's MFMAs issue on the same SIMD -- stand-alone reproduction tools/ubench/pk_mul_src1_swap.hip; inside
the fused actor kernel it gave run-to-run wrong observations in round 4.  The sources keep every such chain scalar behind `asm`
fences; this test is what makes that a property of the binary and not of today's compiler mood: every gfx950 code object of the
in-tree `libcavoid_hip.so` is disassembled and must hold NO packed-float32 arithmetic instruction whose `op_sel` lets a low result
read a high half -- of any source: the first- and third-source swaps measured exact, but nothing here needs them either, and a
guard that has to know which operand positions a part gets wrong is a guard that ages (`op_sel_hi` broadcasts are fine and
common; `v_pk_mov_b32 op_sel:[1,0]`, measured exact, is what the compiler uses to move halves about and is not arithmetic).  A
compiler bump or an innocent edit that re-introduces the pattern fails here, in seconds, instead of in a soak."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "rl_collision_avoidance_amd", "libcavoid_hip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

PK = re.compile(r"\bv_pk_(fma|mul|add)_f32\b")
OPSEL = re.compile(r"\bop_sel:\[([01,]+)\]")


def disassemble_code_objects(lib):
    """-> list of (bundle name, disassembly text) of every gfx950 code object of a HIP fat binary"""
    work = tempfile.mkdtemp(prefix="cavoid_objdump_")
    try:
        local = os.path.join(work, os.path.basename(lib))
        shutil.copy(lib, local)                                   # (--offloading extracts the bundles NEXT TO its input)
        subprocess.run([OBJDUMP, "--offloading", local], cwd=work, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        out = []
        for name in sorted(os.listdir(work)):
            if "amdgcn-amd-amdhsa--gfx950" in name:
                text = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", os.path.join(work, name)], check=True, stdout=subprocess.PIPE,
                                      stderr=subprocess.DEVNULL, text=True).stdout
                out.append((name, text))
        return out
    finally:
        shutil.rmtree(work, ignore_errors=True)


def packed_f32_census(text):
    """(packed float32 instructions, those whose op_sel lets a LOW lane read a HIGH half)"""
    total, swapped = 0, []
    for line in text.splitlines():
        if PK.search(line):
            total += 1
            m = OPSEL.search(line)
            if m and "1" in m.group(1):
                swapped.append(line.strip())
    return total, swapped


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump of the ROCm toolchain not present")
def test_no_packed_float32_instruction_reads_a_swapped_low_half():
    assert os.path.exists(LIB), "build the library first: python -m rl_collision_avoidance_amd.build"
    objs = disassemble_code_objects(LIB)
    assert len(objs) >= 8, [n for n, _ in objs]                   # one code object per kernel translation unit
    total, bad = 0, []
    for name, text in objs:
        n, sw = packed_f32_census(text)
        total += n
        bad += [(name, ln) for ln in sw]
    assert total > 10000, total                                    # the census really saw the kernels (18 k packed float32 instructions)
    assert not bad, "packed float32 with a low-half operand swap (DESIGN.md 3.7 (d)):\n" + "\n".join("%s: %s" % b for b in bad[:20])


def test_the_census_recognises_the_pattern():
    """the regexes against the instruction that really misbehaves, round 4's suspect and their harmless relatives"""
    culprit = "v_pk_mul_f32 v[34:35], v[86:87], v[42:43] op_sel:[0,1] op_sel_hi:[1,0]"
    assert packed_f32_census(culprit) == (1, [culprit])
    offender = "v_pk_fma_f32 v[10:11], v[2:3], v[4:5], v[6:7] op_sel:[0,0,1] op_sel_hi:[1,0,0] neg_hi:[0,0,1]"
    fine = ["v_pk_fma_f32 v[10:11], v[2:3], v[4:5], v[6:7] op_sel_hi:[1,0,1]", "v_pk_mul_f32 v[0:1], v[2:3], v[4:5]",
            "v_pk_add_f32 v[0:1], v[2:3], v[4:5] op_sel:[0,0] op_sel_hi:[1,0]", "v_pk_fma_f16 v1, v2, v3, v4 op_sel:[0,0,1]",
            "v_pk_mov_b32 v[0:1], v[2:3], v[4:5] op_sel:[1,0]"]
    assert packed_f32_census(offender) == (1, [offender])
    assert packed_f32_census("\n".join(fine)) == (3, [])
    assert packed_f32_census("v_pk_add_f32 v[0:1], v[2:3], v[4:5] op_sel:[1,0]")[1]
