#!/usr/bin/env python
"""ISA-level bisect of the packed-float32 failure: take the compiler's assembly of the FAILING translation unit
(cavoid_actor_rvo.hip built with -DCAVOID_DEV_PKFORM=0, i.e. the round-4 source left to the vectoriser), edit it with one of the
named patches below, assemble, link and bundle it back into a variant library .ab/libpk_isa_<patch>.so whose other objects are the
pk_c variant's (tools/experiments/pk_opsel_bisect.sh build).  usage: pk_isa_patch.py <patch> [<patch> ...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
VAR = "/tmp/var_pk_c"
CL = "/opt/rocm/lib/llvm/bin"
F = ("--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I%s/include -I%s -DCAVOID_DEV_ONLY_N -mllvm -disable-machine-licm "
     "-DCAVOID_DEV_PKFORM=0" % (ROOT, VAR)).split()
MUL = re.compile(r"^\s*v_pk_mul_f32 .*neg_lo:\[0,1\]")
FMA = re.compile(r"^\s*v_pk_fma_f32 .*op_sel:\[0,0,1\] op_sel_hi:\[1,0,0\]")


VMLOAD = re.compile(r"^\s*(global_load|buffer_load|flat_load|scratch_load)")


def patch_lines(lines, name):
    out, n = [], 0
    if name.startswith("vmwait_"):                                  # vmwait_<lo>_<hi>: s_waitcnt vmcnt(0) behind the lo-th .. (hi-1)-th VMEM load of the file
        lo, hi = (int(x) for x in name.split("_")[1:3])
        k = 0
        for ln in lines:
            out.append(ln)
            if VMLOAD.match(ln):
                if lo <= k < hi:
                    out.append("\ts_waitcnt vmcnt(0)"); n += 1
                k += 1
        return out, n
    if name in ("remul_before_fma", "remul_hi_before_fma"):        # re-execute each dot product's multiply right in front of its fma
        last_mul = {}
        for ln in lines:
            if MUL.match(ln):
                d = re.findall(r"v\[(\d+):(\d+)\]", ln)[0]
                last_mul[d] = ln
            if FMA.match(ln):
                c = re.findall(r"v\[(\d+):(\d+)\]", ln)[3]
                if c in last_mul and re.findall(r"v\[(\d+):(\d+)\]", last_mul[c])[1] != c:      # (not when the mul ran in place)
                    if name == "remul_before_fma":
                        out.append(last_mul[c])
                    else:
                        r = re.findall(r"v\[(\d+):(\d+)\]", last_mul[c])
                        out.append("\tv_mul_f32_e32 v%s, v%s, v%s" % (r[0][1], r[1][1], r[2][0]))
                    n += 1
            out.append(ln)
        return out, n
    MUL3 = re.compile(r"^\s*v_pk_mul_f32 .*op_sel:\[0,1\] op_sel_hi:\[1,0\]")      # the THIRD neighbour's multiply: lo = A.lo * B.hi, hi = A.hi * B.lo
    if name.startswith("mul3_"):
        for k, ln in enumerate(lines):
            if MUL3.match(ln):
                (d0, d1), (a0, a1), (b0, b1) = [(int(x), int(y)) for x, y in re.findall(r"v\[(\d+):(\d+)\]", ln)]
                n += 1
                if name == "mul3_scalar":
                    assert d0 not in (a1, b0), ln
                    out += ["\tv_mul_f32_e32 v%d, v%d, v%d" % (d0, a0, b1), "\tv_mul_f32_e32 v%d, v%d, v%d" % (d1, a1, b0)]
                elif name == "mul3_nop_before":
                    out += ["\ts_nop 7", "\ts_nop 7", ln]
                elif name == "mul3_nop_after":
                    out += [ln, "\ts_nop 7", "\ts_nop 7"]
                elif name == "mul3_twice":                                  # the same instruction issued twice in a row
                    out += [ln, ln]
                elif name == "mul3_drain_before":
                    out += ["\ts_waitcnt vmcnt(0) lgkmcnt(0)", ln]
                else:
                    raise SystemExit("unknown " + name)
            else:
                out.append(ln)
        return out, n
    for ln in lines:
        if name == "identity":
            out.append(ln)
        elif name == "drain_before_mul" and MUL.match(ln):          # every LDS / scalar read landed before the dot products start
            out += ["\ts_waitcnt lgkmcnt(0)", ln]; n += 1
        elif name == "drain_vm_before_mul" and MUL.match(ln):       # ... and every global load
            out += ["\ts_waitcnt vmcnt(0)", ln]; n += 1
        elif name == "drain_before_fma" and FMA.match(ln):
            out += ["\ts_waitcnt lgkmcnt(0)", ln]; n += 1
        elif name == "nop_before_fma" and FMA.match(ln):
            out += ["\ts_nop 7", ln]; n += 1
        elif name == "nop_after_fma" and FMA.match(ln):
            out += [ln, "\ts_nop 7"]; n += 1
        elif name == "nop_before_mul" and MUL.match(ln):
            out += ["\ts_nop 7", ln]; n += 1
        elif name == "wait_after_ds_write_b128" and re.match(r"^\s*ds_write_b128 ", ln):
            out += [ln, "\ts_waitcnt lgkmcnt(0)"]; n += 1
        elif name == "nop_after_ds_write_b128" and re.match(r"^\s*ds_write_b128 ", ln):
            out += [ln, "\ts_nop 7", "\ts_nop 7"]; n += 1
        elif name == "wait_after_ds_write" and re.match(r"^\s*ds_write", ln):
            out += [ln, "\ts_waitcnt lgkmcnt(0)"]; n += 1
        elif name == "wait_after_ds_read" and re.match(r"^\s*ds_read", ln):
            out += [ln, "\ts_waitcnt lgkmcnt(0)"]; n += 1
        elif name == "wait_after_vmem_load" and re.match(r"^\s*(global_load|buffer_load|flat_load|scratch_load)", ln):
            out += [ln, "\ts_waitcnt vmcnt(0)"]; n += 1
        elif name == "wait_after_vmem_store" and re.match(r"^\s*(global_store|buffer_store|flat_store|scratch_store|global_atomic|buffer_atomic)", ln):
            out += [ln, "\ts_waitcnt vmcnt(0)"]; n += 1
        elif name == "wait_before_barrier" and re.match(r"^\s*s_barrier", ln):
            out += ["\ts_waitcnt vmcnt(0) lgkmcnt(0)", ln]; n += 1
        elif name in ("scalar_mul", "scalar_both") and MUL.match(ln):
            # v_pk_mul_f32 D, A, B op_sel_hi:[1,0] neg_lo:[0,1]  ==  D.lo = A.lo * -B.lo ; D.hi = A.hi * B.lo   (as two scalar multiplies)
            d0, d1, a0, a1, b0, b1 = (int(x) for x in re.findall(r"v\[(\d+):(\d+)\]", ln)[0] + re.findall(r"v\[(\d+):(\d+)\]", ln)[1] + re.findall(r"v\[(\d+):(\d+)\]", ln)[2])
            assert d0 not in (a1, b0), ln
            out += ["\tv_mul_f32_e64 v%d, v%d, -v%d" % (d0, a0, b0), "\tv_mul_f32_e32 v%d, v%d, v%d" % (d1, a1, b0)]; n += 1
        elif name in ("scalar_fma", "scalar_both") and FMA.match(ln):
            # v_pk_fma_f32 D, A, B, C op_sel:[0,0,1] op_sel_hi:[1,0,0]  ==  D.lo = A.lo * B.lo + C.hi ; D.hi = A.hi * B.lo + C.lo
            r = re.findall(r"v\[(\d+):(\d+)\]", ln)
            (d0, d1), (a0, a1), (b0, b1), (c0, c1) = [(int(x), int(y)) for x, y in r]
            if (d0, d1) == (c0, c1):
                out += ["\tv_swap_b32 v%d, v%d" % (c0, c1), "\tv_fma_f32 v%d, v%d, v%d, v%d" % (d0, a0, b0, d0), "\tv_fma_f32 v%d, v%d, v%d, v%d" % (d1, a1, b0, d1)]
            else:
                assert d0 not in (a1, b0, c0), ln
                out += ["\tv_fma_f32 v%d, v%d, v%d, v%d" % (d0, a0, b0, c1), "\tv_fma_f32 v%d, v%d, v%d, v%d" % (d1, a1, b0, c0)]
            n += 1
        else:
            out.append(ln)
    return out, n


def run(cmd, **kw):
    subprocess.run(cmd, check=True, **kw)


def main():
    asm = "/tmp/pk_isa/rvo_c.s"
    os.makedirs("/tmp/pk_isa", exist_ok=True)
    if not os.path.exists(asm):
        run(["hipcc"] + F + ["-S", "--cuda-device-only", "-o", asm, VAR + "/cavoid_actor_rvo.hip"], stderr=subprocess.DEVNULL)
    lines = open(asm).read().splitlines()
    for name in sys.argv[1:]:
        out, n = patch_lines(lines, name)
        base = "/tmp/pk_isa/" + name
        with open(base + ".s", "w") as f:
            f.write("\n".join(out) + "\n")
        run([CL + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", base + ".s", "-o", base + ".dev.o"])
        run([CL + "/lld", "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", base + ".dev.o", "-o", base + ".out"])
        run([CL + "/clang-offload-bundler", "-type=o", "-bundle-align=4096", "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950",
             "-input=/dev/null", "-input=" + base + ".out", "-output=" + base + ".hipfb"])
        run(["hipcc"] + F + ["--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", base + ".hipfb", "-c", VAR + "/cavoid_actor_rvo.hip",
                             "-o", base + ".o"], stderr=subprocess.DEVNULL)
        objs = [os.path.join(VAR, o) for o in sorted(os.listdir(VAR)) if o.endswith(".o") and o != "cavoid_actor_rvo.o"] + [base + ".o"]
        lib = os.path.join(ROOT, ".ab", "libpk_isa_%s.so" % name)
        run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-ldl", "-o", lib])
        print(name, "patched sites:", n, "->", lib, flush=True)


if __name__ == "__main__":
    main()
