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


def patch_lines(lines, name):
    out, n = [], 0
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
