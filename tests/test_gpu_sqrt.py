"""cavoid::sqrt_dist2 (the pair pass's square root: the device library's iteration without its rescaling of tiny arguments)
must return the library's correctly rounded sqrt(double) bit for bit -- collision flags and rewards depend on it.  The check
is a standalone HIP program (tests/hip/sqrt_check.hip) compiled here with hipcc against the kernel header and run on the GPU."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sqrt_dist2_is_the_correctly_rounded_sqrt(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    exe = str(tmp_path / "sqrt_check")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                    "-I" + os.path.join(ROOT, "rl_collision_avoidance_amd", "csrc"), os.path.join(ROOT, "tests", "hip", "sqrt_check.hip"),
                    "-o", exe], check=True, timeout=600)
    out = subprocess.run([exe], check=False, capture_output=True, text=True, timeout=300)
    n, bad = out.stdout.split()
    assert out.returncode == 0 and int(n) > 8000000 and int(bad) == 0, out.stdout + out.stderr
