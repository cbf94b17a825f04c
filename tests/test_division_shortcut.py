"""CPU: exhaustive check that the kernel's 3-instruction division equals IEEE float32 division on the
whole (float16 value) x (clamped float16 scale) domain -- 912 M pairs, a few seconds with OpenMP."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_reciprocal_fma_division_is_exact(tmp_path):
    flags = open("/proc/cpuinfo").read()
    if " f16c" not in flags or " fma" not in flags:
        pytest.skip("host CPU lacks F16C/FMA")
    exe = tmp_path / "div_exhaustive"
    subprocess.check_call(["gcc", "-O2", "-mf16c", "-mfma", "-fopenmp", "-ffp-contract=off",
                           os.path.join(HERE, "tools", "div_exhaustive.c"), "-o", str(exe), "-lm"])
    env = dict(os.environ, OMP_NUM_THREADS=str(min(8, os.cpu_count() or 1)))
    total, bad_f32, bad_h, bad_mul_only = map(int, subprocess.check_output([str(exe)], env=env).split())
    assert total == (0x70E2 - 0x00A8 + 1) * 0x7c00
    assert bad_f32 == 0 and bad_h == 0
    assert bad_mul_only > 0  # the 1-instruction form v * (1/s) is NOT exact: the correction step is needed
