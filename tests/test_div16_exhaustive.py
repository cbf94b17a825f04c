"""CPU: exhaustive check of the packed float16 division of the statistics kernels (div_pk16 in mq_quant.hip) on
the whole (float16 value) x (float16 scale >= 2^-15) domain the fast path admits -- 664 M pairs, a few seconds with
OpenMP (tests/tools/div16_exhaustive.c states what is checked)."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("args", [(), ("-1",), ("-2",), ("4", "452")],
                         ids=["integer formats", "shared power of two, 2x", "shared power of two, 4x", "scaled variant (tables)"])
def test_packed_float16_division_is_exact(tmp_path, args):
    """(): div_pk16 as the integer formats use it (s' in [1, 2), |v / s| <= 600, bits exact from 2^-4 up).
    ("-1",), ("-2",): a row's power of two shared by a group whose scale is up to 2 x / 4 x smaller (pk_div_params'
    k_base): s' in [1/2, 1) / [1/4, 1/2), bits exact from 2^-3 / 2^-2 up, rint equal everywhere.
    ("4", "452"): s' in [16, 32) -- exact from 2^-8 up, which would cover every table's smallest midpoint; not
    used by the kernels (it saves under 5 issue cycles per pair over the mixed-precision division), kept as the
    record of that result."""
    flags = open("/proc/cpuinfo").read()
    if " f16c" not in flags or " fma" not in flags:
        pytest.skip("host CPU lacks F16C/FMA")
    exe = tmp_path / "div16_exhaustive"
    subprocess.check_call(["gcc", "-O2", "-mf16c", "-mfma", "-fopenmp", "-ffp-contract=off",
                           os.path.join(HERE, "tools", "div16_exhaustive.c"), "-o", str(exe), "-lm"])
    env = dict(os.environ, OMP_NUM_THREADS=str(min(8, os.cpu_count() or 1)))
    out = subprocess.check_output([str(exe), *args], env=env)
    pairs, bad_rint, bad_bits, bad_recip, inexact, bad_small = map(int, out.split())
    assert pairs > 600_000_000
    assert bad_rint == 0 and bad_bits == 0 and bad_recip == 0 and bad_small == 0
    assert inexact == 0   # the emulation of the float16 fma was exact everywhere
