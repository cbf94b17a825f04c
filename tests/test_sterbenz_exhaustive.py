"""CPU: exhaustive check that the float16 subtraction v - d of the statistics-only kernels is exact on the whole
domain their fast path admits (tests/tools/sterbenz_exhaustive.c): every (scale, value) pair for the INT codes
the clamp leaves alone, every admitted (min, max) pair of an asymmetric row for the codes it cuts, every
(scale, value) pair of every table with subnormals -- ~1.5e10 cases, under a minute with OpenMP."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_float16_error_subtraction_is_exact(tmp_path):
    flags = open("/proc/cpuinfo").read()
    if " f16c" not in flags or " fma" not in flags:
        pytest.skip("host CPU lacks F16C/FMA")
    exe = tmp_path / "sterbenz_exhaustive"
    subprocess.check_call(["gcc", "-O2", "-mf16c", "-mfma", "-fopenmp", "-ffp-contract=off",
                           os.path.join(HERE, "tools", "sterbenz_exhaustive.c"), "-o", str(exe), "-lm"])
    env = dict(os.environ, OMP_NUM_THREADS=str(min(8, os.cpu_count() or 1)))
    rows = {}
    for line in subprocess.check_output([str(exe)], env=env).decode().splitlines():
        name, total, bad = line.split()
        rows[name] = (int(total), int(bad))
    assert rows["int_unclamped"][0] > 600_000_000 and rows["int_unclamped"][1] == 0
    for bits in range(2, 9):
        total, bad = rows[f"asym_clamped_{bits}"]
        assert total > 300_000_000 and bad == 0, bits
    # tables with subnormals take the float16 subtraction: every case must be exact
    for name in ("fp4", "fp5_e2m2", "fp5_e3m1", "fp6_e2m3", "fp6_e3m2", "int3", "int4", "int5", "int6", "int8"):
        total, bad = rows[f"table_{name}"]
        assert total > 700_000_000 and bad == 0, name
    # tables without subnormals do NOT qualify -- a negative x on the midpoint between 0 and the smallest entry
    # rounds to that entry (ties toward -inf) and d = 2 |v| (1 + 2^-11): 13 patterns for fp8_e4m3 -- which is why
    # quant8_stats takes their error in float32
    assert rows["table_fp8_e4m3"][1] == 13
