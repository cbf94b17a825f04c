"""GPU: a few seconds of the randomised differential runs (tests/fuzz_stats.py: statistics kernels against the CPU
oracle; tests/fuzz_select.py: percentile selection against numpy) with fixed seeds.  Longer runs:
`python tests/fuzz_stats.py 90 1`, `python tests/fuzz_select.py 120 1`."""
import pytest

pytestmark = pytest.mark.gpu


def test_statistics_kernels_random_cases(oracle):
    import fuzz_stats
    assert fuzz_stats.run(6.0, 12345) > 50


def test_percentile_selection_random_cases():
    import fuzz_select
    assert fuzz_select.run(8.0, 54321) > 20
