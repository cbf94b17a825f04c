"""GPU: minmax / histogram / weight_histogram against numpy itself -- the reference's plots call np.histogram
(directly or through plt.hist), so numpy on the same values is the reference.  Counts and edges must be equal."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def em():
    from quantization_amd import _lib, error_metrics
    _lib.lib()
    assert torch.cuda.is_available()
    return error_metrics


def _check(em, x_np, bins, range=None):
    want, edges = np.histogram(x_np.astype(np.float32) if x_np.dtype == np.float16 else x_np, bins=bins, range=range)
    got, got_edges = em.histogram(torch.from_numpy(x_np).cuda(), bins=bins, range=range)
    assert got.dtype == torch.int64
    assert np.array_equal(got_edges, edges) and got_edges.dtype == edges.dtype
    assert np.array_equal(got.cpu().numpy(), want)
    return want


def test_abs_error_histogram_200_bins(em):
    rng = np.random.default_rng(0)
    x = np.abs(rng.standard_normal(1 << 20).astype(np.float32) * 3e-4)
    want = _check(em, x, 200)
    assert want.sum() == x.size
    _check(em, x, 200, range=(0.0, 5e-4))       # values above the range are dropped
    _check(em, x[1:70001], 50)                  # a view that is not 16-byte aligned, with a ragged tail


def test_per_vector_mse_histogram_float64(em):
    rng = np.random.default_rng(1)
    x = rng.lognormal(-18.0, 0.4, 262144)
    _check(em, x, 50)
    _check(em, x, 2000)                          # more bins than the per-wave copies hold


def test_values_on_the_edges_and_degenerate_ranges(em):
    rng = np.random.default_rng(2)
    base = rng.standard_normal(50000).astype(np.float32)
    edges = np.histogram_bin_edges(base, bins=64)
    x = np.concatenate([base, edges, np.nextafter(edges, np.float32(-np.inf)), np.nextafter(edges, np.float32(np.inf))])
    x = np.clip(x, base.min(), base.max()).astype(np.float32)
    _check(em, x, 64)
    _check(em, x, edges)                         # explicit edge array
    _check(em, x, np.array([-1.0, -0.25, 0.0, 0.3, 2.5]))   # non-uniform edges
    _check(em, np.full(1000, 0.25, np.float32), 10)         # min == max: numpy widens the range by +-0.5
    _check(em, np.zeros(0, np.float32), 10)                 # empty: range (0, 1), all counts zero
    with pytest.raises(ValueError):
        em.histogram(torch.tensor([1.0, float("nan")]).cuda(), 10)


def test_weight_histogram_two_pass(em):
    torch.manual_seed(0)
    ws = [(torch.randn(256, 512) * 0.02).half(), (torch.randn(100, 333) * 0.05).half(),
          (torch.randn(64, 64) * 0.01).half()]
    arrs = [w.numpy().astype(np.float32).ravel() for w in ws]
    gmin, gmax = np.inf, -np.inf
    for a in arrs:                               # plot_matmul_weight_rev0.py:223-227
        gmin, gmax = min(gmin, a.min()), max(gmax, a.max())
    edges = np.linspace(gmin, gmax, 201)
    want = np.zeros(200, np.int64)
    for a in arrs:
        want += np.histogram(a, bins=edges)[0]
    got, got_edges = em.weight_histogram([w.cuda() for w in ws], bins=200)
    assert np.array_equal(got_edges, edges) and got_edges.dtype == edges.dtype
    assert np.array_equal(got.cpu().numpy(), want) and want.sum() == sum(a.size for a in arrs)
    mm = em.minmax(ws[0].cuda())
    assert mm.tolist() == [float(arrs[0].min()), float(arrs[0].max())]


def test_analyzer_abs_error_histogram(em):
    torch.manual_seed(1)
    w = (torch.randn(300, 257) * 0.02).half()
    q = (w.float() + torch.randn(300, 257) * 2e-4).half()
    an = em.QuantizationErrorAnalyzer(w.cuda(), q.cuda())
    counts, edges = an.abs_error_histogram(200)
    a = np.abs(w.numpy().astype(np.float32) - q.numpy().astype(np.float32)).ravel()
    want, want_edges = np.histogram(a, bins=200)
    assert np.array_equal(edges, want_edges) and np.array_equal(counts.cpu().numpy(), want)
