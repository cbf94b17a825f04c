"""Randomised differential run of the model-sweep entry points: report.SweepPlan (prepared launches, two streams)
against report.model_records (pass by pass) against report.layer_record (layer by layer), on random sets of
layer shapes, dtypes and passes: sums and percentiles bit for bit, the list mean to 1e-12 -- with one launch per
pass; and the plan that reads a layer once for the passes that differ only in the grouping (the default) against
that one: to the 1e-6 bar (its sums are taken in another order), the group lists of the integer formats bit for bit.
run(seconds, seed); as a script: argv = seconds, seed."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from quantization_amd import report


def run(budget, seed):
    rng = np.random.default_rng(seed)
    g = torch.Generator(device="cuda").manual_seed(seed)
    FORMATS = ["int8", "int8_asym", "int4", "fp8_e4m3", "fp8_e3m4", "fp4"]
    t0, n = time.time(), 0
    while time.time() - t0 < budget:
        nshapes = int(rng.integers(1, 4))
        # (one shape in four is wide: rows of more than 8192 elements are held by a workgroup, not by a wave)
        shapes = [(int(rng.integers(1, 40)), 128 * int(rng.integers(65, 257))) if rng.integers(4) == 0 else
                  (int(rng.integers(1, 300)), 128 * int(rng.integers(1, 24))) for _ in range(nshapes)]
        layers = [shapes[rng.integers(nshapes)] for _ in range(int(rng.integers(1, 40)))]
        dt = [torch.float16, torch.bfloat16][rng.integers(2)]
        scale = float(rng.choice([0.02, 1.0, 1e-3]))
        ws = [(torch.randn(*s, generator=g, device="cuda") * scale).to(dt) for s in layers]
        if rng.integers(3) == 0:
            ws[rng.integers(len(ws))][0, 0] = float("nan")
        passes = [(FORMATS[rng.integers(len(FORMATS))], 0, [None, 32, 64, 128][rng.integers(4)]) for _ in range(int(rng.integers(1, 5)))]
        passes = [(f, 4 if f in ("int4", "fp4") else 8, gs) for f, _, gs in passes]
        what = f"[{n}] {len(ws)} layers of {shapes} {dt} scale={scale} passes={passes}"
        plan = report.SweepPlan(ws, passes, fuse_groupings=False)
        for overlap in (True, False):
            got = plan.run(overlap=overlap).clone()
            for k, (f, b, gs) in enumerate(passes):
                want, vecs = report.model_records(ws, f, b, gs, abs_p99=False, return_vecs=True)
                same = torch.equal(got[k][:, [0, 1, 2, 4]].nan_to_num(-1.0), want[:, [0, 1, 2, 4]].nan_to_num(-1.0))
                assert same, what + f" pass {k} overlap={overlap}: records differ"
                assert torch.allclose(got[k][:, 3], want[:, 3], rtol=1e-12, atol=0, equal_nan=True), what + " mse_rep"
                for a, b_ in zip(plan.lists(k), vecs):
                    assert torch.equal(a.view(torch.int32), b_.view(torch.int32)), what + f" pass {k}: lists differ"
        # the same passes with the groupings of a format computed from one read
        fused = report.SweepPlan(ws, passes)
        for overlap in (True, False):
            fg = fused.run(overlap=overlap)
            for k, (f, b, gs) in enumerate(passes):
                for a, b_ in zip(fused.lists(k), plan.lists(k)):
                    if gs is not None and f.startswith("int"):   # (a NaN's sign and payload are whatever the last operation left)
                        ok = (a.view(torch.int32) == b_.view(torch.int32)) | (a.isnan() & b_.isnan())
                        assert bool(ok.all()), what + f" pass {k}: fused group lists differ"
                    x, y = a.double(), b_.double()
                    assert torch.equal(x.isnan(), y.isnan()) and torch.equal(x.isinf(), y.isinf()), what + f" pass {k}: fused list NaN / inf"
                    fin = y.isfinite()
                    assert bool(((x[fin] - y[fin]).abs() <= 1e-6 * y[fin].abs()).all()), what + f" pass {k}: fused lists"
                x, y = fg[k][:, :5], got[k][:, :5]
                assert torch.equal(x.isnan(), y.isnan()) and torch.equal(x.isinf(), y.isinf()), what + f" pass {k}: fused records NaN / inf"
                fin = y.isfinite()
                assert bool(((x[fin] - y[fin]).abs() <= 3e-6 * y[fin].abs()).all()), what + f" pass {k}: fused records"
        # numpy order (where the kernels offer it for these weights): the group sizes of a format from one read against
        # one launch per pass, everything bit for bit
        if report.numpy_order_available(ws, passes) and rng.integers(2) == 0:
            a = report.SweepPlan(ws, passes, numpy_order=True, fuse_groupings=False)
            c = report.SweepPlan(ws, passes, numpy_order=True)
            ra, rc = a.run(), c.run()
            torch.cuda.synchronize()
            assert torch.equal(ra.view(torch.int64), rc.view(torch.int64)), what + " numpy order: records differ"
            assert torch.equal(a.mse_tensor.view(torch.int64), c.mse_tensor.view(torch.int64)), what + " numpy order: mse_tensor"
            for k in range(len(passes)):
                for x, y in zip(a.lists(k), c.lists(k)):
                    assert torch.equal(x.view(torch.int32), y.view(torch.int32)), what + f" numpy order pass {k}: lists differ"
        # one layer, one pass against the per-layer flow
        k = int(rng.integers(len(passes))); i = int(rng.integers(len(ws)))
        f, b, gs = passes[k]
        one = report.layer_record(ws[i], f, b, gs, form="single", abs_p99=False)
        assert torch.equal(one[[0, 1, 2, 4]].nan_to_num(-1.0), got[k][i][[0, 1, 2, 4]].nan_to_num(-1.0)), what + f" layer {i} pass {k} vs layer_record: {one.tolist()} {got[k][i].tolist()}"
        n += 1
    return n


if __name__ == "__main__":
    t0 = time.time()
    n = run(float(sys.argv[1]), int(sys.argv[2]))
    print("ok:", n, "random cases in", round(time.time() - t0, 1), "s, seed", sys.argv[2], flush=True)
