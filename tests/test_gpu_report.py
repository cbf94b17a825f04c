"""GPU: per-layer report records, the joint (multi-tensor / multi-rank) percentile, and a 2-rank run that
merges percentile histograms across ranks (gloo carrying device tensors: the box has one GPU)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _np_record(orc, w16, datatype, gs, form):
    fmt = {"int8": orc.INT_SYM, "int8_asym": orc.INT_ASYM}.get(datatype, orc.LUT)
    o = orc.quantize(w16, orc.F16, fmt, 8, gs, datatype=datatype if fmt == orc.LUT else None)
    deq = o["deq"].view(np.float16)
    a32, b32 = w16.astype(np.float32), deq.astype(np.float32)
    grouping = "per-channel" if gs is None else "per-group"
    ce = (orc.compute_errors_single if form == "single" else orc.compute_errors_multi)(a32, b32, grouping, gs)
    el = orc.elementwise_f32(w16, deq)
    return ce, el


@pytest.mark.parametrize("form", ["single", "multi"])
def test_layer_record_vs_oracle(oracle, form):
    from quantization_amd import report
    rng = np.random.default_rng(11)
    w16 = (rng.standard_normal((96, 512)) * 0.02).astype(np.float16)
    w = torch.from_numpy(w16).cuda()
    for datatype in ("int8", "int8_asym", "fp8_e4m3", "fp8_e3m4"):
        for gs in (None, 64):
            rec = report.layer_record(w, datatype, 8, gs, form=form).cpu().numpy()
            ce, el = _np_record(oracle, w16, datatype, gs, form)
            assert rec[0] == w16.size
            assert abs(rec[1] / rec[0] - ce["mse_tensor"]) <= 1e-6 * ce["mse_tensor"]
            assert abs(rec[3] - ce["mse_rep"]) <= 1e-6 * ce["mse_rep"]
            assert abs(rec[4] - ce["mse_p99"]) <= 1e-6 * ce["mse_p99"]
            assert np.float32(rec[5]).tobytes() == np.float32(el["p99"]).tobytes()  # abs-error P99: bit-exact
            assert abs(rec[2] - float(el["sum"])) <= 1e-6 * float(el["sum"])
    rows = report.rows_from_records(["l0"], torch.from_numpy(rec)[None], "fp8_e3m4", 8, 64)
    assert rows[0]["granularity"] == "per-group" and rows[0]["mse_p99"] == rec[4]


def test_joint_percentile_single_process():
    from quantization_amd.distributed import joint_percentile
    rng = np.random.default_rng(2)
    for dtype in (np.float32, np.float64):
        parts = [np.abs(rng.standard_normal(n)).astype(dtype) * s for n, s in ((1000, 1), (77777, 1e-3), (5, 10), (300000, 0.1))]
        want = np.percentile(np.concatenate(parts), 99)
        got = joint_percentile([torch.from_numpy(p).cuda() for p in parts], 99).cpu().numpy()
        assert got.tobytes() == np.asarray(want, dtype).tobytes()


def test_wide_percentiles_match_numpy_on_the_float64_array():
    """wide=True: float32 values standing for the float64 array that holds them -- vec = np.array(per_vec) of
    float(np.float32) means, quant_error_single_weight.py:74-79.  Bit for bit np.percentile of that float64 array, on
    every route: one-launch short arrays, the two-pass route, sample + bracket (2^21 values and more), batches,
    lists of different lengths, the joint selection over several tensors."""
    from quantization_amd import error_metrics as em
    from quantization_amd.distributed import joint_percentile
    rng = np.random.default_rng(8)
    arrays = [np.abs(rng.standard_normal(n)).astype(np.float32) ** 2 * s
              for n, s in ((1, 1.0), (5, 10.0), (1000, 1e-6), (40000, 1e-7), (524288, 3e-8), (3000001, 1e-3))]
    arrays[3][:100] = arrays[3][200]          # ties
    for pct in (99, 50, 0, 100, 99.9):
        want = [np.percentile(a.astype(np.float64), pct) for a in arrays]
        dev = [torch.from_numpy(a).cuda() for a in arrays]
        for a, x in zip(want, dev):
            got = em.percentile(x, pct, wide=True)
            assert got.dtype == torch.float64 and got.cpu().numpy().tobytes() == np.float64(a).tobytes(), (pct, x.numel())
        got = em.percentile_multi(dev, pct, wide=True).cpu().numpy()
        assert got.tobytes() == np.asarray(want, np.float64).tobytes(), pct
        j = joint_percentile(dev, pct, wide=True)
        assert j.dtype == torch.float64
        assert j.cpu().numpy().tobytes() == np.float64(np.percentile(np.concatenate(arrays).astype(np.float64), pct)).tobytes()
    m = np.abs(rng.standard_normal((7, 4096))).astype(np.float32)
    got = em.percentile_batch(torch.from_numpy(m).cuda(), 99, wide=True).cpu().numpy()
    assert got.tobytes() == np.percentile(m.astype(np.float64), 99, axis=1).tobytes()
    with pytest.raises(ValueError):
        em.percentile(torch.zeros(8, dtype=torch.float64).cuda(), 99, wide=True)


def test_model_records_vs_oracle(oracle):
    """The model sweep's records against the scripts' numpy arithmetic on the oracle's dequantized weights: the
    per-vector list holds float32 means as the scripts' does, so mse_p99 / mse_rep agree to the float32 summation
    order (1e-6)."""
    from quantization_amd import report
    rng = np.random.default_rng(21)
    shapes = [(64, 512), (32, 1024), (64, 512)]
    w16 = [(rng.standard_normal(s) * 0.02).astype(np.float16) for s in shapes]
    ws = [torch.from_numpy(w).cuda() for w in w16]
    for datatype in ("int8", "int8_asym", "fp8_e4m3"):
        for gs in (None, 32, 128):
            rec, vecs = report.model_records(ws, datatype, 8, gs, return_vecs=True)
            rec = rec.cpu().numpy()
            for i, w in enumerate(w16):
                ce, el = _np_record(oracle, w, datatype, gs, "single")
                assert abs(rec[i, 1] / rec[i, 0] - ce["mse_tensor"]) <= 1e-6 * ce["mse_tensor"]
                assert abs(rec[i, 3] - ce["mse_rep"]) <= 1e-6 * ce["mse_rep"]
                assert abs(rec[i, 4] - ce["mse_p99"]) <= 1e-6 * ce["mse_p99"]
                assert np.float32(rec[i, 5]).tobytes() == np.float32(el["p99"]).tobytes()
                v = vecs[i].cpu().numpy()
                assert v.dtype == np.float32 and v.size == w.size // (gs or w.shape[1])
                # the selection itself is exact: np.percentile of the list the kernel wrote
                assert rec[i, 4] == np.percentile(v.astype(np.float64), 99)
            # numpy_order: the list entries have numpy's bits, so mse_p99 is the reference's value bit for bit --
            # per-group and per-channel (rows go through the wave-per-segment kernel)
            def by_plan():
                plan = report.SweepPlan(ws, [(datatype, 8, gs)], numpy_order=True)
                return plan.run()[0], plan.mse_tensor[0]
            for make in (lambda: report.model_records(ws, datatype, 8, gs, abs_p99=False, numpy_order=True,
                                                      return_mse_tensor=True), by_plan):
                ex, mt = make()
                ex, mt = ex.cpu().numpy(), mt.cpu().numpy()
                for i, w in enumerate(w16):
                    ce, _ = _np_record(oracle, w, datatype, gs, "single")
                    assert ex[i, 4] == ce["mse_p99"], (datatype, gs, i)
                    assert ex[i, 3] == ce["mse_rep"], (datatype, gs, i)   # float(vec.mean()) in numpy's order
                    assert mt[i] == ce["mse_tensor"], (datatype, gs, i)   # float(np.mean((a32 - b32) ** 2))
                    assert abs(ex[i, 1] / ex[i, 0] - ce["mse_tensor"]) <= 1e-6 * ce["mse_tensor"]
    with pytest.raises(ValueError):   # float32 weights: not offered in numpy's order
        report.model_records([w.float() for w in ws], "int8", 8, None, abs_p99=False, numpy_order=True)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from quantization_amd.distributed import joint_percentile
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(40)
    parts = [np.abs(rng.standard_normal(n)).astype(np.float32) * s for n, s in ((50000, 1e-3), (120001, 2e-3), (333, 1.0))]
    mine = [torch.from_numpy(p).cuda() for i, p in enumerate(parts) if i % world == rank]
    got = joint_percentile(mine, 99).cpu().numpy()
    want = np.float32(np.percentile(np.concatenate(parts), 99))
    q.put((rank, got.tobytes() == want.tobytes()))
    dist.barrier()
    dist.destroy_process_group()


def test_histogram_merge_two_ranks():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


SWEEP_SHAPES = [(256, 1024), (64, 512), (256, 1024), (128, 2048), (64, 512), (256, 1024), (512, 256), (128, 2048)]
SWEEP_PASSES = [("int8", 8, None), ("int8_asym", 8, 32), ("fp8_e4m3", 8, 128)]


def _sweep_weights():
    g = torch.Generator(device="cuda").manual_seed(77)
    return [(torch.randn(*s, generator=g, device="cuda") * 0.02).to(torch.bfloat16) for s in SWEEP_SHAPES]


def _sweep_worker(rank, world, port, backend, q):
    """One rank of the sharded sweep as bench.py runs it: SweepPlan over this rank's layers (two streams), the
    record all-reduce and the joint percentile enqueued straight behind it, several steps in a row."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from quantization_amd.distributed import gather_records, joint_percentile, lpt_partition
    from quantization_amd.report import RECORD_FIELDS, SweepPlan
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    ws = _sweep_weights()
    mine = lpt_partition([w.numel() for w in ws], world)[rank]
    plan = SweepPlan([ws[i] for i in mine], SWEEP_PASSES)
    nl, width = len(ws), len(RECORD_FIELDS)
    index = torch.as_tensor([p * nl + i for p in range(len(SWEEP_PASSES)) for i in mine], device=dev)
    total = sum(w.shape[0] for w in ws)
    seen = []
    for it in range(4):
        if it % 2 == 0:   # the whole sweep, then the model-wide percentile
            local = plan.run().reshape(-1, width)
            p99 = joint_percentile(plan.lists(0), 99.0, dtype=torch.float32, total=total, wide=True)
        else:             # bench.py's order: the percentile (and its collectives) beside the last pass's selections
            plan.launch()
            p99 = joint_percentile(plan.lists(0), 99.0, dtype=torch.float32, total=total, wide=True)
            local = plan.finish().reshape(-1, width)
        if world > 1:
            table = gather_records(index, local, nl * len(SWEEP_PASSES), width, dev)
        else:   # one rank: the collective itself, enqueued behind the two-stream sweep
            table = torch.zeros(nl * len(SWEEP_PASSES), width, dtype=torch.float64, device=dev)
            table[index] = local
            dist.all_reduce(table)
        seen.append((table.clone(), p99.clone()))
    torch.cuda.synchronize()
    for t, p in seen[1:]:
        assert torch.equal(t.view(torch.int64), seen[0][0].view(torch.int64)) and torch.equal(p, seen[0][1])
    q.put((rank, table.cpu().numpy(), float(p99)))
    dist.barrier()
    dist.destroy_process_group()


def _run_sweep_ranks(world, backend):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000 + (7 if backend == "nccl" else 0)
    procs = [ctx.Process(target=_sweep_worker, args=(r, world, port, backend, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=240) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


def _sweep_single_process():
    from quantization_amd.report import SweepPlan
    ws = _sweep_weights()
    plan = SweepPlan(ws, SWEEP_PASSES)
    rec = plan.run().cpu().numpy().reshape(len(SWEEP_PASSES) * len(ws), -1)
    lists = np.concatenate([v.cpu().numpy() for v in plan.lists(0)]).astype(np.float64)
    return rec, float(np.percentile(lists, 99))


def test_sharded_sweep_two_ranks_equals_one():
    """Two ranks (gloo standing in for RCCL, both on this GPU): record table and model-wide P99 of the sharded
    sweep equal the single-process sweep bit for bit, with the collectives issued while the sweep's two streams
    still hold work."""
    want_rec, want_p99 = _sweep_single_process()
    for rank, table, p99 in _run_sweep_ranks(2, "gloo"):
        assert np.array_equal(table, want_rec), rank
        assert p99 == want_p99, rank


def test_rccl_collective_behind_two_stream_sweep():
    """RCCL itself (one rank: every box has one GPU): an all-reduce enqueued right behind the sweep's two streams
    completes and leaves the table unchanged."""
    want_rec, want_p99 = _sweep_single_process()
    (rank, table, p99), = _run_sweep_ranks(1, "nccl")
    assert np.array_equal(table, want_rec)
    assert p99 == want_p99


def test_model_records_equals_per_layer_records():
    """The batched whole-model pass (layers grouped by shape, one fused launch per group, P99 of |err| selected
    inside it) gives the per-layer flow's records: percentiles bit for bit, sums exactly (same kernels, same
    order), the mean of the per-vector list to rounding."""
    import torch
    from quantization_amd import report
    g = torch.Generator(device="cuda").manual_seed(11)
    shapes = [(2048, 1024), (256, 1024), (2048, 1024), (1024, 2752), (256, 1024), (2048, 1024), (96, 40)]
    ws = [(torch.randn(*s, generator=g, device="cuda") * 0.02).half() for s in shapes]
    for datatype, gs in (("int8_asym", 64), ("int8", None), ("fp8_e3m4", None)):
        if gs and any(s[1] % gs for s in shapes):
            sub = [w for w in ws if w.shape[1] % gs == 0]
        else:
            sub = ws
        got = report.model_records(sub, datatype, 8, gs).cpu()
        want = torch.stack([report.layer_record(w, datatype, 8, gs, form="single") for w in sub]).cpu()
        assert torch.equal(got[:, [0, 1, 2, 4, 5]], want[:, [0, 1, 2, 4, 5]]), (datatype, gs)
        assert torch.allclose(got[:, 3], want[:, 3], rtol=1e-12, atol=0), (datatype, gs)


def test_layer_record_explicit_list_table(oracle):
    """Formats that go through mq_quantize_table (flint4 here) in the per-layer record and the model sweep."""
    from quantization_amd import fused, report
    torch.manual_seed(2)
    w = (torch.randn(32, 256) * 0.02).half()
    exp = oracle.quantize_table(w.numpy(), oracle.F16, fused.TABLE_LISTS["flint4"], oracle.TABLE_F16, 64)
    st = oracle.compute_errors_single(w.numpy().astype(np.float32),
                                      exp["deq"].view(np.float16).astype(np.float32), "per-group", 64)
    rec = report.layer_record(w.cuda(), "flint4", 4, 64).cpu().numpy()
    assert rec[0] == w.numel()
    assert abs(rec[1] / rec[0] - st["mse_tensor"]) <= 1e-6 * st["mse_tensor"]
    assert abs(rec[3] - st["mse_rep"]) <= 1e-6 * st["mse_rep"] and abs(rec[4] - st["mse_p99"]) <= 1e-6 * st["mse_p99"]
    both = report.model_records([w.cuda(), w.cuda()], "flint4", 4, 64).cpu().numpy()
    assert np.array_equal(both[0], rec) and np.array_equal(both[1], rec)


def test_orderings_the_reference_reports_show():
    """A sanity pin, NOT a bit-wise one: the only error-analysis numbers the reference holds are its two CSV reports
    on an OPT-1.3B weight (MSE&P99/channel_mse_report.csv:2-5, group_mse_report.csv:2-5: 2048 channels, 16 groups
    of 128 per channel), whose weights are not available.  On a seeded 2048 x 2048 N(0, 0.02^2) float16 tensor the
    same relations must hold: tensor MSE int8_asym < int8 < fp8_e3m4 < fp8_e4m3 at both granularities, per-group
    below per-channel for every format, mse_rep = mse_tensor to float32 rounding (the CSV rows agree to 3e-8
    ... 3e-7), and mse_p99 above mse_rep."""
    from quantization_amd import report
    g = torch.Generator(device="cuda").manual_seed(2048)
    w = (torch.randn(2048, 2048, generator=g, device="cuda") * 0.02).half()
    fmts = ["int8_asym", "int8", "fp8_e3m4", "fp8_e4m3"]
    mse = {}
    for gs in (None, 128):
        recs = report.model_records([w], fmts[0], 8, gs)   # (the sweep path)
        for f in fmts:
            r = report.layer_record(w, f, 8, gs, form="single").cpu().numpy()
            if f == fmts[0]:
                assert np.array_equal(r, recs[0].cpu().numpy())
            mse_tensor, mse_rep, mse_p99 = r[1] / r[0], r[3], r[4]
            mse[(f, gs)] = mse_tensor
            assert abs(mse_rep - mse_tensor) <= 1e-6 * mse_tensor, (f, gs)
            assert mse_p99 > mse_rep, (f, gs)
        vals = [mse[(f, gs)] for f in fmts]
        assert vals == sorted(vals) and len(set(vals)) == 4, (gs, vals)
    for f in fmts:
        assert mse[(f, 128)] < mse[(f, None)], f
    # the magnitudes are those of the reports (same formats, a weight of similar scale): within a factor 10
    assert 1e-9 < mse[("int8_asym", 128)] < 1e-7 and 1e-8 < mse[("fp8_e4m3", None)] < 1e-6


def test_percentile_rows_equals_numpy():
    from quantization_amd import error_metrics as em
    rng = np.random.default_rng(5)
    a = np.abs(rng.standard_normal((5, 4096))).astype(np.float32)
    b = np.abs(rng.standard_normal((3, 50000))).astype(np.float32)
    pad = torch.zeros(4, 1008, dtype=torch.float32, device="cuda")          # padded rows: a [:, :1001] view
    c = np.abs(rng.standard_normal((4, 1001))).astype(np.float32)
    pad[:, :1001] = torch.from_numpy(c).cuda()
    got = em.percentile_rows([torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), pad[:, :1001]], 99, wide=True)
    want = np.concatenate([np.percentile(x.astype(np.float64), 99, axis=1) for x in (a, b, c)])
    assert got.dtype == torch.float64 and got.cpu().numpy().tobytes() == want.tobytes()
    got32 = em.percentile_rows([torch.from_numpy(a).cuda()], 50)
    assert got32.cpu().numpy().tobytes() == np.percentile(a, 50, axis=1).astype(np.float32).tobytes()
    with pytest.raises(ValueError):
        em.percentile_rows([torch.from_numpy(a).cuda()[:, 1:]], 99)        # rows off the 16-byte grid


def test_model_sweep_equals_model_records():
    """model_sweep (selections of pass k on a second stream under the launches of pass k + 1) returns the records
    and lists of model_records pass by pass, bit for bit, call after call."""
    from quantization_amd import report
    g = torch.Generator(device="cuda").manual_seed(3)
    shapes = [(1024, 2048), (256, 2048), (1024, 2048), (2048, 5632), (256, 2048), (4096, 1024)]
    ws = [(torch.randn(*s, generator=g, device="cuda") * 0.02).bfloat16() for s in shapes]
    passes = [("int8", 8, None), ("int8", 8, 32), ("int8_asym", 8, 128), ("fp8_e4m3", 8, 64), ("int8", 8, 128)]
    for _ in range(3):
        got, vecs = report.model_sweep(ws, passes, vecs_of=1, numpy_order=False, fuse_groupings=False)
        for k, (dt, bits, gs) in enumerate(passes):
            want, wv = report.model_records(ws, dt, bits, gs, abs_p99=False, return_vecs=True)
            assert torch.equal(got[k].view(torch.int64), want.view(torch.int64)), (dt, gs)
            if k == 1:
                assert all(torch.equal(a, b) for a, b in zip(vecs, wv))
        # the default reads a layer once for the three int8 passes: the same numbers to the 1e-6 bar, the group lists
        # of the integer formats bit for bit
        fg, fvecs = report.model_sweep(ws, passes, vecs_of=1, numpy_order=False)
        assert all(torch.equal(a, b) for a, b in zip(fvecs, vecs))
        assert torch.allclose(fg, got, rtol=3e-6, atol=0.0)
    # the default is numpy's summation order wherever the kernels offer it: the numbers of the reference's CSV rows
    assert report.numpy_order_available(ws, passes)
    got, mt = report.model_sweep(ws, passes, return_mse_tensor=True)
    plan = report.SweepPlan(ws, passes, numpy_order=True)
    want = plan.run()
    assert torch.equal(got.view(torch.int64), want.view(torch.int64))
    assert torch.equal(mt.view(torch.int64), plan.mse_tensor.view(torch.int64))
    assert not report.numpy_order_available([w.float() for w in ws], passes)
    assert not report.numpy_order_available(ws, [("int8", 8, -1)])
    torch.cuda.synchronize()


def test_single_launch_sweep_equals_per_shape_batches():
    """SweepPlan's one-launch form (mq_sweep_run: every layer of a pass in one grid, per-tensor block ranges) against
    the batches of same-shaped layers (mq_quantize_batch): per-vector lists and P99s bit for bit, sums to float64
    rounding.  Shapes: Llama-like and ragged rows, a tensor smaller than one workgroup's share, NaN / inf / one-sided
    rows; float16 and bfloat16 mixed in one model; a float32 weight and a per-tensor pass (not taken by the sweep
    form) ride along in the fallback."""
    from quantization_amd import report
    g = torch.Generator(device="cuda").manual_seed(11)
    shapes = [(1024, 2048), (96, 2048), (3000, 2048), (512, 5632), (8, 2048), (640, 11264), (256, 1024), (33, 8192)]
    ws = [(torch.randn(*s, generator=g, device="cuda") * 0.02) for s in shapes]
    ws[2][5, 7] = float("nan")
    ws[2][9, :64] = float("inf")
    ws[3][3] = ws[3][3].abs() + 0.01                     # a row on one side of zero
    ws[5][1] *= 1e4                                      # scale cut by the clamp
    mixed = [w.bfloat16() if i % 3 else w.half() for i, w in enumerate(ws)]
    for weights in (mixed, [w.bfloat16() for w in ws], mixed[:3] + [ws[3].float()] + mixed[4:]):
        passes = [("int8", 8, None), ("int8", 8, 32), ("int8_asym", 8, 64), ("fp8_e4m3", 8, 128), ("fp4", 4, 256),
                  ("int8_asym", 8, None), ("fp8_e3m4", 8, None), ("int4", 4, 512), ("int8", 8, -1), ("int8", 8, 16)]
        one = report.SweepPlan(weights, passes, fuse_groupings=False)
        per = report.SweepPlan(weights, passes, single_launch=False)
        kinds = [{la[0] for la in ps["launches"]} for ps in one.plan]
        assert all(k == {"batch"} for k in ([{la[0] for la in ps["launches"]} for ps in per.plan]))
        if all(w.dtype != torch.float32 for w in weights):
            assert all(k == {"sweep"} for k in kinds[:8]) and kinds[8] == {"batch"} and kinds[9] == {"batch"}
            # per-group passes: one launch per dtype; per-channel passes: one per (dtype, column count)
            assert len(one.plan[1]["launches"]) == len({w.dtype for w in weights})
        for _ in range(2):
            a, b = one.run().clone(), per.run().clone()
            torch.cuda.synchronize()
            for k in range(len(passes)):
                for la, lb in zip(one.lists(k), per.lists(k)):
                    assert torch.equal(la.view(torch.int32), lb.view(torch.int32)), passes[k]
                # numel, mse_p99: bit for bit; sums over a different number of workgroups: float64 rounding
                for col in (0, 4, 5):
                    assert torch.equal(a[k][:, col].view(torch.int64), b[k][:, col].view(torch.int64)), (passes[k], col)
                for col in (1, 2, 3):
                    x, y = a[k][:, col], b[k][:, col]
                    same = (x == y) | (x.isnan() & y.isnan()) | ((x - y).abs() <= 1e-13 * y.abs())
                    assert bool(same.all()), (passes[k], col)
    # numpy-order entries through the sweep forms: the numpy-order launch (tensors of whole 8192-element chunks: list
    # entries, mse_rep and mse_tensor from the leaf chains of k_np_*) and, with one tensor that is not, the older
    # form (mq_sweep_run with entries in numpy's order, means layer by layer) -- both against the per-shape batches
    bf = [w.bfloat16() for w in ws]
    odd = (torch.randn(10, 1152, generator=g, device="cuda") * 0.02).bfloat16()
    for gs in (32, 64, 128, None):
        for weights, kind in ((bf, "npsweep"), (bf + [odd], "sweep")):
            if gs is None and kind == "sweep":
                continue
            one = report.SweepPlan(weights, [("int8", 8, gs)], numpy_order=True)
            per = report.SweepPlan(weights, [("int8", 8, gs)], numpy_order=True, single_launch=False)
            assert {la[0] for la in one.plan[0]["launches"]} == {kind}, (gs, kind)
            a, b = one.run().clone(), per.run().clone()
            assert all(torch.equal(x.view(torch.int32), y.view(torch.int32)) for x, y in zip(one.lists(0), per.lists(0)))
            for col in (3, 4):   # mse_rep, mse_p99
                x, y = a[0][:, col], b[0][:, col]
                assert bool(((x.view(torch.int64) == y.view(torch.int64)) | (x.isnan() & y.isnan())).all()), (gs, kind, col)
            x, y = one.mse_tensor[0], per.mse_tensor[0]
            assert bool(((x == y) | (x.isnan() & y.isnan())).all()), (gs, kind, "mse_tensor")
    torch.cuda.synchronize()


def test_numpy_list_means_have_numpys_bits():
    """report.numpy_list_means: np.array(v, dtype=float64).mean() of float32 lists, bit for bit -- lengths below 8,
    around the leaf size (128), around and far beyond the 8192-element chunks of numpy's reduction, odd lengths;
    values spanning many binades; a list holding a NaN."""
    from quantization_amd import report
    rng = np.random.default_rng(17)
    lens = [1, 5, 7, 8, 9, 100, 127, 128, 129, 255, 1000, 4096, 8191, 8192, 8193, 8200, 12345, 16384, 128256, 1048579,
            4096, 4096, 128256]
    lists = [(rng.standard_normal(n) ** 2 * 10.0 ** rng.uniform(-9, -5)).astype(np.float32) for n in lens]
    lists[3][2] = np.nan
    dev = [torch.from_numpy(v).cuda() for v in lists]
    got = report.numpy_list_means(dev).cpu().numpy()
    with np.errstate(invalid="ignore"):
        want = np.array([v.astype(np.float64).mean() for v in lists])
    same = (got.view(np.uint64) == want.view(np.uint64)) | (np.isnan(got) & np.isnan(want))
    assert same.all(), [(lens[i], got[i], want[i]) for i in np.nonzero(~same)[0]]


def test_bench_two_ranks_end_to_end(tmp_path):
    """bench.py --gpus 2 as the driver starts it, with gloo standing in for RCCL (both ranks on this GPU): the run
    ends with one JSON line that says n_gpus = 2, both shards hold layers, and the merged record table and model-wide
    P99 of the two ranks equal the one-rank run's bit for bit."""
    import json
    import subprocess
    env = dict(os.environ, MQ_BENCH_BACKEND="gloo")
    outs = {}
    for n in (1, 2):
        path = str(tmp_path / f"table{n}.npz")
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1",
               "--no-extra", "--no-cpu-baseline", "--watchdog", "200", "--dump-table", path]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=400)
        assert r.returncode == 0, r.stderr[-3000:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
        outs[n] = (json.loads(line), np.load(path), r.stderr)
    j2, t2, err2 = outs[2]
    assert j2["n_gpus"] == 2 and j2["config"]["backend"] == "gloo" and j2["scaling"] == "strong"
    assert len(j2["config"]["layers_per_rank"]) == 2 and min(j2["config"]["layers_per_rank"]) > 0
    assert sum(j2["config"]["layers_per_rank"]) == 225
    assert "rank 0/2" in err2 and "rank 1/2" in err2          # every rank names its shard before the first collective
    assert j2["config"]["weight_bytes_per_step_whole_job"] == outs[1][0]["config"]["weight_bytes_per_step_whole_job"]
    t1 = outs[1][1]
    assert t1["table"].shape == t2["table"].shape == (4 * 225, 6)
    assert np.array_equal(t1["table"].view(np.int64), t2["table"].view(np.int64))
    assert t1["p99"].tobytes() == t2["p99"].tobytes()


def _feed_weights(dist="normal"):
    g = torch.Generator(device="cuda").manual_seed(91)
    ws = []
    for s in [(2048, 4096), (4096, 4096), (1024, 4096), (4096, 14336), (512, 2048)]:
        x = torch.randn(*s, generator=g, device="cuda")
        if dist == "student_t":
            chi = sum(torch.randn(*s, generator=g, device="cuda") ** 2 for _ in range(3))
            x = x / torch.sqrt(chi / 3)
        ws.append((x * 0.02).to(torch.bfloat16))
    return ws


@pytest.mark.parametrize("dist", ["normal", "student_t"])
def test_list_percentile_selected_inside_the_pass(dist):
    """Per-group passes over large tensors: the list is not written, its P99 comes from the entries the pass kept
    inside the sampled bracket -- every record equals the one taken from the written list, bit for bit."""
    from quantization_amd import report
    ws = _feed_weights(dist)
    ws[1][7, :64] = float("nan")                       # a tensor whose list holds a NaN: mse_p99 is NaN either way
    passes = [("int8", 8, 32), ("int8_asym", 8, 64), ("fp8_e4m3", 8, 32), ("int8", 8, 128), ("int8", 8, None), ("fp4", 4, 64)]
    fed = report.SweepPlan(ws, passes, feed_lists=True)
    ref = report.SweepPlan(ws, passes, fuse_groupings=False)
    assert bool(fed.fed(0)[:2].all()) and bool(fed.fed(0)[3]) and not bool(fed.fed(0)[2]) and not bool(fed.fed(0)[4])
    assert not bool(fed.fed(4).any()) and not bool(ref.fed(0).any())       # rows are never fed
    assert bool(fed.fed(1)[3])                                            # g = 64: the large tensors
    for it in range(3):
        a = fed.run(overlap=it != 1).clone()
        b = ref.run(overlap=it != 1).clone()
        torch.cuda.synchronize()
        same = (a.view(torch.int64) == b.view(torch.int64)) | (a.isnan() & b.isnan())
        assert bool(same.all()), (dist, it, same.logical_not().nonzero()[:6].tolist())


def test_list_percentile_survives_a_misleading_sample():
    """The sampling launch evaluates wave step c * unit + ((c * 0x9E3779B1 mod 2^32) >> 8) mod unit of every unit of
    steps (2048 weights each at g = 32).  Zero weights in exactly those steps make every sample entry 0 and the bracket
    [0, 0]: the wanted ranks are not inside, the repair launch writes the list after all and the result is the same."""
    from quantization_amd import report
    g = torch.Generator(device="cuda").manual_seed(92)
    rows, cols, gs = 4096, 4096, 32
    w = (torch.randn(rows, cols, generator=g, device="cuda") * 0.02).to(torch.bfloat16)
    nsteps = rows * cols // 2048
    unit = max(16, nsteps // 512)
    c = np.arange(nsteps // unit, dtype=np.uint64)
    step = c * np.uint64(unit) + (((c * np.uint64(0x9E3779B1)) & np.uint64(0xFFFFFFFF)) >> np.uint64(8)) % np.uint64(unit)
    w.view(-1, 2048)[torch.from_numpy(step.astype(np.int64)).cuda()] = 0
    other = (torch.randn(2048, 4096, generator=g, device="cuda") * 0.02).to(torch.bfloat16)
    fed = report.SweepPlan([w, other], [("int8", 8, gs)], feed_lists=True)
    ref = report.SweepPlan([w, other], [("int8", 8, gs)], fuse_groupings=False)
    assert bool(fed.fed(0).all())
    a, b = fed.run().clone(), ref.run().clone()
    torch.cuda.synchronize()
    assert torch.equal(a.view(torch.int64), b.view(torch.int64))
    assert float(a[0, 0, 4]) > 0
    # the repaired tensor's list was written after all
    assert torch.equal(fed.lists(0)[0].view(torch.int32), ref.lists(0)[0].view(torch.int32))


@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_fused_groupings_equal_separate_passes(dt):
    """SweepPlan reads a layer ONCE for the passes of a format that differ only in the grouping (per row, groups of
    32 / 64 / 128: mq_sweep_multi_run) -- against the same plan with one launch per pass: every list entry, every
    record field to the 1e-6 bar of the float64-sum mode (the sums are taken in another order; NaN / inf patterns
    equal), on whole and ragged rows, short rows, rows a workgroup shares, and rows with inf / NaN / one-sided /
    clamped-scale segments that leave the fast path."""
    import numpy as np
    from quantization_amd import report
    from test_gpu_stats_kernels import _inputs
    shapes = [(64, 1024), (16, 4096), (8, 14336), (16, 5632), (32, 11008), (4, 28672), (64, 256), (8, 8192), (32, 1536),
              (40, 128), (12, 2048), (6, 16384),
              # rows that leave whole wave-runs of the kernel's grid empty (skipped), with segments off the fast path
              (6, 28672), (7, 9216), (5, 20480), (9, 14336), (3, 24576 + 128)]
    ws = []
    g = torch.Generator().manual_seed(5)
    for k, (r, c) in enumerate(shapes):
        w = torch.randn(r, c, generator=g) * 0.02 if (k % 2 and k < 12) else _inputs(r, c, 900 + k)
        ws.append((w.half() if dt == "f16" else w.bfloat16()).cuda())
    ws.append((torch.randn(8, 1024, generator=g) * 0.02).cuda())      # float32: not fused (the batched launches take it)
    fams = [("int8", 8), ("int8_asym", 8), ("int4_asym", 4), ("fp8_e4m3", 8), ("fp8_e3m4", 8), ("fp4", 4), ("int2", 2)]
    passes = [(n, b, gs) for n, b in fams for gs in (None, 32, 64, 128)]
    passes += [("int6", 6, 32), ("int6", 6, 128), ("int3", 3, None), ("int3", 3, 64), ("int5", 5, 32)]   # partial families, a single pass
    fused = report.SweepPlan(ws, passes, fuse_groupings=True)
    plain = report.SweepPlan(ws, passes, fuse_groupings=False)
    kinds = [la[0] for ps in fused.plan for la in ps["launches"]]
    assert kinds.count("multisweep") >= 9 * 7, kinds.count("multisweep")
    assert "multisweep" not in [la[0] for ps in plain.plan for la in ps["launches"]]
    a, b = fused.run().cpu().numpy(), plain.run().cpu().numpy()

    def close(x, y, what, tol=1e-6):
        x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
        assert np.array_equal(np.isnan(x), np.isnan(y)), what + ": NaN pattern"
        assert np.array_equal(np.isinf(x), np.isinf(y)), what + ": inf pattern"
        fin = np.isfinite(y)
        err = np.abs(x[fin] - y[fin])
        bad = err > tol * np.abs(y[fin])
        assert not bad.any(), (what, int(bad.sum()), float(np.max(err / np.maximum(np.abs(y[fin]), 1e-300))))

    for k, ps in enumerate(passes):
        for i in range(len(ws)):
            close(fused.lists(k)[i].cpu().numpy(), plain.lists(k)[i].cpu().numpy(), f"{ps} {tuple(ws[i].shape)} list")
        close(a[k, :, :4], b[k, :, :4], f"{ps} records")
        close(a[k, :, 4], b[k, :, 4], f"{ps} P99", tol=3e-6)
    # twice: the plan's buffers are reused
    a2 = fused.run().cpu().numpy()
    assert np.array_equal(a, a2, equal_nan=True)


@pytest.mark.parametrize("fmt,bits", [("int8", 8), ("int4", 4), ("fp8_e4m3", 8), ("fp4", 4)])
def test_fused_groupings_with_equal_maxima_and_partial_families(fmt, bits):
    """The fused launch quantizes only the half of a group that does NOT hold the group's maximum again (formats scaled
    by the absolute maximum).  Here the maxima tie -- every run of 32, or every other one, or one run in four holds the
    same largest magnitude, with either sign -- and the families leave groupings out (64 + 128 without 32, 32 + 128
    without 64, rows + 64), on rows of one wave and of a workgroup: every list entry against the one-launch-per-pass
    kernels, bit for bit for the integer formats."""
    from quantization_amd import report
    g = torch.Generator().manual_seed(11)
    ws = []
    for k, (r, c) in enumerate([(24, 4096), (9, 14336), (16, 1024), (5, 9216), (12, 2048), (7, 8192)]):
        w = torch.randn(r, c, generator=g) * 0.02
        step = (32, 64, 128)[k % 3]
        sign = torch.where(torch.rand(r, c // step, generator=g) < 0.5, -1.0, 1.0)
        w[:, (k * 5) % 32::step] = 0.25 * sign            # the tied maxima (0.25 is exact in float16 / bfloat16)
        if k % 2:
            w[1, :] = 0.0                                  # a zero row: off the fast path
            w[2, 64:128] = 0.0                             # a zero group of 64 inside a live row
        ws.append(w.to(torch.bfloat16 if k % 2 else torch.float16).cuda())
    for sizes in ([64, 128], [32, 128], [None, 64], [None, 32, 64, 128], [32, 64]):
        passes = [(fmt, bits, gs) for gs in sizes]
        fused = report.SweepPlan(ws, passes, fuse_groupings=True)
        plain = report.SweepPlan(ws, passes, fuse_groupings=False)
        assert "multisweep" in [la[0] for ps in fused.plan for la in ps["launches"]]
        a, b = fused.run().cpu().numpy(), plain.run().cpu().numpy()
        for k, ps in enumerate(passes):
            for i in range(len(ws)):
                x, y = fused.lists(k)[i], plain.lists(k)[i]
                if fmt.startswith("int") and ps[2] is not None:
                    assert torch.equal(x.view(torch.int32), y.view(torch.int32)), (ps, tuple(ws[i].shape))
                else:
                    assert torch.allclose(x.double(), y.double(), rtol=1e-6, atol=0, equal_nan=True), (ps, tuple(ws[i].shape))
            assert np.allclose(a[k, :, :5], b[k, :, :5], rtol=3e-6, atol=0, equal_nan=True), ps


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_percentiles_beside_a_busy_chip_equal_the_plain_call(dtype):
    """mq_percentile_multi_beside (256-thread workgroups, for calls that run next to a chip-filling kernel) against
    mq_percentile_multi and numpy: every route -- one-launch short arrays, two-pass histograms, sample + bracket --
    bit for bit, also while another stream keeps the chip busy."""
    from quantization_amd.error_metrics import percentile_multi
    rng = np.random.default_rng(12)
    lens = [1, 2, 7, 100, 2048, 4096, 32768, 32769, 100000, 1 << 20]
    if dtype == torch.float32:
        lens += [(1 << 21) + 5, 5_000_000]
    arrs = [(rng.standard_normal(n) ** 2 * 10.0 ** rng.uniform(-8, -3)).astype(np.float32 if dtype == torch.float32 else np.float64)
            for n in lens]
    arrs[4][:1000] = 0.0                      # ties
    dev = [torch.from_numpy(a).cuda() for a in arrs]
    for pct in (99.0, 50.0, 0.0, 100.0):
        want = np.array([np.percentile(a, pct) for a in arrs], dtype=arrs[0].dtype)
        plain = percentile_multi(dev, pct, nan_check=False).cpu().numpy()
        beside = percentile_multi(dev, pct, nan_check=False, beside=True).cpu().numpy()
        assert plain.tobytes() == want.tobytes()
        assert beside.tobytes() == want.tobytes(), (pct, beside, want)
    # next to a long-running neighbour on another stream
    big = torch.randn(64, 4096, 4096, device="cuda", dtype=torch.float16)
    other = torch.cuda.Stream()
    with torch.cuda.stream(other):
        for _ in range(4):
            big.mul_(1.0001)
    beside = percentile_multi(dev, 99.0, nan_check=False, beside=True).cpu().numpy()
    torch.cuda.synchronize()
    assert beside.tobytes() == np.array([np.percentile(a, 99.0) for a in arrs], dtype=arrs[0].dtype).tobytes()


def test_records_written_straight_into_a_larger_table():
    """SweepPlan.set_table: the plan's rows land in the rows of a table over more layers, every other row is zero on
    EVERY run (the table is all-reduced in place between runs), and the rows equal the plain plan's."""
    from quantization_amd import report
    ws = _feed_weights()
    passes = [("int8", 8, None), ("int8", 8, 32), ("int8", 8, 128), ("fp8_e4m3", 8, 64)]
    want = report.SweepPlan(ws, passes).run().cpu().numpy()
    plan = report.SweepPlan(ws, passes)
    ids = [7, 0, 3, 11, 4]
    plan.set_table(12, ids)
    for rep in range(2):
        got = plan.run()
        torch.cuda.synchronize()
        g = got.cpu().numpy()
        assert g.shape == (len(passes), 12, len(report.RECORD_FIELDS))
        for i, row in enumerate(ids):
            assert g[:, row].tobytes() == want[:, i].tobytes()
        others = [r for r in range(12) if r not in ids]
        assert not g[:, others].any()
        got.add_(1.0)      # what an in-place all-reduce leaves behind must not survive the next run
    with pytest.raises(ValueError):
        plan.set_table(4, ids)
    with pytest.raises(ValueError):
        plan.set_table(12, [1, 1, 2, 3, 4])
