"""CPU: host-side logic that needs no GPU -- sharding, report naming / CSV schema, write_results, and the
world_size-2 gather of per-layer records over gloo."""
import csv
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lpt_partition_balanced_and_complete():
    from quantization_amd import shapes
    from quantization_amd.distributed import lpt_partition
    for model in shapes.MODELS:
        sizes = [a * b for _, (a, b) in shapes.linear_shapes(model)]
        for world in (1, 2, 4, 8):
            parts = lpt_partition(sizes, world)
            assert sorted(i for p in parts for i in p) == list(range(len(sizes)))
            loads = [sum(sizes[i] for i in p) for p in parts]
            assert max(loads) <= 1.01 * sum(sizes) / world + max(sizes) * (world > 1) * 0.0 + 1 or world == 1
            assert max(loads) / (sum(sizes) / world) < 1.02
    assert lpt_partition([5, 5, 5], 2) == [[0, 2], [1]]
    assert lpt_partition([], 3) == [[], [], []]


def test_model_shape_inventory():
    from quantization_amd import shapes
    counts = {m: len(shapes.linear_shapes(m)) for m in shapes.MODELS}
    assert counts == {"tinyllama-1.1b": 155, "llama-3-8b": 225, "llama-3-70b": 561}
    n8 = sum(a * b for _, (a, b) in shapes.linear_shapes("llama-3-8b"))
    assert n8 == 7_504_658_432


def test_report_naming_and_csv_schema(tmp_path):
    from quantization_amd import report
    assert report.parse_quant_config(os.path.join("onnx_graph", "m", "per-group", "int8_asym", "w_8_gs_128", "model.onnx")) \
        == ("per-group", "int8_asym", 8, 128)
    assert report.parse_quant_config(os.path.join("onnx_graph", "m", "per-channel", "fp8_e4m3", "w_8_gs_none", "model.onnx")) \
        == ("per-channel", "fp8_e4m3", 8, None)
    assert report.parse_quant_config(os.path.join("onnx_graph", "m", "fp16_baseline", "model.onnx")) \
        == ("baseline", "fp16_baseline", None, None)
    assert report.quant_dir_name(8, 128) == "w_8_gs_128" and report.quant_dir_name(8, -1) == "w_8_gs_none"
    assert report.format_code("int8_asym")[0] == 1 and report.format_code("int4")[0] == 0
    assert report.format_code("fp8_e3m4") == (2, "fp8_e3m4")
    with pytest.raises(ValueError):
        report.format_code("bogus")
    recs = torch.tensor([[100.0, 1e-6, 2e-3, 1.1e-8, 3e-8, 4e-4]], dtype=torch.float64)
    rows = report.rows_from_records(["model.layers.0.self_attn.k_proj.weight"], recs, "int8", 8, None)
    path = tmp_path / "r.csv"
    report.write_csv(rows, str(path))
    got = list(csv.reader(open(path)))
    # header of the reference's MSE&P99 multi-weight report (quant_error_multi_weight.py:142-150)
    assert got[0] == ["initializer", "datatype", "granularity", "bits", "mse_tensor", "mse_rep", "mse_p99"]
    assert got[1][:4] == ["model.layers.0.self_attn.k_proj.weight", "int8", "per-channel", "8"]
    assert float(got[1][4]) == 1e-6 / 100.0 and got[1][5] == repr(1.1e-8)


def test_write_results_layout(tmp_path, monkeypatch, capsys):
    from quantization_amd import write_results
    monkeypatch.chdir(tmp_path)
    write_results(12.5, "facebook/opt-1.3b", "wikitext", 8, "int8_asym", 128)
    p = tmp_path / "results_quant" / "wikitext" / "facebook__opt-1.3b" / "w_8_gs_128" / "int8_asym.txt"
    assert p.read_text() == "wikitext perplexity: 12.5 \n"
    write_results(9.0, "tiny", "c4", 16, "fp16", -1)
    assert (tmp_path / "results_quant" / "c4" / "tiny__" / "fp16.txt").read_text() == "c4 perplexity: 9.0 \n"
    write_results(9.5, "a/b", "c4", 4, "int4", 0)
    assert (tmp_path / "results_quant" / "c4" / "a__b" / "w_4_gs_none" / "int4.txt").exists()
    assert "Successfully written results." in capsys.readouterr().out


def _gloo_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from oracle import oracle as orc
    from quantization_amd.distributed import analyze_sharded
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shapes_ = [(8, 64), (32, 128), (4, 64), (16, 256), (8, 128)]
    sizes = [a * b for a, b in shapes_]

    def analyze(i):  # CPU stand-in for the GPU per-layer pass: C oracle + numpy statistics
        rng = np.random.default_rng(100 + i)
        w = (rng.standard_normal(shapes_[i]) * 0.02).astype(np.float16)
        o = orc.quantize(w, orc.F16, orc.INT_SYM, 8, None)
        d = w.astype(np.float32) - o["deq"].view(np.float16).astype(np.float32)
        return torch.tensor([w.size, float((d * d).astype(np.float64).sum()), float(np.abs(d).sum())], dtype=torch.float64)

    table, mine = analyze_sharded(list(range(len(shapes_))), sizes, analyze, 3, torch.device("cpu"))
    full = torch.stack([analyze(i) for i in range(len(shapes_))])
    q.put((rank, mine, bool(torch.equal(table, full))))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_gather_world2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, _, ok in res)
    assert sorted(res[0][1] + res[1][1]) == [0, 1, 2, 3, 4] and res[0][1] and res[1][1]


def test_search_datatype_rejects_other_dtypes_as_the_reference_does():
    """search_datatype on bfloat16 / float32 / float64 weights: the reference's masked assignments raise
    (quant_weight.py:644-645); exception type and message recorded from the imported reference by
    tests/golden/gen_golden.py (search_errors.json).  The check precedes every device call."""
    import json
    import os
    import pytest
    import torch
    pytest.importorskip("ctypes")
    try:
        from quantization_amd import quant_weight as qw
    except ImportError:
        pytest.skip("libmq.so is not built")
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "golden", "search_errors.json")) as f:
        recorded = json.load(f)
    assert len(recorded) == 6
    for key, (exc, msg) in recorded.items():
        dt = getattr(torch, key.split("/")[0])
        gs = None if key.endswith("none") else int(key.split("/")[1])
        with pytest.raises(RuntimeError) as info:
            qw.search_datatype(torch.zeros(8, 256, dtype=dt), 4, "mixed_bitmod", gs)
        assert exc == "RuntimeError" and str(info.value) == msg
