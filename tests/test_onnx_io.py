"""The ONNX initializer reader (include/mq_onnx.h) against files written by tests/onnx_writer.py (no `onnx` package
is available, so that writer follows onnx.proto as published) and against files written by PyTorch's own exporter
(tests/golden/onnx/); host-only code, runs without a GPU."""
import os

import numpy as np
import pytest

from onnx_writer import write_model


@pytest.fixture(scope="module")
def io():
    from quantization_amd import onnx_io
    return onnx_io


def _model(tmp, sub, scale=1.0, seed=0):
    rng = np.random.default_rng(seed)
    inits = {
        "fc1.weight": (rng.standard_normal((64, 48)) * 0.02 * scale).astype(np.float16),      # Gemm weight, external
        "attn.q_proj.weight": (rng.standard_normal((32, 32)) * 0.02 * scale).astype(np.float16),  # via Transpose
        "small.weight": (rng.standard_normal((4, 8)) * scale).astype(np.float16),              # MatMul weight, inline
        "fc1.bias": rng.standard_normal(64).astype(np.float16),                                # not a weight
        "embed.weight": rng.standard_normal((100, 16)).astype(np.float32),                     # used by Gather only
        "shape_const": np.array([1, -1, 32], np.int64),
        "fc1.weight_q": rng.integers(0, 255, (64, 48)).astype(np.uint8),
    }
    nodes = [("Gather", ["embed.weight", "ids"], ["h0"]),
             ("Gemm", ["h0", "fc1.weight", "fc1.bias"], ["h1"]),
             ("Transpose", ["attn.q_proj.weight"], ["q_t"]),
             ("MatMul", ["h1", "q_t"], ["h2"]),
             ("Reshape", ["h2", "shape_const"], ["h3"]),
             ("MatMul", ["h3", "small.weight"], ["h4"]),
             ("MatMul", ["h4", "h0"], ["h5"])]
    path = os.path.join(tmp, sub, "model.onnx")
    write_model(path, inits, nodes)
    return path, inits


def test_reader_finds_weights_and_bytes(io, tmp_path):
    path, inits = _model(str(tmp_path), "fp16")
    assert os.path.exists(path + ".data")
    with io.OnnxWeights(path) as m:
        assert m.names() == list(inits)
        assert m.compute_weights() == ["fc1.weight", "attn.q_proj.weight", "small.weight"]
        for name, arr in inits.items():
            got = m.array(name)
            assert got.dtype == arr.dtype and got.shape == arr.shape
            assert got.tobytes() == arr.tobytes(), name


def test_reader_on_files_written_by_pytorch(io):
    """Files from PyTorch's own ONNX serializer (tests/golden/onnx/, made by gen_golden.py onnx: the TorchScript
    exporter behind the reference's export script), not from this repository's test writer: every parameter comes
    back by name with its bits and shape, and the MatMul / Gemm weight selection -- directly or through one
    Transpose, pre-transposed constants after folding -- matches what the exporter's graph says."""
    import json
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "onnx")
    with open(os.path.join(gold, "meta.json")) as f:
        meta = json.load(f)
    params = np.load(os.path.join(gold, "params.npz"))
    assert len(meta) == 4
    for name, m in meta.items():
        path = os.path.join(gold, name + ".onnx")
        assert os.path.getsize(path) == m["bytes"]
        with io.OnnxWeights(path) as ow:
            names = ow.names()
            assert set(m["params"]) <= set(names)
            assert sorted(ow.compute_weights()) == m["matmul_gemm_weights"]
            for k in m["params"]:
                want = np.ascontiguousarray(params[f"{name}/{k}"])   # (folded constants are stored transposed)
                got = np.ascontiguousarray(ow.array(k))
                assert tuple(ow.shape(k)) == want.shape, (name, k)
                if got.dtype == np.float16:
                    assert want.dtype == np.int16 and np.array_equal(got.view(np.int16), want), (name, k)
                else:
                    assert got.dtype == want.dtype and np.array_equal(got.view(np.uint8), want.view(np.uint8)), (name, k)


def test_reader_errors(io, tmp_path):
    with pytest.raises(OSError):
        io.OnnxWeights(str(tmp_path / "missing.onnx"))
    bad = tmp_path / "bad.onnx"
    bad.write_bytes(b"\x3a\xff\xff\xff\xff\x0f")  # graph field whose length runs past the end of the file
    with pytest.raises(OSError):
        io.OnnxWeights(str(bad))
    path, _ = _model(str(tmp_path), "trunc")
    with open(path + ".data", "r+b") as f:
        f.truncate(100)
    with pytest.raises(OSError):
        io.OnnxWeights(path)


@pytest.mark.gpu
def test_compare_onnx_report(io, tmp_path, oracle):
    """Two exported models -> the rows of quant_error_multi_weight.py, against the numpy restatement."""
    import torch
    import quantization_amd as qa
    from quantization_amd import report
    base_path, inits = _model(str(tmp_path), os.path.join("onnx_graph", "m", "baseline", "fp16"))
    qinits = dict(inits)
    for name in ("fc1.weight", "attn.q_proj.weight"):
        w = torch.from_numpy(inits[name]).cuda()
        qinits[name] = qa.quant_weight.quant_int_asym(w, 8, 16).cpu().numpy()
    del qinits["small.weight"]          # a weight the quantized model does not hold is skipped
    qpath = os.path.join(str(tmp_path), "onnx_graph", "m", "per-group", "int8_asym", "w_8_gs_16", "model.onnx")
    write_model(qpath, qinits, [])
    rows = report.compare_onnx(base_path, qpath)
    assert [r["initializer"] for r in rows] == ["fc1.weight", "attn.q_proj.weight"]
    for r in rows:
        a = inits[r["initializer"]].astype(np.float32)
        b = qinits[r["initializer"]].astype(np.float32)
        want = oracle.compute_errors_multi(a, b, "per-group", 16)
        assert r["datatype"] == "int8_asym" and r["granularity"] == "per-group" and r["bits"] == 8
        for k in ("mse_tensor", "mse_rep", "mse_p99"):
            assert abs(r[k] - want[k]) <= 1e-6 * abs(want[k]), (r["initializer"], k)
    out = tmp_path / "report.csv"
    report.write_csv(rows, str(out))
    assert out.read_text().splitlines()[0] == "initializer,datatype,granularity,bits,mse_tensor,mse_rep,mse_p99"


@pytest.mark.gpu
def test_compare_onnx_on_the_reference_quantized_exports(io, oracle):
    """quant_error_multi_weight.py's flow on PyTorch-written files: the float16 model and the same model after the
    REFERENCE's quant_model (tests/golden/onnx/tree/..., _LinearDump modules exported with their buffers).
    (1) compare_onnx's rows against the scripts' numpy arithmetic on the stored parameters; (2) this package's
    quant_model on the same float16 weights reproduces every weight / weight_q / weight_scale / weight_zp the
    reference exported, bit for bit."""
    import json
    import torch
    import torch.nn as nn
    from quantization_amd import quant_weight as qw, report
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "onnx")
    with open(os.path.join(gold, "meta.json")) as f:
        meta = json.load(f)
    params = np.load(os.path.join(gold, "params.npz"))
    base = os.path.join(gold, "block_f16.onnx")
    for name, m in meta.items():
        if not name.startswith("tree/"):
            continue
        qpath = os.path.join(gold, *name.split("/")) + ".onnx"
        grouping, dtype, bits, gs = report.parse_quant_config(qpath)
        rows = report.compare_onnx(base, qpath, form="multi", numpy_order=False)
        assert sorted(r["initializer"] for r in rows) == m["matmul_gemm_weights"]
        for r in rows:
            a = params[f"block_f16/{r['initializer']}"].view(np.float16).astype(np.float32)
            b = params[f"{name}/{r['initializer']}"].view(np.float16).astype(np.float32)
            want = oracle.compute_errors_multi(a, b, grouping, gs)
            assert (r["datatype"], r["granularity"], r["bits"]) == (dtype, grouping, bits)
            for k in ("mse_tensor", "mse_rep", "mse_p99"):
                assert abs(r[k] - want[k]) <= 1e-6 * abs(want[k]), (name, r["initializer"], k)
        # numpy_order (the default): the rows the script itself would write, to the last bit
        for r in report.compare_onnx(base, qpath, form="multi"):
            a = params[f"block_f16/{r['initializer']}"].view(np.float16).astype(np.float32)
            b = params[f"{name}/{r['initializer']}"].view(np.float16).astype(np.float32)
            want = oracle.compute_errors_multi(a, b, grouping, gs)
            for k in ("mse_tensor", "mse_rep", "mse_p99"):
                assert r[k] == want[k], (name, r["initializer"], k, "numpy_order")
        # (2) the same model through this package's quant_model
        model = nn.ModuleDict({k: nn.Linear(*reversed(params[f"block_f16/{k}.weight"].shape), bias=False)
                               for k in ("q_proj", "up", "down")}).half().cuda()
        for k in ("q_proj", "up", "down"):
            w = torch.from_numpy(params[f"block_f16/{k}.weight"].view(np.float16).copy())
            model[k].weight.data = w.cuda()
        import contextlib
        import io as _io
        with contextlib.redirect_stdout(_io.StringIO()):
            qw.quant_model(model, bits, dtype, gs)
        for k in ("q_proj", "up", "down"):
            for buf in ("weight", "weight_q", "weight_scale", "weight_zp"):
                key = f"{name}/{k}.{buf}"
                if key not in params:
                    assert not hasattr(model[k], buf)
                    continue
                got = getattr(model[k], buf).detach().cpu().numpy()
                want = params[key]
                want = want.view(np.float16) if want.dtype == np.int16 else want
                assert got.dtype == want.dtype and got.shape == want.shape, (key, got.dtype, want.dtype, got.shape)
                assert np.array_equal(got.view(np.uint8), np.ascontiguousarray(want).view(np.uint8)), key


@pytest.mark.gpu
def test_compare_onnx_single_weight(io, tmp_path, oracle):
    """One initializer against several quantized models -> the rows of quant_error_single_weight.py and the counts
    its plt.hist(vec, bins=50) calls would draw, against the numpy restatement / np.histogram."""
    import torch
    import quantization_amd as qa
    from quantization_amd import report
    root = str(tmp_path)
    base_path, inits = _model(root, os.path.join("onnx_graph", "m", "baseline", "fp16"))
    w = torch.from_numpy(inits["fc1.weight"]).cuda()
    variants = {("per-channel", "fp4", "w_4_gs_none"): qa.quant_weight.quant_datatype(w, 4, "fp4", None),
                ("per-group", "int8", "w_8_gs_16"): qa.quant_weight.quant_int(w, 8, 16),
                ("per-group", "flint4", "w_4_gs_16"): qa.quant_weight.quant_datatype(w, 4, "flint4", 16)}
    qpaths, deqs = [], []
    for (grouping, dtype, leaf), deq in variants.items():
        qp = os.path.join(root, "onnx_graph", "m", grouping, dtype, leaf, "model.onnx")
        write_model(qp, {"fc1.weight": deq.cpu().numpy()}, [])
        qpaths.append(qp)
        deqs.append(deq.cpu().numpy())
    other = os.path.join(root, "onnx_graph", "m", "per-channel", "int8", "w_8_gs_none", "model.onnx")
    write_model(other, {"something.else": inits["small.weight"]}, [])     # lacks the initializer: skipped
    rows, hists = report.compare_onnx_single(base_path, qpaths + [other], "fc1.weight")
    assert [(r["granularity"], r["datatype"], r["bits"]) for r in rows] == [("per-channel", "fp4", 4),
                                                                           ("per-group", "int8", 8),
                                                                           ("per-group", "flint4", 4)]
    a = inits["fc1.weight"].astype(np.float32)
    for r, deq, gs in zip(rows, deqs, (None, 16, 16)):
        want = oracle.compute_errors_single(a, deq.astype(np.float32), r["granularity"], gs)
        for k in ("mse_tensor", "mse_rep", "mse_p99"):
            assert abs(r[k] - want[k]) <= 1e-6 * abs(want[k]), (r["datatype"], k)
        counts, edges = hists[f"{r['datatype']}-{r['granularity']}"]
        assert counts.sum().item() == want["per_vec"].size and len(edges) == 51
    with pytest.raises(ValueError):
        report.compare_onnx_single(base_path, qpaths, "no.such.weight")


def test_reader_survives_corrupt_files_under_sanitizers(tmp_path):
    """The reader parses files it did not write: build it with AddressSanitizer + UBSan (host code only) and
    feed it truncations and random byte flips of a valid model."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "onnx_fuzz")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-static-libasan", "-static-libubsan",
                           "-I", os.path.join(root, "include"), os.path.join(root, "tests", "tools", "onnx_fuzz.cpp"),
                           os.path.join(root, "quantization_amd", "csrc", "mq_onnx.cpp"), "-o", exe])
    path, _ = _model(str(tmp_path), "seed")
    good = open(path, "rb").read()
    rng = np.random.default_rng(0)
    files = [path]
    for i in range(150):
        d = tmp_path / f"mut{i}"
        d.mkdir()
        shutil.copy(path + ".data", d / "model.onnx.data")
        b = bytearray(good)
        if i % 3 == 0:
            b = b[: int(rng.integers(0, len(b)))]
        else:
            for _ in range(int(rng.integers(1, 6))):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        (d / "model.onnx").write_bytes(bytes(b))
        if i % 10 == 0:
            with open(d / "model.onnx.data", "r+b") as f:
                f.truncate(int(rng.integers(0, 4000)))
        files.append(str(d / "model.onnx"))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1")
    r = subprocess.run([exe] + files, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.startswith("opened ")
