"""Randomised differential run of the general quantize kernels (every output) against the CPU oracle: dequantized
weights, codes, scales and zero-points bit for bit on random shapes (aligned or not, any divisor as group size,
per-tensor), float16 / bfloat16 / float32 weights, nine formats, zero / one-sided / inf / NaN / subnormal /
heavy-tailed contents; the float32 |error| dump bit for bit against numpy on the oracle's values.
run(seconds, seed); as a script: argv = seconds, seed."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import golden_util as gu
from oracle import oracle as orc
from quantization_amd import _lib
from quantization_amd.fused import quantize


def run(budget, seed):
    orc.lib()
    rng = np.random.default_rng(seed)
    FORMATS = [("int8", 8), ("int8_asym", 8), ("int4", 4), ("int4_asym", 4), ("int2", 2), ("fp8_e4m3", 8), ("fp8_e3m4", 8),
               ("fp8_e2m5", 8), ("fp6_e3m2", 6), ("fp4", 4)]
    t0, n = time.time(), 0
    while time.time() - t0 < budget:
        name, bits = FORMATS[rng.integers(len(FORMATS))]
        rows = int(rng.integers(1, 50))
        cols = int(rng.choice([int(rng.integers(1, 700)), 8 * int(rng.integers(1, 300)), 1024, 4096, 9000, 20000]))
        divs = [d for d in (1, 2, 3, 4, 5, 8, 16, 24, 32, 40, 64, 96, 128, 256, 512, 1000, 1024, 2048, 4096) if cols % d == 0]
        gs = [None, -1, int(rng.choice(divs))][rng.integers(3)]
        dt = ["f16", "bf16", "f32"][rng.integers(3)]
        scale = float(rng.choice([0.02, 1.0, 300.0, 1e-3, 3e-6]))
        w = torch.from_numpy(rng.standard_normal((rows, cols)).astype(np.float32)) * scale
        kind = rng.integers(8)
        if kind == 0: w[rng.integers(rows)] = 0
        if kind == 1: w[rng.integers(rows)] = w[rng.integers(rows)].abs() + scale * 0.1
        if kind == 2: w[rng.integers(rows), rng.integers(cols)] = float("inf")
        if kind == 3: w[rng.integers(rows), rng.integers(cols)] = float("nan")
        if kind == 4: w[:, rng.integers(cols)] = -scale * 40
        if kind == 5: w = torch.from_numpy(rng.standard_t(3, (rows, cols)).astype(np.float32)) * scale
        if kind == 6:   # non-negative rows with zeros of both signs: the sign of a zero zero-point
            w = w.abs()
            zs = torch.from_numpy(rng.random((rows, cols)) < 0.3)
            w[zs] = 0.0
            w[zs & torch.from_numpy(rng.random((rows, cols)) < 0.5)] = -0.0
        wt = {"f16": w.half(), "bf16": w.bfloat16(), "f32": w}[dt]
        if name.startswith("int"):
            asym = "asym" in name
            code, table, oargs = (_lib.INT_ASYM if asym else _lib.INT_SYM), None, dict(fmt=orc.INT_ASYM if asym else orc.INT_SYM, bits=bits)
        else:
            code, table, oargs = _lib.LUT_FP, name, dict(fmt=orc.LUT, bits=bits, datatype=name)
        what = f"[{n}] {name}/{dt}/{rows}x{cols}/g={gs}/scale={scale}/kind={kind}"
        res = quantize(wt.cuda(), code, bits, gs, table, deq=True, codes=True, scale=True, abs_err=torch.float32, totals=True)
        if dt == "f32":
            raw, odt = wt.numpy(), orc.F32
        else:
            raw, odt = wt.contiguous().view(torch.int16).numpy().view(np.uint16), (orc.F16 if dt == "f16" else orc.BF16)
        exp = orc.quantize(raw, odt, group_size=gs, **oargs)
        bits16 = lambda t: t.contiguous().view(torch.int16).cpu().numpy().view(np.uint16)
        if not gu.same_f16_bits(bits16(res.deq).reshape(-1), exp["deq"].reshape(-1)):
            g_, e_ = bits16(res.deq).reshape(-1), exp["deq"].reshape(-1)
            bad = np.nonzero(g_ != e_)[0]
            for i in bad[:6]:
                print(f"  elem {i}: v16 {exp['v16'].reshape(-1)[i]:#06x} got {g_[i]:#06x} want {e_[i]:#06x}", flush=True)
            print("  scale got", bits16(res.scale).reshape(-1)[:4], "want", exp["scale"].reshape(-1)[:4], "mismatches", bad.size, flush=True)
            raise AssertionError(what + " deq")
        assert gu.same_f16_bits(bits16(res.scale).reshape(-1), exp["scale"].reshape(-1)), what + " scale"
        if "asym" in name:
            # (the sign of a zero zero-point included: the first zero in memory decides it below 64 elements, -0 orders
            #  below +0 from there on -- in the kernels as in the oracle)
            g_, e_ = bits16(res.zero_point).reshape(-1).copy(), exp["zp"].reshape(-1).copy()
            if not gu.same_f16_bits(g_, e_):
                L_ = rows * cols if gs == -1 else (gs or cols)
                for i in np.nonzero(g_ != e_)[0][:6]:
                    seg = exp["v16"].reshape(-1, L_)[i]
                    print(f"  seg {i}: zp got {g_[i]:#06x} want {e_[i]:#06x} scale {exp['scale'].reshape(-1)[i]:#06x} values {[hex(x) for x in seg[:8]]}", flush=True)
                raise AssertionError(what + " zp")
        if code == _lib.LUT_FP:
            assert gu.same_f16_bits(bits16(res.codes).reshape(-1), exp["q"].reshape(-1)), what + " codes"
        else:
            got_q = res.codes.cpu().numpy().reshape(-1).view(np.uint8)
            ok = got_q == exp["q"].reshape(-1)
            nan_seg = np.isnan(exp["deq"].view(np.float16).astype(np.float32)).reshape(-1)   # codes of NaN segments: unspecified
            assert (ok | nan_seg).all(), what + " codes"
        v = exp["v16"].view(np.float16).astype(np.float32)
        with np.errstate(invalid="ignore", over="ignore"):
            e = np.abs(v - exp["deq"].view(np.float16).astype(np.float32))
        got_e = res.abs_err.cpu().numpy().reshape(rows, cols)
        assert np.array_equal(got_e, e, equal_nan=True), what + " abs_err"
        n += 1
    return n


if __name__ == "__main__":
    t0 = time.time()
    n = run(float(sys.argv[1]), int(sys.argv[2]))
    print("ok:", n, "random cases in", round(time.time() - t0, 1), "s, seed", sys.argv[2], flush=True)
