"""Randomised differential run of the explicit-list formats through the reference-shaped API (_fp_quant_meta /
quant_datatype / quant_mx / search_datatype: mq_quantize_table, mq_search_tables, and the closed-form tables'
float32 per-tensor branch) against the CPU oracle, every output bit for bit (a NaN equals a NaN).
run(seconds, seed); as a script: argv = seconds, seed."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import oracle as orc
from quantization_amd import fused, quant_weight as qw


def _bits(t):
    t = t.contiguous().cpu()
    if t.dtype == torch.float32:
        return t.numpy().reshape(-1)
    return t.view(torch.int16).numpy().view(np.uint16).reshape(-1)


def _same(a, b, what):
    a, b = np.asarray(a).reshape(-1), np.asarray(b).reshape(-1)
    if a.dtype == np.uint16:
        an, bn = (a & 0x7fff) > 0x7c00, (b & 0x7fff) > 0x7c00
        ok = np.array_equal(an, bn) and np.array_equal(a[~an], b[~bn])
    else:
        ok = np.array_equal(a, b, equal_nan=True) and np.array_equal(np.signbit(a[a == 0]), np.signbit(b[b == 0]))
    if not ok:
        bad = np.nonzero(a != b)[0]
        raise AssertionError(f"{what}: {bad.size} mismatches, first {[(int(i), a[i], b[i]) for i in bad[:5]]}")


def run(budget, seed):
    orc.lib()
    rng = np.random.default_rng(seed)
    LUTS = [("fp8_e4m3", 8), ("fp8_e3m4", 8), ("fp8_e2m5", 8), ("fp6_e3m2", 6), ("fp6_e2m3", 6), ("fp5_e2m2", 5), ("fp4", 4)]
    LISTS = [("fp3", 3), ("flint4", 4), ("fp4_er_pos", 4), ("fp4_ea_neg", 4), ("fp3_er_neg", 3), ("fp3_ea_pos", 3)]
    MX = [("mx_int3", 3), ("mx_fp3", 3), ("mx_int4", 4), ("mx_fp4", 4)]
    MIXED = ["mixed_bitmod", "mixed_er", "mixed_ea", "mixed_ant"]
    t0, n = time.time(), 0
    while time.time() - t0 < budget:
        fam = ["lutpt", "list", "list", "mx", "search", "e5m2"][rng.integers(6)]
        rows = int(rng.integers(1, 40))
        cols = int(rng.choice([int(rng.integers(1, 700)), 8 * int(rng.integers(1, 300)), 1024, 4096, 9000]))
        divs = [d for d in (1, 2, 3, 4, 5, 8, 16, 24, 32, 40, 64, 96, 128, 256, 512, 1024, 4096) if cols % d == 0]
        gs = [None, -1, int(rng.choice(divs)), 0][rng.integers(4)]
        dtn = ["f16", "bf16", "f32"][rng.integers(3)]
        scale = float(rng.choice([0.02, 1.0, 300.0, 1e-3, 3e-6, 9000.0]))
        w = torch.from_numpy(rng.standard_normal((rows, cols)).astype(np.float32)) * scale
        kind = rng.integers(8)
        if kind == 0: w[rng.integers(rows)] = 0
        if kind == 1: w[rng.integers(rows)] = w[rng.integers(rows)].abs() + scale * 0.1
        if kind == 2: w[rng.integers(rows), rng.integers(cols)] = float("inf")
        if kind == 3: w[rng.integers(rows), rng.integers(cols)] = float("nan")
        if kind == 4: w[:, rng.integers(cols)] = -scale * 40
        if kind == 5: w = torch.from_numpy(rng.standard_t(3, (rows, cols)).astype(np.float32)) * scale
        wt = {"f16": w.half(), "bf16": w.bfloat16(), "f32": w}[dtn]
        raw = wt.numpy() if dtn == "f32" else wt.contiguous().view(torch.int16).numpy().view(np.uint16)
        raw = raw if dtn != "f16" else raw.view(np.float16)
        odt = {"f16": orc.F16, "bf16": orc.BF16, "f32": orc.F32}[dtn]
        what = f"[{n}] {fam}/{dtn}/{rows}x{cols}/g={gs}/scale={scale}/kind={kind}"
        dev = wt.cuda()
        if fam == "lutpt":     # closed-form tables: quant_datatype's float32 per-tensor branch and the float16 one
            name, b = LUTS[rng.integers(len(LUTS))]
            what += f"/{name}"
            d2 = qw.quant_datatype(dev, wq_bits=b, datatype=name, group_size=gs)
            exp = orc.quantize(raw, odt, fmt=orc.LUT, bits=b, datatype=name, group_size=gs, lut_f32_path=(gs == -1))
            _same(_bits(d2), exp["deq"], what + " deq(quant_datatype)")
        elif fam == "list":
            name, b = LISTS[rng.integers(len(LISTS))]
            what += f"/{name}"
            d, q, s = qw._fp_quant_meta(dev, wq_bits=b, datatype=name, group_size=gs)
            exp = orc.quantize_table(raw, odt, fused.TABLE_LISTS[name], orc.TABLE_F16, gs)
            _same(_bits(d), exp["deq"], what + " deq"); _same(_bits(q), exp["q"], what + " q"); _same(_bits(s), exp["scale"], what + " scale")
            d2 = qw.quant_datatype(dev, wq_bits=b, datatype=name, group_size=gs)
            if gs == -1:
                exp = orc.quantize_table(raw, odt, fused.TABLE_LISTS[name], orc.TABLE_F32, gs)
            _same(_bits(d2), exp["deq"], what + " deq(quant_datatype)")
        elif fam == "e5m2":
            d, q, s = qw._fp_quant_meta(dev, wq_bits=8, datatype="fp8_e5m2", group_size=gs)
            exp = orc.quantize_table(raw, odt, fused.TABLE_LISTS["fp8_e5m2"], orc.TABLE_F32, gs, deq_f32=True, neg_inf_zero=True)
            _same(_bits(d), exp["deq"], what + " deq"); _same(_bits(q), exp["q"], what + " q"); _same(_bits(s), exp["scale"], what + " scale")
        elif fam == "mx":
            name, b = MX[rng.integers(len(MX))]
            g = int(rng.choice([d_ for d_ in divs if d_ >= 2] or [cols]))
            what += f"/{name}/mxg={g}"
            d = qw.quant_mx(dev, wq_bits=b, datatype=name, group_size=g)
            exp = orc.quantize_table(raw, odt, fused.MX_TABLES[name], orc.TABLE_MX, g)
            _same(_bits(d), exp["deq"], what + " deq")
        else:
            if dtn != "f16":
                continue
            b, kindname = int(rng.choice([3, 4])), MIXED[rng.integers(4)]
            g = None if gs in (None, -1, 0) else gs
            what += f"/{b}/{kindname}/sg={g}"
            d = qw.search_datatype(dev, wq_bits=b, datatype=kindname, group_size=g)
            tables = [fused._SEARCH_VALUES[t] for t in fused.MIXED_LISTS[(b, kindname)]]
            exp = orc.search_tables(raw.view(np.uint16), tables, g)
            _same(_bits(d), exp["deq"], what + " deq")
        n += 1
    return n


if __name__ == "__main__":
    t0 = time.time()
    n = run(float(sys.argv[1]), int(sys.argv[2]))
    print("ok:", n, "random cases in", round(time.time() - t0, 1), "s, seed", sys.argv[2], flush=True)
