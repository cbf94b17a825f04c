"""Randomised differential run of the CPU oracle against the IMPORTED REFERENCE (/root/reference, build container
only -- nothing under tests/ that runs on the GPU box loads this file): quant_int / quant_int_asym /
_int_quant_meta / quant_datatype / _fp_quant_meta / quant_mx / search_datatype on random shapes, groupings,
dtypes and contents, every output bit for bit (a NaN equals a NaN; the sign of a zero zero-point is left out for
segments of 64 elements or more, where torch.amin's choice between +0 and -0 depends on the CPU's vector lanes --
below that it is the first zero in memory, and compared).  argv: seconds, seed."""
import contextlib, io, os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference"); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np, torch
from quant_utils import quant_weight as qw
from oracle import oracle as orc
from quantization_amd import fused   # value lists only (no device call)
orc.lib()
budget, seed = float(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
def bits(t):
    t = t.contiguous()
    if t.dtype == torch.float32: return t.numpy().reshape(-1)
    return t.view(torch.int16).numpy().view(np.uint16).reshape(-1)
def same(a, b, what, zero_sign=False):
    a, b = np.asarray(a).reshape(-1), np.asarray(b).reshape(-1)
    if a.dtype == np.uint16:
        an, bn = (a & 0x7fff) > 0x7c00, (b & 0x7fff) > 0x7c00
        if zero_sign:
            a, b = a.copy(), b.copy(); a[a == 0x8000] = 0; b[b == 0x8000] = 0
        ok = np.array_equal(an, bn) and np.array_equal(a[~an], b[~bn])
    else:
        ok = np.array_equal(a, b, equal_nan=True) and np.array_equal(np.signbit(a[a == 0]), np.signbit(b[b == 0]))
    if not ok:
        bad = np.nonzero(a != b)[0]
        raise AssertionError(f"{what}: {bad.size} mismatches, first {[(i, a[i], b[i]) for i in bad[:5]]}")
INTS = [("int8", 8, False), ("int8", 8, True), ("int4", 4, False), ("int4", 4, True), ("int2", 2, False), ("int3", 3, True), ("int6", 6, False)]
LUTS = [("fp8_e4m3", 8), ("fp8_e3m4", 8), ("fp8_e2m5", 8), ("fp6_e3m2", 6), ("fp6_e2m3", 6), ("fp5_e2m2", 5), ("fp4", 4)]
LISTS = [("fp3", 3), ("flint4", 4), ("fp4_er_pos", 4), ("fp4_ea_neg", 4), ("fp3_er_neg", 3), ("fp3_ea_pos", 3)]
MX = [("mx_int3", 3), ("mx_fp3", 3), ("mx_int4", 4), ("mx_fp4", 4)]
MIXED = ["mixed_bitmod", "mixed_er", "mixed_ea", "mixed_ant"]
t0, n, per = time.time(), 0, {}
while time.time() - t0 < budget:
    fam = ["int", "int", "lut", "list", "mx", "search", "e5m2"][rng.integers(7)]
    rows = int(rng.integers(1, 24))
    small = fam in ("lut", "list", "e5m2", "search")          # the reference's table paths are slow
    cols = int(rng.choice([int(rng.integers(1, 300)), 8 * int(rng.integers(1, 80)), 512, 1024] if small else
                          [int(rng.integers(1, 700)), 8 * int(rng.integers(1, 300)), 1024, 4096]))
    divs = [d for d in (1, 2, 3, 4, 5, 8, 16, 24, 32, 40, 64, 96, 128, 256, 512) if cols % d == 0]
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
    if kind == 6:   # non-negative rows with zeros of both signs: the sign of a zero zero-point
        w = w.abs()
        zs = torch.from_numpy(rng.random((rows, cols)) < 0.3)
        w[zs] = 0.0
        w[zs & torch.from_numpy(rng.random((rows, cols)) < 0.5)] = -0.0
    wt = {"f16": w.half(), "bf16": w.bfloat16(), "f32": w}[dtn]
    raw = wt.numpy() if dtn == "f32" else wt.contiguous().view(torch.int16).numpy().view(np.uint16)
    raw = raw if dtn != "f16" else raw.view(np.float16)
    odt = {"f16": orc.F16, "bf16": orc.BF16, "f32": orc.F32}[dtn]
    what = f"[{n}] {fam}/{dtn}/{rows}x{cols}/g={gs}/scale={scale}/kind={kind}"
    with np.errstate(all="ignore"), contextlib.redirect_stdout(io.StringIO()):
        if fam == "int":
            name, b, asym = INTS[rng.integers(len(INTS))]
            what += f"/{name}{'_asym' if asym else ''}"
            d, q, s, zp = qw._int_quant_meta(wt, wq_bits=b, asym=asym, group_size=gs)
            d2 = (qw.quant_int_asym if asym else qw.quant_int)(wt, wq_bits=b, group_size=gs)
            exp = orc.quantize(raw, odt, fmt=orc.INT_ASYM if asym else orc.INT_SYM, bits=b, group_size=gs)
            same(bits(d), exp["deq"], what + " deq"); same(bits(d2), exp["deq"], what + " deq(quant_int)")
            same(bits(s), exp["scale"], what + " scale")
            L_ = rows * cols if gs == -1 else (gs if gs and gs > 0 else cols)
            if asym: same(bits(zp), exp["zp"], what + " zp", zero_sign=L_ >= 64)
            nanseg = np.isnan(exp["deq"].view(np.float16).astype(np.float32)).reshape(-1)
            qb = q.contiguous().view(torch.uint8).numpy().reshape(-1)
            assert ((qb == exp["q"].reshape(-1)) | nanseg).all(), what + " codes"
        elif fam == "lut":
            name, b = LUTS[rng.integers(len(LUTS))]
            what += f"/{name}"
            d, q, s = qw._fp_quant_meta(wt, wq_bits=b, datatype=name, group_size=gs)
            d2 = qw.quant_datatype(wt, wq_bits=b, datatype=name, group_size=gs)
            exp = orc.quantize(raw, odt, fmt=orc.LUT, bits=b, datatype=name, group_size=gs)
            same(bits(d), exp["deq"], what + " deq"); same(bits(q), exp["q"], what + " q"); same(bits(s), exp["scale"], what + " scale")
            if gs == -1:
                exp = orc.quantize(raw, odt, fmt=orc.LUT, bits=b, datatype=name, group_size=gs, lut_f32_path=True)
            same(bits(d2), exp["deq"], what + " deq(quant_datatype)")
        elif fam == "list":
            name, b = LISTS[rng.integers(len(LISTS))]
            what += f"/{name}"
            d, q, s = qw._fp_quant_meta(wt, wq_bits=b, datatype=name, group_size=gs)
            exp = orc.quantize_table(raw, odt, fused.TABLE_LISTS[name], orc.TABLE_F16, gs)
            same(bits(d), exp["deq"], what + " deq"); same(bits(q), exp["q"], what + " q"); same(bits(s), exp["scale"], what + " scale")
            d2 = qw.quant_datatype(wt, wq_bits=b, datatype=name, group_size=gs)
            if gs == -1:
                exp = orc.quantize_table(raw, odt, fused.TABLE_LISTS[name], orc.TABLE_F32, gs)
            same(bits(d2), exp["deq"], what + " deq(quant_datatype)")
        elif fam == "e5m2":
            d, q, s = qw._fp_quant_meta(wt, wq_bits=8, datatype="fp8_e5m2", group_size=gs)
            exp = orc.quantize_table(raw, odt, fused.TABLE_LISTS["fp8_e5m2"],
                                     orc.TABLE_F32, gs, deq_f32=True, neg_inf_zero=True)
            same(d.numpy(), exp["deq"], what + " deq"); same(q.numpy(), exp["q"], what + " q"); same(s.numpy(), exp["scale"], what + " scale")
        elif fam == "mx":
            name, b = MX[rng.integers(len(MX))]
            g = int(rng.choice([d_ for d_ in divs if d_ >= 2] or [cols]))
            what += f"/{name}/mxg={g}"
            d = qw.quant_mx(wt, wq_bits=b, datatype=name, group_size=g)
            exp = orc.quantize_table(raw, odt, fused.MX_TABLES[name], orc.TABLE_MX, g)
            same(bits(d), exp["deq"], what + " deq")
        else:
            if dtn != "f16":
                continue
            b, kindname = int(rng.choice([3, 4])), MIXED[rng.integers(4)]
            g = None if gs in (None, -1, 0) else gs
            what += f"/{b}/{kindname}/sg={g}"
            d = qw.search_datatype(wt, wq_bits=b, datatype=kindname, group_size=g)
            tables = [fused._SEARCH_VALUES[t] for t in fused.MIXED_LISTS[(b, kindname)]]
            exp = orc.search_tables(raw.view(np.uint16), tables, g)
            same(bits(d), exp["deq"], what + " deq")
    per[fam] = per.get(fam, 0) + 1
    n += 1
print("ok:", n, "random cases", per, "in", round(time.time() - t0, 1), "s, seed", seed, flush=True)
