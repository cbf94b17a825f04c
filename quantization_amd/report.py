"""Per-layer error reports: the per-tensor flow of MSE&P99/quant_error_multi_weight.py:119-164 and
MSE&P99/plot.py:18-99 (reference paths under /root/reference), with the quantization itself fused in
instead of read back from two ONNX files, and the same CSV schemas."""
import csv
import os
from typing import Dict, Iterable, List, Optional, Tuple

import torch

from . import _lib
from .error_metrics import error_stats, percentile, percentile_rows, true_div
from .fused import LUT_FORMATS, TABLE_LISTS, quantize, quantize_batch, quantize_table

MULTI_COLUMNS = ["initializer", "datatype", "granularity", "bits", "mse_tensor", "mse_rep", "mse_p99"]
SINGLE_COLUMNS = ["datatype", "granularity", "bits", "num_channels", "groups_per_channel", "mse_tensor", "mse_rep",
                  "mse_p99"]
# columns of the per-layer record tensor produced on the device
RECORD_FIELDS = ["numel", "sse", "sum_abs", "mse_rep", "mse_p99", "abs_p99"]


def format_code(datatype: str) -> Tuple[int, Optional[str]]:
    """'int8' / 'int4_asym' / 'fp8_e4m3' -> (mq_format, table name); the reference dispatches on the same
    substrings (quant_weight.py:653-706)."""
    if datatype.startswith("int"):
        return (_lib.INT_ASYM if "asym" in datatype else _lib.INT_SYM), None
    if datatype in LUT_FORMATS or (datatype in TABLE_LISTS and datatype != "fp8_e5m2"):
        return _lib.LUT_FP, datatype   # explicit-list tables go through mq_quantize_table (layer_record)
    raise ValueError(f"Unsupported datatype {datatype}")


def granularity(group_size) -> str:
    return "per-group" if (group_size is not None and group_size > 0) else "per-channel"


def quant_dir_name(bits: int, group_size) -> str:
    """w_<bits>_gs_<gs|none>: the directory level export_and_quant.py:52-55 writes and parse_quant_config reads."""
    return f"w_{bits}_gs_{group_size if (group_size is not None and group_size > 0) else 'none'}"


def parse_quant_config(model_path: str):
    """(grouping, datatype, bits, group_size) from .../<per-channel|per-group>/<dtype>/w_<b>_gs_<g>/model.onnx
    -- quant_error_multi_weight.py:41-52."""
    parts = model_path.split(os.sep)
    for i, p in enumerate(parts):
        if p in ("per-channel", "per-group"):
            _, bits, _, gs = parts[i + 2].split("_")
            return p, parts[i + 1], int(bits), (None if gs == "none" else int(gs))
    return "baseline", os.path.basename(os.path.dirname(model_path)), None, None


@torch.no_grad()
def layer_record(w: torch.Tensor, datatype: str, bits: int, group_size, form: str = "single",
                 abs_p99: bool = True, scratch: Optional[dict] = None) -> torch.Tensor:
    """One weight -> float64[len(RECORD_FIELDS)] on the device, nothing synchronised.

    form="single": the per-vector MSE list follows the quantization segments (rows / (row, group)), taken
    from the fused pass for free.  form="multi": columns / column blocks of the stored tensor, which needs
    the dequantized tensor and a second, column-wise pass."""
    fmt, table = format_code(datatype)
    K, C = w.shape
    L = C if granularity(group_size) == "per-channel" else group_size
    scratch = scratch if scratch is not None else {}
    single = form == "single"
    if table in TABLE_LISTS:
        res = quantize_table(w, TABLE_LISTS[table], _lib.TABLE_F16, group_size, deq=not single, codes=False,
                             scale=False, abs_err=torch.float32 if abs_p99 else None, seg_mse=single, totals=True)
    else:
        res = quantize(w, fmt, bits, group_size, table, deq=not single, codes=False, scale=False,
                       abs_err=torch.float32 if abs_p99 else None, seg_mse=single, totals=True,
                       out=scratch.get("res"))
        scratch["res"] = res
    rec = torch.zeros(len(RECORD_FIELDS), dtype=torch.float64, device=w.device)
    rec[0] = w.numel()
    rec[1:3] = res.totals
    if single:
        # float32 list entries, float64 percentile arithmetic; the list's mean is the tensor MSE (see model_records)
        rec[3] = res.totals[0] / w.numel()
        rec[4] = percentile(res.seg_mse, 99, wide=True)
    else:
        col = error_stats(w.half() if w.dtype != torch.float16 else w, res.deq, cols=True, totals=False)["col_sse"]
        per_vec = true_div(col, K) if L == C else true_div(col.reshape(C // L, L).sum(dim=1), K * L)
        rec[3] = per_vec.mean()
        rec[4] = percentile(per_vec, 99)
    if abs_p99:
        rec[5] = percentile(res.abs_err, 99).double()
    return rec


@torch.no_grad()
def model_records(weights, datatype: str, bits: int, group_size, abs_p99: bool = True, pct: float = 99.0,
                  max_err_bytes: int = 8 << 30, return_vecs: bool = False, numpy_order: bool = False,
                  return_mse_tensor: bool = False):
    """layer_record(form="single") for every weight of a model at once: float64 [len(weights), len(RECORD_FIELDS)]
    on the device, nothing synchronised.

    The counterpart of the per-initializer loop of quant_error_multi_weight.py:133-151 with the quantization
    of quant_weight.py:660-704 fused in: layers are grouped by (shape, dtype) -- a transformer has a handful --
    and every group goes through ONE fused launch per 32 tensors (their pointers, no stacking copy); with
    abs_p99 the P99 of |err| is selected inside that launch.  max_err_bytes bounds the float32 |err| scratch
    of one launch.  return_vecs: also return the per-vector MSE lists (float32, one per weight, in the order of
    `weights`) -- what a model-wide percentile over several GPUs is taken from (distributed.joint_percentile with
    wide=True).

    The per-vector list holds float32 values, as the scripts' does (compute_mse returns float(np.float32 mean),
    quant_error_single_weight.py:46-48): the kernels write float32(sum / L) per segment, the P99 is selected on the
    float32 bits and interpolated in float64 like np.percentile of the float64 array of those values (:79).
    mse_rep (:78, the mean of the list) is the tensor's squared-error sum over its element count: every vector of
    a weight has the same length, so the two are the same number up to the float32 rounding of the list entries
    (6e-8 relative), and no pass over the list is needed for it.

    numpy_order (abs_p99=False; float16 / bfloat16 weights, per-channel or per-group with segments of a multiple of
    8 and at most 32768 elements): the list entries are summed in numpy's order -- pairwise, in chunks of 8192
    elements -- and have the bits of the scripts' entries, so mse_p99 equals the reference's
    float(np.percentile(vec, 99)) bit for bit.  Groups of 32 / 64 / 128 elements stay in the bandwidth kernels (~T more
    float32 additions per element pair); rows and other group sizes take a wave-per-segment kernel that is several
    times slower.  mse_rep is then the float64 mean of the list in numpy's summation order (numpy_list_means): the
    reference's float(vec.mean()) bit for bit; the tensor sums (numel, sse, sum_abs) stay as above."""
    fmt, table = format_code(datatype)
    weights = list(weights)
    if not weights:
        raise ValueError("model_records() needs at least one weight")
    dev = weights[0].device
    if table in TABLE_LISTS:   # explicit-list tables: layer by layer (their kernel takes one tensor)
        if return_vecs:
            raise ValueError("return_vecs is not available for the explicit-list tables")
        return torch.stack([layer_record(w, datatype, bits, group_size, abs_p99=abs_p99) for w in weights])
    if numpy_order and abs_p99:
        raise ValueError("numpy_order is a statistics-only mode: pass abs_p99=False")
    state = _launch_pass(weights, fmt, table, bits, group_size, abs_p99, pct, max_err_bytes, numpy_order)
    out = _finish_pass(weights, state, abs_p99, pct)
    mse_tensor = None
    if numpy_order:   # mse_rep in numpy's bits: the float64 mean of the list, in numpy's summation order
        vecs = [v for r in state[1] for v in r.seg_mse.unbind(0)]
        rep = numpy_list_means(vecs)
        out[torch.as_tensor(state[0], device=out.device), 3] = rep
    if return_mse_tensor:
        # mse_tensor = float(np.mean((a32 - b32) ** 2)): sse / numel to 1e-6, or -- numpy_order -- numpy's bits
        mse_tensor = out[:, 1] / out[:, 0]
        if numpy_order:
            from .fused import make_cfg
            cfg = make_cfg(fmt, bits, group_size, table)
            m32 = torch.empty(len(weights), dtype=torch.float32, device=dev)
            k = 0
            for r in state[1]:
                for j in range(r.scale.shape[0]):
                    i = state[0][k]
                    tensor_mse_numpy(weights[i], cfg, r.scale[j], None if r.zero_point is None else r.zero_point[j], m32[i])
                    k += 1
            mse_tensor = m32.double()
    extra = []
    if return_vecs:
        vec_rows, keep = state[0], state[1]
        vecs = [v for r in keep for v in r.seg_mse.unbind(0)]
        order = {i: k for k, i in enumerate(vec_rows)}
        extra.append([vecs[order[i]] for i in range(len(weights))])
    if return_mse_tensor:
        extra.append(mse_tensor)
    return (out, *extra) if extra else out


def _launch_pass(weights, fmt, table, bits, group_size, abs_p99, pct, max_err_bytes, numpy_order=False):
    """The fused quantize launches of one pass over a model (current stream): layers grouped by (shape, dtype), one
    launch per 32 of them.  -> (layer index per result row, results, totals, |err| P99s)"""
    groups: Dict[Tuple, List[int]] = {}
    for i, w in enumerate(weights):
        if w.dim() != 2:
            raise ValueError("model_records() takes 2-D weights")
        groups.setdefault((tuple(w.shape), w.dtype), []).append(i)
    vec_rows, keep, tots, p99s = [], [], [], []
    for (shape, _), idx in groups.items():
        K, C = shape
        per = K * C * 4 if abs_p99 else 1
        chunk = max(1, min(32, max_err_bytes // per))
        for lo in range(0, len(idx), chunk):
            sel = idx[lo:lo + chunk]
            # (every chunk keeps its own result: the lists are read by the selection after the last launch)
            res = quantize_batch([weights[i] for i in sel], fmt, bits, group_size, table, deq=False, codes=False,
                                 scale=bool(numpy_order), abs_err=torch.float32 if abs_p99 else None,
                                 seg_mse="numpy" if numpy_order else True, totals=True,
                                 p99=pct if abs_p99 else None)
            tots.append(res.totals)
            if abs_p99:
                p99s.append(res.abs_p99)
                res.abs_err = None
                res._bufs.pop("abs_err", None)   # the |err| scratch of this launch is free for the next one
            keep.append(res)
            vec_rows += sel
    return vec_rows, keep, tots, p99s


def _finish_pass(weights, state, abs_p99, pct):
    """Per-vector P99s of every layer of a pass (one set of select launches) and the record table, on the current
    stream; the launches' small results are gathered with one concatenation and one permutation (no per-launch
    host <-> device traffic)."""
    vec_rows, keep, tots, p99s = state
    dev = weights[0].device
    p99v = percentile_rows([r.seg_mse for r in keep], pct, wide=True)
    # (pinned + non_blocking: a pageable host-to-device copy would make the host wait for every launch above)
    host = torch.tensor([[float(weights[i].numel()), float(i)] for i in vec_rows], dtype=torch.float64,
                        pin_memory=True).to(dev, non_blocking=True)
    numel, tot = host[:, 0], torch.cat(tots)
    mse = tot[:, 0] / numel
    # a list that holds a NaN has a NaN mean; np.percentile of it is NaN too
    p99v = torch.where(torch.isnan(mse), mse, p99v)
    a99 = torch.cat(p99s).double() if abs_p99 else torch.zeros_like(mse)
    rec = torch.stack([numel, tot[:, 0], tot[:, 1], mse, p99v, a99], dim=1)
    out = torch.empty_like(rec)
    out[host[:, 1].long()] = rec            # launch order -> the order of `weights`
    return out


def numpy_list_means(lists, stream=None) -> torch.Tensor:
    """float(np.array(v, dtype=float64).mean()) for every float32 list v of `lists`, bit for bit (numpy's float64
    summation order; mq_list_mean_numpy): the scripts' mse_rep = float(vec.mean()) (quant_error_single_weight.py:78).
    -> float64 [len(lists)] on the lists' device, enqueued on `stream` (default: the current one)."""
    import ctypes
    from .fused import bind_batch
    L = bind_batch()
    lists = list(lists)
    dev = lists[0].device
    out = torch.empty(len(lists), dtype=torch.float64, device=dev)
    st = (stream or torch.cuda.current_stream(dev)).cuda_stream
    by_n: Dict[int, List[int]] = {}
    for i, v in enumerate(lists):
        if v.dtype != torch.float32 or not v.is_contiguous() or v.dim() != 1 or v.numel() == 0:
            raise ValueError("numpy_list_means takes non-empty contiguous 1-D float32 lists")
        by_n.setdefault(v.numel(), []).append(i)
    for n, idx in by_n.items():
        xs = (ctypes.c_void_p * len(idx))(*[lists[i].data_ptr() for i in idx])
        os_ = (ctypes.c_void_p * len(idx))(*[out.data_ptr() + 8 * i for i in idx])
        _lib.check(L.mq_list_mean_numpy(xs, len(idx), n, os_, st))
    return out


def tensor_mse_numpy(w: torch.Tensor, cfg, scale: torch.Tensor, zero_point: Optional[torch.Tensor], out: torch.Tensor,
                     stream=None) -> None:
    """out[()] (float32, device) = np.mean((a32 - b32) ** 2) over the whole weight, bit for bit: compute_mse of
    quant_error_multi_weight.py:55-58 in numpy's summation order (mq_tensor_mse_numpy).  cfg: the mq_quant_cfg of the
    pass; scale / zero_point: the float16 segment parameters a statistics pass wrote for this weight."""
    from .fused import _DTYPES, bind_batch, workspace
    L = bind_batch()
    K, C = w.shape
    st = stream or torch.cuda.current_stream(w.device)
    with torch.cuda.stream(st):
        ws, _ = workspace(w.device, L.mq_tensor_mse_numpy_workspace_bytes(K, C))
    _lib.check(L.mq_tensor_mse_numpy(w.data_ptr(), _DTYPES[w.dtype], K, C, cfg, scale.data_ptr(),
                                     zero_point.data_ptr() if zero_point is not None else None, out.data_ptr(),
                                     ws.data_ptr(), ws.numel(), st.cuda_stream))


_side_streams: Dict = {}   # device index (and, for the third stream, (index, "fin")) -> stream


class SweepPlan:
    """The error-statistics sweep of export_and_quant.sh / quant_error_multi_weight.sh -- every (datatype, bits,
    group_size) of `passes` over every weight of a model -- prepared once for weights that stay resident: launch
    descriptors, output buffers and the selection's pointer tables are built here, run() only enqueues.

    run() -> float64 [len(passes), len(weights), len(RECORD_FIELDS)] records (numel, sse, sum_abs, mse_rep, mse_p99,
    0).  lists(k) -> the float32 per-vector MSE lists of pass k, in the order of `weights`.  Records and lists live in
    buffers of the plan: the next run() overwrites them.

    feed_lists: in per-group passes the list of a large tensor (a quarter of a million entries and more) is NOT
    written -- a sampling launch gives the tensor a bracket around the wanted quantile, the pass counts the entries
    under it and keeps those inside, and the percentile is selected among the kept ones (exact: the sample only decides
    where to look; a bracket that misses makes the repair launch write the list after all).  lists(k) of such a tensor
    then holds nothing (plan.fed(k) tells which).  It removes the list's stores and the selection's re-reads (0.94 GB
    each way per g = 32 pass of Llama-3-8B) but not time: measured on MI355X the pass alone gets 4 - 6 % shorter and
    the whole step stays where it was -- the per-entry bookkeeping costs the issue-bound kernel about what the stores
    did -- so it is off by default.

    Per pass ONE fused launch takes every layer whose segments have one length, whatever the shapes (mq_sweep_run:
    per-group passes the whole model, per-channel passes the layers of one column count; statistics only, float32
    MSE lists); layers the sweep form does not take go in batches of 32 same-shaped ones (mq_quantize_batch;
    single_launch=False: everything does).  The selections of pass k -- chains of short launches that leave most of
    the chip idle -- run on a second HIP stream while the current stream already streams the weights for pass
    k + 1; the streams are joined before run() returns, so the result is ordered on the current stream.
    numpy_order: as for model_records (per-channel passes and group sizes other than 32 / 64 / 128 then go through the
    batched launches and the slower wave-per-segment kernel); mse_rep and mse_p99 of the records and the tensor means
    in `mse_tensor` [len(passes), len(weights)] then have the reference's bits."""

    def __init__(self, weights, passes, pct: float = 99.0, numpy_order: bool = False, single_launch: bool = True,
                 feed_lists: bool = False, fuse_groupings: bool = True):
        import ctypes
        import os
        from .fused import _DTYPES, _MultiItem, _QuantItem, bind_batch, make_cfg
        self.weights = list(weights)
        self.passes = list(passes)
        self.pct = float(pct)
        self.numpy_order = bool(numpy_order)
        if not self.weights or not self.passes:
            raise ValueError("SweepPlan needs at least one weight and one pass")
        dev = self.dev = self.weights[0].device
        groups: Dict[Tuple, List[int]] = {}
        for i, w in enumerate(self.weights):
            if w.dim() != 2 or not w.is_cuda or w.device != dev or w.dtype not in _DTYPES or not w.is_contiguous():
                raise ValueError("SweepPlan takes contiguous 2-D float16 / bfloat16 / float32 ROCm weights on one device")
            groups.setdefault((tuple(w.shape), w.dtype), []).append(i)
        self.order = [i for idx in groups.values() for i in idx]           # row of the tables -> index into weights
        nl = len(self.order)
        self.L = L = bind_batch()
        from . import error_metrics
        error_metrics._bind()
        self.numel = torch.tensor([float(self.weights[i].numel()) for i in self.order], dtype=torch.float64, device=dev)
        self.perm = torch.tensor(self.order, dtype=torch.int64, device=dev)
        self.records = torch.zeros(len(self.passes), nl, len(RECORD_FIELDS), dtype=torch.float64, device=dev)
        # numpy_order: float64 [len(passes), len(weights)], the scripts' mse_tensor bit for bit (filled by run())
        self.mse_tensor = torch.zeros(len(self.passes), nl, dtype=torch.float64, device=dev) if numpy_order else None
        # percentiles: [passes, layers] (row k: pass k's), followed by the results of the families' selection calls (each
        # call writes one contiguous run; p99_index says where (pass, layer)'s value lies)
        npass = len(self.passes)
        self.p99_store = torch.zeros(2 * npass * nl, dtype=torch.float64, device=dev)
        self.p99_all = self.p99_store[:npass * nl].view(npass, nl)
        p99_index = list(range(npass * nl))
        p99_used = npass * nl
        self.totals_all = torch.zeros(npass, nl, 2, dtype=torch.float64, device=dev)
        self.table = None     # set_table(): the records go straight into a table over more layers than this plan holds
        self.ws_bytes = 0
        self.multi_ws_bytes = 0
        self.side_ws_bytes = 0
        self.select_beside = os.environ.get("MQ_SWEEP_SELECT_BESIDE", "1") != "0"   # (measurements either way)
        self.beside_bytes = int(os.environ.get("MQ_SWEEP_BESIDE_MB", "256")) << 20
        self.fin_ws_bytes = 1 << 16
        self.fin_ws = None
        # numpy order: every finishing call writes its tensors' mse_rep / mse_tensor into a run of these two buffers;
        # finish() moves them to their rows with one scatter each (np_dst: flat (pass, weight) index per entry)
        self.np_rep = torch.empty(len(self.passes) * nl, dtype=torch.float64, device=dev) if numpy_order else None
        self.np_mt = torch.empty(len(self.passes) * nl, dtype=torch.float32, device=dev) if numpy_order else None
        np_dst: List[int] = []

        def np_slot(k, weight_ids):
            a = len(np_dst)
            np_dst.extend(k * nl + int(i) for i in weight_ids)
            return self.np_rep[a:len(np_dst)], self.np_mt[a:len(np_dst)]
        self.plan = []
        if os.environ.get("MQ_SWEEP_SINGLE", "1") == "0":   # (measurements of the per-shape launches)
            single_launch = False
        if os.environ.get("MQ_SWEEP_FEED") in ("0", "1"):   # (measurements either way)
            feed_lists = os.environ["MQ_SWEEP_FEED"] == "1"
        if os.environ.get("MQ_SWEEP_FUSE") in ("0", "1"):   # (measurements either way)
            fuse_groupings = os.environ["MQ_SWEEP_FUSE"] == "1"
        # Passes that differ only in the grouping -- per row, groups of 32 / 64 / 128 -- form a family: the layers the
        # several-groupings launch takes (float16 / bfloat16, a multiple of 128 columns up to 32768) are read ONCE for
        # all of the family's passes (mq_sweep_multi_run, enqueued with the family's first pass); the passes keep their
        # own lists, totals, selections and record rows.
        SLOT = {0: 0, 32: 1, 64: 2, 128: 3}
        self.read_bytes = 0    # weight bytes the quantize launches of one run() read (a family's launch reads a layer once)
        self.list_bytes = 0    # list bytes they write
        self.families = []
        fam_of: Dict[int, dict] = {}
        if fuse_groupings and single_launch and not numpy_order and not feed_lists:
            by_fmt: Dict[Tuple, Dict[int, int]] = {}
            for k, (datatype, bits, group_size) in enumerate(self.passes):
                gs0 = 0 if group_size is None or (group_size <= 0 and group_size != -1) else group_size
                if gs0 in SLOT and format_code(datatype)[1] not in TABLE_LISTS:
                    by_fmt.setdefault((datatype, bits), {}).setdefault(SLOT[gs0], k)
            for slots in by_fmt.values():
                if len(slots) >= 2:
                    fam = dict(slots=slots, first=min(slots.values()), pos={})
                    self.families.append(fam)
                    for k in slots.values():
                        fam_of[k] = fam

        def fusable(pos):
            w = self.weights[self.order[pos]]
            C = w.shape[1]
            return (w.dtype in (torch.float16, torch.bfloat16) and C % 128 == 0 and 128 <= C <= 32768
                    and w.data_ptr() % 16 == 0 and w.numel() * 2 <= 0xF0000000)
        for k_pass, (datatype, bits, group_size) in enumerate(self.passes):
            fmt, table = format_code(datatype)
            if table in TABLE_LISTS:
                raise ValueError("SweepPlan: the explicit-list tables go layer by layer (model_records)")
            cfg = make_cfg(fmt, bits, group_size, table)
            gs = cfg.group_size
            totals = self.totals_all[k_pass]
            p99 = self.p99_all[k_pass]
            # the per-vector MSE lists of the pass: one float32 buffer, every list 32-byte aligned
            ns, offs, total = [], [], 0
            for i in self.order:
                K, C = self.weights[i].shape
                Lseg = K * C if gs == -1 else (C if gs <= 0 else gs)
                if gs > 0 and C % gs:
                    raise RuntimeError(f"shape '[{K}, {C // gs}, {gs}]' is invalid for input of size {K * C}")
                ns.append(K * C // Lseg)
                offs.append(total)
                total += (ns[-1] + 7) // 8 * 8
            mse = torch.empty(total, dtype=torch.float32, device=dev)
            lists = [None] * len(self.weights)
            for pos, i in enumerate(self.order):
                lists[i] = mse[offs[pos]:offs[pos] + ns[pos]]
            ptrs = [mse.data_ptr() + o * 4 for o in offs]
            # numpy_order: the segments' scales (and zero-points) are kept -- the tensor mean in numpy's order reads them
            scl = torch.empty(total, dtype=torch.float16, device=dev) if numpy_order else None
            zpt = torch.empty(total, dtype=torch.float16, device=dev) if numpy_order and fmt == _lib.INT_ASYM else None

            def make_items(poss, params=True):
                items = (_QuantItem * len(poss))()
                for j, pos in enumerate(poss):
                    items[j].w = self.weights[self.order[pos]].data_ptr()
                    o = items[j].out
                    o.abs_err_dtype = _lib.F32
                    o.seg_mse_f32 = 2 if numpy_order else 1   # (2: entries in numpy's summation order)
                    o.seg_sse = ptrs[pos]
                    if os.environ.get("MQ_SWEEP_NOLIST") == "1":   # (measurements: the pass without its lists)
                        o.seg_sse = None
                    o.totals = totals.data_ptr() + pos * 16
                    if scl is not None and params:
                        o.scale = scl.data_ptr() + offs[pos] * 2
                    if zpt is not None and params:
                        o.zero_point = zpt.data_ptr() + offs[pos] * 2
                return items

            # Launches.  Layers whose segments have one length go through ONE launch whatever their shapes (the
            # library's sweep form: per-group passes take the whole model, per-channel passes the layers of one
            # column count); what it does not take -- float32 weights, per-tensor passes, other group sizes --
            # goes in batches of up to 32 same-shaped layers.
            launches = []
            keys: Dict[Tuple, List[int]] = {}
            fam = fam_of.get(k_pass)
            for pos, i in enumerate(self.order):
                w = self.weights[i]
                if fam is not None and fusable(pos):     # (read by the family's launch)
                    fam["pos"].setdefault((w.dtype, w.shape[1]), set()).add(pos)
                    continue
                keys.setdefault((w.dtype, w.shape[1] if gs <= 0 else None), []).append(pos)
            slow_pos = []     # numpy_order: layers whose means come from the per-layer kernels
            for (dtype, _), poss in keys.items():
                n = len(poss)
                rows = (ctypes.c_int64 * n)(*[self.weights[self.order[pos]].shape[0] for pos in poss])
                cols = (ctypes.c_int64 * n)(*[self.weights[self.order[pos]].shape[1] for pos in poss])
                if single_launch and gs != -1 and numpy_order and dtype in (torch.float16, torch.bfloat16):
                    # every mean in numpy's order from ONE bandwidth launch: list entries and the sums of the
                    # flattened tensors' 128-element leaves; the finishing kernels make mse_tensor and mse_rep of them
                    nleaves = sum(self.weights[self.order[pos]].numel() // 128 for pos in poss)
                    leaf = torch.empty(max(64, nleaves), dtype=torch.float32, device=dev)
                    host = torch.empty(L.mq_sweep_numpy_table_bytes(n), dtype=torch.uint8)
                    seg_len, blocks, tl, lc = (ctypes.c_int64(0) for _ in range(4))
                    rc = L.mq_sweep_numpy_prepare(make_items(poss, params=False), rows, cols, n, _DTYPES[dtype],
                                                  ctypes.byref(cfg), leaf.data_ptr(), host.data_ptr(), host.numel(),
                                                  ctypes.byref(seg_len), ctypes.byref(blocks), ctypes.byref(tl),
                                                  ctypes.byref(lc))
                    if rc == _lib.OK:
                        widx = torch.tensor([self.order[pos] for pos in poss], dtype=torch.int64, device=dev)
                        # (where its mse_rep / mse_tensor go is settled once the families have taken their layers)
                        launches.append(("npsweep", host.to(dev), n, seg_len.value, blocks.value, _DTYPES[dtype], leaf,
                                         tl.value, lc.value, widx, None, None))
                        self.ws_bytes = max(self.ws_bytes, L.mq_sweep_workspace_bytes(blocks.value))
                        self.fin_ws_bytes = max(self.fin_ws_bytes,
                                                L.mq_sweep_numpy_finish_workspace_bytes(tl.value, lc.value))
                        continue
                    if rc != _lib.ERR_UNSUPPORTED:
                        _lib.check(rc)
                slow_pos += poss
                if single_launch and gs != -1:
                    items = make_items(poss)
                    host = torch.empty(L.mq_sweep_table_bytes(n), dtype=torch.uint8)
                    seg_len, blocks = ctypes.c_int64(0), ctypes.c_int64(0)
                    # the lists of large tensors are not written: their percentile is selected inside the pass
                    # (a sampling launch gives each a bracket, the pass counts below it and keeps what lies inside)
                    nfed, fws_b, aux_b = ctypes.c_int32(0), ctypes.c_size_t(0), ctypes.c_size_t(0)
                    if feed_lists and gs > 0 and not numpy_order:
                        _lib.check(L.mq_sweep_feed_sizes(rows, cols, n, ctypes.byref(cfg), self.pct, ctypes.byref(nfed),
                                                         ctypes.byref(fws_b), ctypes.byref(aux_b)))
                    if nfed.value > 0:
                        fws = torch.empty(fws_b.value, dtype=torch.uint8, device=dev)
                        aux = torch.empty(aux_b.value, dtype=torch.uint8)
                        tab_dev = torch.empty(host.numel(), dtype=torch.uint8, device=dev)   # (the brackets land in it)
                        sblocks = ctypes.c_int64(0)
                        fed_rows = (ctypes.c_int32 * n)()
                        rc = L.mq_sweep_feed_prepare(items, rows, cols, n, _DTYPES[dtype], ctypes.byref(cfg), self.pct,
                                                     fws.data_ptr(), fws.numel(), tab_dev.data_ptr(), host.data_ptr(),
                                                     host.numel(), aux.data_ptr(), aux.numel(), ctypes.byref(seg_len),
                                                     ctypes.byref(blocks), ctypes.byref(sblocks), ctypes.byref(nfed),
                                                     fed_rows)
                        if rc == _lib.OK:
                            frows = [fed_rows[j] for j in range(nfed.value)]
                            # (table, tensors, segment length, workgroups, dtype, aux, fed tensors, sampling workgroups,
                            #  feed workspace, fed rows of the table, their positions in the pass, results by table row)
                            tab_dev.copy_(host)
                            launches.append(("feedsweep", tab_dev, n, seg_len.value, blocks.value, _DTYPES[dtype],
                                             aux.to(dev), nfed.value, sblocks.value, fws,
                                             torch.tensor(frows, dtype=torch.int64, device=dev),
                                             torch.tensor([poss[r] for r in frows], dtype=torch.int64, device=dev),
                                             torch.zeros(n, dtype=torch.float64, device=dev)))
                            self.ws_bytes = max(self.ws_bytes, L.mq_sweep_workspace_bytes(blocks.value))
                            continue
                        if rc != _lib.ERR_UNSUPPORTED:
                            _lib.check(rc)
                    rc = L.mq_sweep_prepare(items, rows, cols, n, _DTYPES[dtype], ctypes.byref(cfg), host.data_ptr(),
                                            host.numel(), ctypes.byref(seg_len), ctypes.byref(blocks))
                    if rc == _lib.OK:
                        launches.append(("sweep", host.to(dev), n, seg_len.value, blocks.value, _DTYPES[dtype],
                                         2 if numpy_order else 1))
                        self.ws_bytes = max(self.ws_bytes, L.mq_sweep_workspace_bytes(blocks.value))
                        continue
                    if rc != _lib.ERR_UNSUPPORTED:
                        _lib.check(rc)
                by_shape: Dict[Tuple, List[int]] = {}
                for pos in poss:
                    by_shape.setdefault(tuple(self.weights[self.order[pos]].shape), []).append(pos)
                for (K, C), same in by_shape.items():
                    for lo in range(0, len(same), 32):
                        sel = same[lo:lo + 32]
                        self.ws_bytes = max(self.ws_bytes,
                                            L.mq_quantize_batch_workspace_bytes(len(sel), K, C, ctypes.byref(cfg)))
                        launches.append(("batch", make_items(sel), len(sel), _DTYPES[dtype], K, C))
            self.read_bytes += sum(self.weights[self.order[pos]].numel() * self.weights[self.order[pos]].element_size()
                                   for poss in keys.values() for pos in poss)
            params = None
            if numpy_order:
                params = [(scl[offs[pos]:offs[pos] + ns[pos]],
                           None if zpt is None else zpt[offs[pos]:offs[pos] + ns[pos]]) for pos in range(nl)]
            # which lists are written (and selected from as always): all but the fed tensors of a feeding launch
            fed = torch.zeros(nl, dtype=torch.bool)
            for la in launches:
                if la[0] == "feedsweep":
                    fed[la[11].cpu()] = True
            # (the lists a family's launch writes are selected from right behind that launch, chunk by chunk)
            in_family = set().union(*fam["pos"].values()) if fam is not None and fam["pos"] else set()
            written = [pos for pos in range(nl) if not bool(fed[pos]) and pos not in in_family]
            sel = dict(ptrs=(ctypes.c_void_p * nl)(*ptrs), ns=(ctypes.c_int64 * nl)(*ns), n=nl, idx=None, tmp=None)
            if len(written) < nl:
                sel = dict(ptrs=(ctypes.c_void_p * len(written))(*[ptrs[pos] for pos in written]),
                           ns=(ctypes.c_int64 * len(written))(*[ns[pos] for pos in written]), n=len(written),
                           idx=torch.tensor(written, dtype=torch.int64, device=dev),
                           tmp=torch.zeros(max(1, len(written)), dtype=torch.float64, device=dev))
            self.list_bytes += sum(4 * ns[pos] for pos in range(nl) if not bool(fed[pos]))
            self.plan.append(dict(cfg=cfg, totals=totals, p99=p99, launches=launches, lists=lists, mse=mse, params=params,
                                  sel=sel, slow_pos=slow_pos, fed=fed, ptrs=ptrs, ns=ns, written=written, np_chunks=[]))
        # The families' launches, attached to the family's first pass: per (dtype, column count), in chunks of layers --
        # the selections over a chunk's lists (every pass of the family) are enqueued right behind its launch and run
        # on the second stream beside the next chunk's, so only the last chunk's selections follow the last launch.
        for fam in self.families:
            mask = sum(1 << c for c in fam["slots"])
            first = self.plan[fam["first"]]
            nbytes = lambda pos: self.weights[self.order[pos]].numel() * 2
            fam_bytes = sum(nbytes(pos) for poss in fam["pos"].values() for pos in poss)
            # (every chunk costs the host a launch and a selection call: a rank's small share goes in few chunks)
            chunk_bytes = max(int(os.environ.get("MQ_SWEEP_CHUNK_MB", "768")) << 20,
                              fam_bytes // int(os.environ.get("MQ_SWEEP_CHUNKS", "8")))
            keys = sorted(fam["pos"].items(), key=lambda kv: -sum(nbytes(pos) for pos in kv[1]))
            for ki, ((dtype, C), poss) in enumerate(keys):
                poss = sorted(poss, key=lambda pos: -nbytes(pos))
                chunks, cur, acc = [], [], 0
                for pos in poss:
                    cur.append(pos)
                    acc += nbytes(pos)
                    if acc >= chunk_bytes:
                        chunks.append(cur)
                        cur, acc = [], 0
                if cur:
                    chunks.append(cur)
                tail_bytes = int(os.environ.get("MQ_SWEEP_TAIL_MB", "256")) << 20
                if ki == len(keys) - 1 and len(chunks[-1]) > 1 and sum(nbytes(pos) for pos in chunks[-1]) > 2 * tail_bytes:
                    # the family's last chunk is a small one: its selections are what follows the last launch
                    cut, acc = 0, 0
                    while cut < len(chunks[-1]) - 1 and (cut == 0 or acc + nbytes(chunks[-1][-cut - 1]) <= tail_bytes):
                        acc += nbytes(chunks[-1][-cut - 1])
                        cut += 1
                    chunks[-1], tail = chunks[-1][:-cut], chunks[-1][-cut:]
                    chunks.append(tail)
                for chunk in chunks:
                    n = len(chunk)
                    self.read_bytes += sum(nbytes(pos) for pos in chunk)
                    items = (_MultiItem * n)()
                    for j, pos in enumerate(chunk):
                        items[j].w = self.weights[self.order[pos]].data_ptr()
                        for c, k in fam["slots"].items():
                            items[j].list[c] = self.plan[k]["ptrs"][pos]
                            items[j].totals[c] = self.plan[k]["totals"].data_ptr() + pos * 16
                    rows = (ctypes.c_int64 * n)(*[self.weights[self.order[pos]].shape[0] for pos in chunk])
                    host = torch.empty(L.mq_sweep_multi_table_bytes(n), dtype=torch.uint8)
                    blocks = ctypes.c_int64(0)
                    _lib.check(L.mq_sweep_multi_prepare(items, rows, C, n, _DTYPES[dtype], ctypes.byref(first["cfg"]), mask,
                                                        host.data_ptr(), host.numel(), ctypes.byref(blocks)))
                    # ONE selection call for the chunk's lists of every pass of the family (group lists first: the
                    # short per-row lists take another route and form their own run), one scatter of the results
                    members = sorted(fam["slots"].values(), key=lambda k: -self.plan[k]["ns"][chunk[0]])
                    m = n * len(members)
                    job = dict(ptrs=(ctypes.c_void_p * m)(*[self.plan[k]["ptrs"][pos] for k in members for pos in chunk]),
                               ns=(ctypes.c_int64 * m)(*[self.plan[k]["ns"][pos] for k in members for pos in chunk]), n=m,
                               out=self.p99_store[p99_used:p99_used + m],
                               bytes=sum(nbytes(pos) for pos in chunk))
                    for j, (k, pos) in enumerate((k, pos) for k in members for pos in chunk):
                        p99_index[k * nl + pos] = p99_used + j
                    p99_used += m
                    first["launches"].append(("multisweep", host.to(dev), n, C, blocks.value, _DTYPES[dtype], mask, job))
                    self.side_ws_bytes = max(self.side_ws_bytes, L.mq_percentile_batch_workspace_bytes(min(m, 64)))
                    # (a workspace of its own per chunk: the totals launch of chunk i runs on the second stream, beside
                    #  the statistics launch of chunk i + 1)
                    job["ws_off"] = self.multi_ws_bytes
                    job["ws_len"] = L.mq_sweep_multi_workspace_bytes(blocks.value)
                    self.multi_ws_bytes += (job["ws_len"] + 255) // 256 * 256
                    self.ws_bytes = max(self.ws_bytes, self.multi_ws_bytes)
        # numpy order: the passes of one format with groups of 32 / 64 / 128 share ONE read too (mq_sweep_numpy_run_groups:
        # their launch tables list the same tensors in the same order); each pass keeps its own finishing launches
        self.np_families = 0
        if numpy_order and fuse_groupings and single_launch:
            by_fmt: Dict[Tuple, Dict[int, int]] = {}
            for k, (datatype, bits, group_size) in enumerate(self.passes):
                if group_size in (32, 64, 128):
                    by_fmt.setdefault((datatype, bits), {}).setdefault(group_size, k)
            for sizes in by_fmt.values():
                if len(sizes) < 2:
                    continue
                first = min(sizes.values())
                per_dt: Dict[int, Dict[int, tuple]] = {}
                for gsz, k in sizes.items():
                    for j, la in enumerate(self.plan[k]["launches"]):
                        if la[0] == "npsweep" and la[3] == gsz:
                            per_dt.setdefault(la[5], {})[gsz] = (k, j, la)
                for dt, found in per_dt.items():
                    las = [v[2] for v in found.values()]
                    if len(found) < 2 or any(la[2] != las[0][2] or la[4] != las[0][4] or
                                             not torch.equal(la[9], las[0][9]) for la in las):
                        continue
                    ids = [int(i) for i in las[0][9].cpu().tolist()]
                    wbytes = sum(self.weights[i].numel() * 2 for i in ids)
                    self.read_bytes -= (len(found) - 1) * wbytes
                    self.np_families += 1
                    pos_of = {i: pos for pos, i in enumerate(self.order)}
                    poss = [pos_of[i] for i in ids]
                    if any(b != a + 1 for a, b in zip(poss, poss[1:])):
                        # (layers of another dtype in between: one launch over all of them, the passes' own tables)
                        tabs = [found[g][2][1] if g in found else None for g in (32, 64, 128)]
                        n, blocks = las[0][2], las[0][4]
                        for gsz, (k, j, la) in found.items():
                            self.plan[k]["launches"][j] = ("npsweep_fused",) + tuple(la[1:])
                        self.plan[first]["launches"].insert(0, ("npgroups", tabs, n, blocks, dt))
                        self.ws_bytes = max(self.ws_bytes, L.mq_sweep_numpy_groups_workspace_bytes(blocks))
                        continue
                    # In chunks of layers, like the families of the default mode: a chunk's selections and finishing
                    # launches (every member pass's) run beside the next chunk's launch.  A chunk is a run of
                    # consecutive table rows, so a selection call writes straight into the pass's percentiles.
                    nbytes = lambda pos: self.weights[self.order[pos]].numel() * 2
                    chunk_bytes = max(int(os.environ.get("MQ_SWEEP_CHUNK_MB", "768")) << 20,
                                      wbytes // int(os.environ.get("MQ_SWEEP_CHUNKS", "8")))
                    chunks, cur, acc = [], [], 0
                    for pos in poss:
                        cur.append(pos)
                        acc += nbytes(pos)
                        if acc >= chunk_bytes:
                            chunks.append(cur)
                            cur, acc = [], 0
                    if cur:
                        chunks.append(cur)
                    # (the chunk with the largest tensor first: its finishing chain is the longest, see launch())
                    chunks.sort(key=lambda ch: -max(nbytes(pos) for pos in ch))
                    fused = []
                    for chunk in chunks:
                        n = len(chunk)
                        rows = (ctypes.c_int64 * n)(*[self.weights[self.order[pos]].shape[0] for pos in chunk])
                        cols = (ctypes.c_int64 * n)(*[self.weights[self.order[pos]].shape[1] for pos in chunk])
                        widx = torch.tensor([self.order[pos] for pos in chunk], dtype=torch.int64, device=dev)
                        nleaves = sum(self.weights[self.order[pos]].numel() // 128 for pos in chunk)
                        tabs, jobs, nblocks = {}, [], 0
                        for gsz, (k, j, la) in sorted(found.items()):
                            ps = self.plan[k]
                            items = (_QuantItem * n)()
                            for jj, pos in enumerate(chunk):
                                items[jj].w = self.weights[self.order[pos]].data_ptr()
                                items[jj].out.abs_err_dtype = _lib.F32
                                items[jj].out.seg_mse_f32 = 2
                                items[jj].out.seg_sse = ps["ptrs"][pos]
                                items[jj].out.totals = ps["totals"].data_ptr() + pos * 16
                            leaf = torch.empty(max(64, nleaves), dtype=torch.float32, device=dev)
                            host = torch.empty(L.mq_sweep_numpy_table_bytes(n), dtype=torch.uint8)
                            seg_len, blocks, tl, lc = (ctypes.c_int64(0) for _ in range(4))
                            _lib.check(L.mq_sweep_numpy_prepare(items, rows, cols, n, dt, ctypes.byref(ps["cfg"]),
                                                                leaf.data_ptr(), host.data_ptr(), host.numel(),
                                                                ctypes.byref(seg_len), ctypes.byref(blocks),
                                                                ctypes.byref(tl), ctypes.byref(lc)))
                            tab = host.to(dev)
                            rep, mt = np_slot(k, [self.order[pos] for pos in chunk])
                            sel = dict(ptrs=(ctypes.c_void_p * n)(*[ps["ptrs"][pos] for pos in chunk]),
                                       ns=(ctypes.c_int64 * n)(*[ps["ns"][pos] for pos in chunk]), n=n,
                                       out=self.p99_all[k][chunk[0]:chunk[0] + n])
                            jobs.append((k, sel, (tab, n, seg_len.value, leaf, tl.value, lc.value, rep, mt)))
                            ps["np_chunks"].append((widx, rep, mt))   # (kept for inspection: finish() scatters them)
                            tabs[gsz], nblocks = tab, blocks.value
                            self.fin_ws_bytes = max(self.fin_ws_bytes,
                                                    L.mq_sweep_numpy_finish_workspace_bytes(tl.value, lc.value))
                        fused.append(("npgroups", [tabs.get(g) for g in (32, 64, 128)], n, nblocks, dt, jobs,
                                      sum(nbytes(pos) for pos in chunk)))
                        self.ws_bytes = max(self.ws_bytes, L.mq_sweep_numpy_groups_workspace_bytes(nblocks))
                    taken = set(poss)
                    for gsz, (k, j, la) in found.items():   # the passes' own launches and selections give way
                        ps = self.plan[k]
                        ps["launches"] = [x for x in ps["launches"] if x is not la]
                        written = [pos for pos in ps["written"] if pos not in taken]
                        ps["written"] = written
                        ps["sel"] = dict(ptrs=(ctypes.c_void_p * len(written))(*[ps["ptrs"][pos] for pos in written]),
                                         ns=(ctypes.c_int64 * len(written))(*[ps["ns"][pos] for pos in written]),
                                         n=len(written), idx=torch.tensor(written, dtype=torch.int64, device=dev),
                                         tmp=torch.zeros(max(1, len(written)), dtype=torch.float64, device=dev))
                    self.plan[first]["launches"] = fused + self.plan[first]["launches"]
        self.side_ws_bytes = max(self.side_ws_bytes, self.L.mq_percentile_batch_workspace_bytes(min(nl, 64)))
        for k, ps in enumerate(self.plan):
            for j, la in enumerate(ps["launches"]):
                if la[0] in ("npsweep", "npsweep_fused"):
                    ps["launches"][j] = tuple(la[:10]) + np_slot(k, la[9].cpu().tolist())
        self.np_dst = torch.tensor(np_dst, dtype=torch.int64, device=dev) if np_dst else None
        self.p99_index = torch.tensor(p99_index, dtype=torch.int64, device=dev)
        self.numel_all = self.numel.repeat(npass)
        # row (k, order[pos]) of the records <- source k * nl + pos
        self.set_table(len(self.weights), list(range(len(self.weights))))

    def set_table(self, n_layers: int, layer_ids) -> None:
        """The records of run() become a table over n_layers layers, float64 [passes, n_layers, fields]: weight i's
        rows are the rows layer_ids[i], every other row is written as zeros on each run -- the rank's contribution to
        the all-reduced report table of a sweep whose layers are sharded over GPUs (the default is the plan's own
        weights in their order).  Not in numpy_order mode, whose passes fill their rows one by one."""
        nl, npass = len(self.order), len(self.passes)
        layer_ids = list(layer_ids)
        if len(layer_ids) != len(self.weights) or len(set(layer_ids)) != len(layer_ids) or \
                any(not 0 <= g < n_layers for g in layer_ids):
            raise ValueError("set_table(): one distinct row in [0, n_layers) per weight")
        if self.numpy_order and (n_layers != len(self.weights) or layer_ids != list(range(n_layers))):
            raise ValueError("set_table(): not available in numpy_order mode")
        src = [-1] * (npass * n_layers)
        for k in range(npass):
            for pos, i in enumerate(self.order):
                src[k * n_layers + layer_ids[i]] = k * nl + pos
        self.table_src = torch.tensor(src, dtype=torch.int64, device=self.dev)
        if self.records.shape[1] != n_layers:
            self.records = torch.zeros(npass, n_layers, len(RECORD_FIELDS), dtype=torch.float64, device=self.dev)

    def lists(self, k: int):
        return self.plan[k]["lists"]

    def fed(self, k: int) -> torch.Tensor:
        """bool [len(weights)]: whose list of pass k is selected inside the pass instead of written."""
        out = torch.zeros(len(self.weights), dtype=torch.bool)
        out[torch.tensor(self.order)] = self.plan[k]["fed"]
        return out

    @torch.no_grad()
    def run(self, overlap: bool = True) -> torch.Tensor:
        """overlap=False keeps everything on the current stream (the selections of a pass then run between the
        passes, alone on the chip): for measurements of the quantize launches by themselves."""
        self.launch(overlap)
        return self.finish()

    @torch.no_grad()
    def launch(self, overlap: bool = True) -> None:
        """run() without its last step: every pass is enqueued -- the quantize launches on the current stream, each
        pass's selections and record rows on the second one -- but the current stream does not wait for the second
        one yet.  What the caller enqueues on the current stream before finish() (a model-wide percentile over
        lists(k), a collective) runs beside the last pass's selections: the lists are complete there, the records
        and per-layer percentiles are not."""
        import ctypes
        from .fused import workspace
        dev, L = self.dev, self.L
        nl = len(self.order)
        with torch.cuda.device(dev):
            main = torch.cuda.current_stream(dev)
            side = _side_streams.get(dev.index) if overlap else main
            if side is None:
                side = _side_streams[dev.index] = torch.cuda.Stream(device=dev)
            side.wait_stream(main)      # (the previous consumers of the plan's buffers are ordered on `main`)
            # numpy order: the finishing launches of pass k on stream k mod 4 of their own (a tensor's float32 chunk sums
            # are added by ONE thread in numpy's order -- 0.3 ms for the 64 128 chunks of an lm_head -- and the passes'
            # chains are independent of each other)
            fins = [main] * 4
            if self.numpy_order and overlap:
                for j in range(4):
                    f = _side_streams.get((dev.index, "fin", j))
                    if f is None:
                        f = _side_streams[(dev.index, "fin", j)] = torch.cuda.Stream(device=dev)
                    f.wait_stream(main)
                    fins[j] = f
            if self.numpy_order and (self.fin_ws is None or self.fin_ws[0].numel() < self.fin_ws_bytes):
                self.fin_ws = [torch.empty(self.fin_ws_bytes, dtype=torch.uint8, device=dev) for _ in range(4)]
            ws, _ = workspace(dev, self.ws_bytes)
            # the sampling launches of pass k (fed lists: samples -> brackets) run on a stream of their own beside the
            # launches of pass k - 1 (behind the selections on the second stream they would hold up pass k)
            pre = main
            if overlap and any(la[0] == "feedsweep" for ps in self.plan for la in ps["launches"]):
                pre = _side_streams.get((dev.index, "pre"))
                if pre is None:
                    pre = _side_streams[(dev.index, "pre")] = torch.cuda.Stream(device=dev)
                pre.wait_stream(main)

            def sample(k):
                ev = None
                for la in self.plan[k]["launches"]:
                    if la[0] == "feedsweep":
                        _lib.check(L.mq_sweep_feed_sample(la[6].data_ptr(), la[7], la[3], la[8], la[5],
                                                          ctypes.byref(self.plan[k]["cfg"]), la[9].data_ptr(),
                                                          pre.cuda_stream))
                        ev = ev or torch.cuda.Event()
                if ev is not None:
                    ev.record(pre)
                return ev
            sampled = sample(0)
            for k, ps in enumerate(self.plan):
                cfg = ps["cfg"]
                if sampled is not None and pre is not main:
                    main.wait_event(sampled)
                chunk_done, np_done = [], []
                for la in ps["launches"]:
                    if la[0] == "multisweep":
                        # (each chunk needs the whole workspace: the partial sums of chunk i are added up before chunk
                        #  i + 1 starts, on this stream)
                        job = la[7]
                        # (the totals launch of a chunk on the second stream, beside the next statistics launch -- but
                        #  behind the step's last launch there is nothing to run beside)
                        last = la is ps["launches"][-1] and not any(ps2["launches"] for ps2 in self.plan[k + 1:])
                        job["split"] = side is not main and not last and os.environ.get("MQ_SWEEP_SPLIT_FINALIZE", "1") != "0"
                        run = L.mq_sweep_multi_launch if job["split"] else L.mq_sweep_multi_run
                        _lib.check(run(la[1].data_ptr(), la[2], la[3], la[4], la[5], ctypes.byref(cfg), la[6],
                                       ws.data_ptr() + job["ws_off"], job["ws_len"], main.cuda_stream))
                        done = torch.cuda.Event()
                        done.record(main)
                        chunk_done.append((la, done))
                    elif la[0] == "npsweep":
                        _lib.check(L.mq_sweep_numpy_run(la[1].data_ptr(), la[2], la[3], la[4], la[5], ctypes.byref(cfg),
                                                        ws.data_ptr(), ws.numel(), main.cuda_stream))
                    elif la[0] == "npgroups":
                        t32, t64, t128 = (None if t is None else t.data_ptr() for t in la[1])
                        _lib.check(L.mq_sweep_numpy_run_groups(t32, t64, t128, la[2], la[3], la[4], ctypes.byref(cfg),
                                                               ws.data_ptr(), ws.numel(), main.cuda_stream))
                        if len(la) > 5:   # a chunk: its passes' selections and finishing launches follow it at once
                            done = torch.cuda.Event()
                            done.record(main)
                            np_done.append((la, done))
                    elif la[0] == "npsweep_fused":
                        pass      # (written by the family's npgroups launch)
                    elif la[0] == "feedsweep":
                        _lib.check(L.mq_sweep_feed_run(la[1].data_ptr(), la[2], la[3], la[4], la[5], ctypes.byref(cfg),
                                                       ws.data_ptr(), ws.numel(), main.cuda_stream))
                    elif la[0] == "sweep":
                        _, tab, n, seg_len, blocks, dt, flag = la
                        _lib.check(L.mq_sweep_run(tab.data_ptr(), n, seg_len, blocks, dt, ctypes.byref(cfg), flag,
                                                  ws.data_ptr(), ws.numel(), main.cuda_stream))
                    else:
                        _, items, B, dt, K, C = la
                        _lib.check(L.mq_quantize_batch(items, B, dt, K, C, ctypes.byref(cfg), ws.data_ptr(), ws.numel(),
                                                       main.cuda_stream))
                ready = torch.cuda.Event()
                ready.record(main)
                sampled = sample(k + 1) if k + 1 < len(self.plan) else None
                # Selections that run beside a later launch take the small-workgroup form, which finds room on a busy
                # chip; those behind the step's last launch have the chip to themselves and take the plain one.
                later = any(ps2["launches"] for ps2 in self.plan[k + 1:])
                small = side is not main and self.select_beside

                def select(beside):
                    return L.mq_percentile_multi_beside if (small and beside) else L.mq_percentile_multi
                with torch.cuda.stream(side):
                    sws, _ = workspace(dev, self.side_ws_bytes)
                    # the lists of a family's chunks, every pass of the family: chunk i's beside the launch of chunk
                    # i + 1 (enqueued after ALL of the launches above: the host's calls for them would otherwise hold
                    # up the next launch)
                    for j, (la, done) in enumerate(chunk_done):
                        side.wait_event(done)
                        job = la[7]
                        if job["split"]:   # the chunk's totals from its workgroups' partial sums
                            _lib.check(L.mq_sweep_multi_finalize(la[1].data_ptr(), la[2], la[4], ws.data_ptr() + job["ws_off"],
                                                                 job["ws_len"], side.cuda_stream))
                        # (beside the NEXT launch; a short one is over before the slower small workgroups are)
                        nxt = chunk_done[j + 1][0] if j + 1 < len(chunk_done) else None
                        beside = nxt is not None and nxt[7]["bytes"] >= self.beside_bytes
                        _lib.check(select(beside)(job["ptrs"], job["ns"], job["n"], _lib.F32_AS_F64, self.pct,
                                                  job["out"].data_ptr(), None, sws.data_ptr(), sws.numel(),
                                                  side.cuda_stream))
                    for j, (la, done) in enumerate(np_done):
                        side.wait_event(done)
                        # (beside the next chunk's launch, or -- the last chunk -- beside the launches of later passes)
                        nxt = np_done[j + 1][0] if j + 1 < len(np_done) else None
                        beside = (nxt[6] >= self.beside_bytes) if nxt is not None else (later or len(ps["launches"]) > len(np_done))
                        for k2, sel2, _ in la[5]:
                            _lib.check(select(beside)(sel2["ptrs"], sel2["ns"], sel2["n"], _lib.F32_AS_F64, self.pct,
                                                      sel2["out"].data_ptr(), None, sws.data_ptr(), sws.numel(),
                                                      side.cuda_stream))
                    side.wait_event(ready)
                    sel = ps["sel"]
                    if sel["n"]:
                        out = ps["p99"] if sel["idx"] is None else sel["tmp"]
                        _lib.check(select(later)(sel["ptrs"], sel["ns"], sel["n"], _lib.F32_AS_F64, self.pct,
                                                 out.data_ptr(), None, sws.data_ptr(), sws.numel(), side.cuda_stream))
                        if sel["idx"] is not None:
                            ps["p99"].index_copy_(0, sel["idx"], out)
                    for la in ps["launches"]:   # the fed lists: the wanted ranks among the kept entries
                        if la[0] == "feedsweep":
                            _lib.check(L.mq_sweep_feed_finish(la[6].data_ptr(), la[7], la[3], la[8], la[5], ctypes.byref(cfg),
                                                              la[9].data_ptr(), la[12].data_ptr(), side.cuda_stream))
                            ps["p99"].index_copy_(0, la[11], la[12][la[10]])
                    # the rows of the record table (mean = sse / numel, NaN lists, row order): every pass's in ONE launch
                    # behind the last pass -- or pass by pass in numpy_order mode, whose means follow on a third stream
                    if self.numpy_order:
                        _lib.check(L.mq_sweep_records(ps["totals"].data_ptr(), ps["p99"].data_ptr(), self.numel.data_ptr(),
                                                      self.perm.data_ptr(), nl, self.records[k].data_ptr(),
                                                      side.cuda_stream))
                    elif k + 1 == len(self.plan):
                        _lib.check(L.mq_sweep_table(self.totals_all.data_ptr(), self.p99_store.data_ptr(),
                                                    self.p99_index.data_ptr(), self.numel_all.data_ptr(),
                                                    self.table_src.data_ptr(), self.table_src.numel(),
                                                    self.records.data_ptr(), side.cuda_stream))
                    rec_done = None
                    if self.numpy_order:
                        rec_done = torch.cuda.Event()
                        rec_done.record(side)
                if self.numpy_order:
                    # mse_rep = float(vec.mean()) and mse_tensor = float(np.mean((a32 - b32) ** 2)) in numpy's bits, on a
                    # THIRD stream: the selections' histogram kernels want a whole compute unit per workgroup and wait
                    # long for one beside the next pass, while these small workgroups slip in at once -- behind the
                    # selections they would pile up at the end of the step
                    for la, ev in np_done:   # the chunks' leaf sums and lists -> mse_tensor, mse_rep of their passes
                        for k2, _, (tab, n, seg_len, leaf, tl, lc, rep, mt) in la[5]:
                            f2, fws2 = fins[k2 % 4], self.fin_ws[k2 % 4]
                            f2.wait_event(ev)
                            _lib.check(L.mq_sweep_numpy_finish(tab.data_ptr(), n, seg_len, leaf.data_ptr(), tl, lc,
                                                               mt.data_ptr(), rep.data_ptr(), fws2.data_ptr(),
                                                               fws2.numel(), f2.cuda_stream))
                    fin, fws = fins[k % 4], self.fin_ws[k % 4]
                    with torch.cuda.stream(fin):
                        fin.wait_event(ready)
                        done = []
                        # from the launch's leaf sums and lists (the layers of a numpy-order launch) ...
                        for la in ps["launches"]:
                            if la[0] not in ("npsweep", "npsweep_fused"):
                                continue
                            _, tab, n, seg_len, _, _, leaf, tl, lc, widx, rep, mt = la
                            _lib.check(L.mq_sweep_numpy_finish(tab.data_ptr(), n, seg_len, leaf.data_ptr(), tl, lc,
                                                               mt.data_ptr(), rep.data_ptr(), fws.data_ptr(),
                                                               fws.numel(), fin.cuda_stream))
                        # ... or layer by layer (shapes the numpy-order launch does not take)
                        slow = ps["slow_pos"]
                        if slow:
                            sidx = torch.tensor([self.order[pos] for pos in slow], dtype=torch.int64, device=dev)
                            rep = numpy_list_means([ps["lists"][self.order[pos]] for pos in slow], fin)
                            m32 = torch.empty(len(slow), dtype=torch.float32, device=dev)
                            for j, pos in enumerate(slow):
                                sc, zp = ps["params"][pos]
                                tensor_mse_numpy(self.weights[self.order[pos]], cfg, sc, zp, m32[j], fin)
                            done.append((sidx, rep, m32))
                        fin.wait_event(rec_done)     # (the pass's record rows are written: now their mse_rep column)
                        for widx, rep, mt in done:
                            self.records[k][:, 3].index_copy_(0, widx, rep)
                            self.mse_tensor[k].index_copy_(0, widx, mt.double())
            self._pending = (main, side, fins)

    @torch.no_grad()
    def finish(self) -> torch.Tensor:
        """Joins the second stream into the one launch() ran on and returns the records (ordered on that stream)."""
        main, side, fins = self._pending
        self._pending = None
        if side is not main:
            main.wait_stream(side)
        for fin in fins:
            if fin is not main:
                main.wait_stream(fin)
        if self.numpy_order and self.np_dst is not None:
            # mse_rep and mse_tensor of every finishing call of the run to their rows: one scatter each
            m = self.np_dst.numel()
            self.records.view(-1, len(RECORD_FIELDS))[:, 3].index_copy_(0, self.np_dst, self.np_rep[:m])
            self.mse_tensor.view(-1).index_copy_(0, self.np_dst, self.np_mt[:m].double())
        return self.records


@torch.no_grad()
def numpy_order_available(weights, passes) -> bool:
    """Can every pass over these weights be summed in numpy's order (SweepPlan(numpy_order=True))?  float16 / bfloat16
    weights, per-channel or per-group passes (not per-tensor), segments of a multiple of 8 and at most 32768
    elements."""
    for w in weights:
        if w.dtype not in (torch.float16, torch.bfloat16):
            return False
    for _, _, gs in passes:
        if gs == -1:
            return False
        for w in weights:
            L = gs if (gs is not None and gs > 0) else w.shape[1]
            if L % 8 or L > 32768 or w.shape[1] % L:
                return False
    return True


@torch.no_grad()
def model_sweep(weights, passes, pct: float = 99.0, vecs_of: Optional[int] = None, numpy_order: Optional[bool] = None,
                return_mse_tensor: bool = False, fuse_groupings: bool = True):
    """model_records(weights, datatype, bits, group_size, abs_p99=False) for every (datatype, bits, group_size) of
    `passes`, one shot: float64 [len(passes), len(weights), len(RECORD_FIELDS)] (and, with vecs_of = k, the
    per-vector MSE lists of pass k; with return_mse_tensor the tensor means [len(passes), len(weights)]).  Builds a
    SweepPlan and runs it once; keep the plan to sweep the same resident weights repeatedly.

    numpy_order (default: on wherever the kernels offer it, numpy_order_available): mse_rep, mse_p99 and the tensor
    means have the BITS of the reference's report rows (quant_error_multi_weight.py:55-58, quant_error_single_weight
    .py:74-79); False: float64 sums, equal to those to the 1e-6 bar on small tensors and a few 1e-6 away on whole
    LLM weights, where numpy's float32 sums drift."""
    weights, passes = list(weights), list(passes)
    if numpy_order is None:
        numpy_order = numpy_order_available(weights, passes)
    plan = SweepPlan(weights, passes, pct, numpy_order=numpy_order, fuse_groupings=fuse_groupings)
    out = plan.run()
    res = [out]
    if vecs_of is not None:
        res.append(plan.lists(vecs_of))
    if return_mse_tensor:
        res.append(plan.mse_tensor if numpy_order else out[:, :, 1] / out[:, :, 0])
    return tuple(res) if len(res) > 1 else out


def compare_onnx(baseline_model: str, quant_model: str, form: str = "multi", device=None,
                 numpy_order: bool = True) -> List[Dict]:
    """main() of MSE&P99/quant_error_multi_weight.py:119-164 on two ONNX files, without the `onnx` package and
    with the arithmetic on the device: for every MatMul / Gemm weight of the baseline model that the quantized
    model also holds, float32 cast, tensor MSE, per-vector MSE list (form="multi": columns / column blocks as
    that script; form="single": rows / (row, group) as quant_error_single_weight.py), its mean and P99.
    numpy_order (the default: this is the report path): every mean in numpy's summation order -- the script's
    numbers bit for bit (compute_errors); False: float64 sums in one pass, equal to those to 1e-6.
    Returns the CSV rows (write them with write_csv)."""
    from .error_metrics import compute_errors
    from .onnx_io import OnnxWeights
    grouping, dtype, bits, gs = parse_quant_config(quant_model)
    rows = []
    with OnnxWeights(baseline_model) as base, OnnxWeights(quant_model) as quant:
        wanted = set(base.compute_weights())
        have = set(quant.names())
        for name in base.names():
            if name not in wanted or name not in have:
                continue
            a = base.tensor(name, device).float()   # proto_to_array: everything becomes float32
            b = quant.tensor(name, device).float()
            stats = compute_errors(a, b, grouping, gs, form=form, numpy_order=numpy_order)
            rows.append({"initializer": name, "datatype": dtype, "granularity": grouping, "bits": bits or "",
                         "mse_tensor": stats["mse_tensor"], "mse_rep": stats.get("mse_rep", ""),
                         "mse_p99": stats.get("mse_p99", "")})
    return rows


def compare_onnx_single(baseline_model: str, quant_models: Iterable[str], initializer: str, bins: int = 50,
                        device=None, numpy_order: bool = True):
    """main() of MSE&P99/quant_error_single_weight.py:85-140 for one initializer of the baseline model and several
    quantized models: one CSV row per quantized model (tensor MSE; mean and P99 of the per-row / per-(row, group)
    MSE list) and, in place of the script's plt.hist(vec, bins=50) calls (:118-124), each list's np.histogram
    counts and edges computed on the device.  numpy_order: as for compare_onnx.
    -> (rows, {label: (counts int64 tensor, edges ndarray)})"""
    from .error_metrics import compute_errors, histogram
    from .onnx_io import OnnxWeights
    rows, hists = [], {}
    with OnnxWeights(baseline_model) as base:
        if initializer not in set(base.names()):
            raise ValueError(f"{initializer} not in baseline.")   # :90-91
        a = base.tensor(initializer, device).float()
    for qpath in quant_models:
        grouping, dtype, bits, gs = parse_quant_config(qpath)
        with OnnxWeights(qpath) as quant:
            if initializer not in set(quant.names()):
                continue                                           # the script warns and skips (:101-103)
            b = quant.tensor(initializer, device).float()
        stats = compute_errors(a, b, grouping, gs, form="single", keep_per_vec=True, numpy_order=numpy_order)
        rows.append({"datatype": dtype, "granularity": grouping, "bits": bits or "",
                     "mse_tensor": stats["mse_tensor"], "mse_rep": stats.get("mse_rep", ""),
                     "mse_p99": stats.get("mse_p99", "")})
        if "per_vec" in stats:
            hists[f"{dtype}-{grouping}"] = histogram(stats["per_vec"], bins=bins)
    return rows, hists


def rows_from_records(names: Iterable[str], records: torch.Tensor, datatype: str, bits: int, group_size,
                      mse_tensor: Optional[torch.Tensor] = None) -> List[Dict]:
    """Device records -> the CSV rows of quant_error_multi_weight.py:142-151 (one host copy).  mse_tensor: the
    tensor means when they were computed apart (numpy_order: model_records(return_mse_tensor=True),
    SweepPlan.mse_tensor), else sse / numel."""
    host = records.cpu().tolist()
    mt = mse_tensor.cpu().tolist() if mse_tensor is not None else None
    rows = []
    for k, (name, r) in enumerate(zip(names, host)):
        rows.append({"initializer": name, "datatype": datatype, "granularity": granularity(group_size),
                     "bits": bits or "", "mse_tensor": mt[k] if mt is not None else (r[1] / r[0] if r[0] else float("nan")),
                     "mse_rep": r[3], "mse_p99": r[4], "abs_sum": r[2], "abs_p99": r[5]})
    return rows


def write_csv(rows: List[Dict], path: str, columns=MULTI_COLUMNS) -> None:
    """csv.DictWriter with the reference's column order; floats are written with repr() as csv does."""
    with open(path, "w", newline="") as f:
        wr = csv.DictWriter(f, fieldnames=columns, extrasaction="ignore")
        wr.writeheader()
        wr.writerows(rows)
