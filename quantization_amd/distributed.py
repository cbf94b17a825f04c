"""Whole-model runs over the GPUs of one node: one process per GPU (torch.distributed, backend "nccl" =
RCCL over xGMI).  Linear layers are independent, so they are sharded by size and no weight ever crosses
GPUs; RCCL carries only (1) the per-layer statistics records -- one small all-reduce per sweep -- and
(2) the uint32 histograms of a percentile taken jointly over tensors on several GPUs.  The reference has
no counterpart (single CPU process)."""
import ctypes
from typing import Callable, List, Optional, Sequence

import torch

from . import _lib


def lpt_partition(sizes: Sequence[int], world: int) -> List[List[int]]:
    """Greedy longest-processing-time assignment of items to `world` bins; returns per-rank index lists
    (each in ascending order).  Deterministic: ties go to the lower rank."""
    order = sorted(range(len(sizes)), key=lambda i: (-sizes[i], i))
    load = [0] * world
    parts: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        parts[r].append(i)
        load[r] += sizes[i]
    return [sorted(p) for p in parts]


def all_reduce_(t: torch.Tensor, op=None, group=None) -> torch.Tensor:
    """dist.all_reduce of a device tensor, in place.  RCCL ("nccl"): enqueued on the current stream, as it is.
    Any other backend -- gloo, the CPU stand-in for RCCL in rehearsals and tests -- gets a HOST copy made here:
    gloo's own staging of device tensors (pinned buffers, a worker thread) has been seen to wait forever on ROCm,
    with queued work on the device and, intermittently, with two ranks sharing one GPU even after the device was
    drained.  The device is drained first (every stream), the result is copied back."""
    import torch.distributed as dist
    op = dist.ReduceOp.SUM if op is None else op
    if dist.get_backend(group) == "nccl" or not t.is_cuda:
        dist.all_reduce(t, op=op, group=group)
        return t
    torch.cuda.synchronize()
    h = t.cpu()
    dist.all_reduce(h, op=op, group=group)
    t.copy_(h)
    return t


def gather_records(local_idx: Sequence[int], local_records: Optional[torch.Tensor], n_total: int, width: int,
                   device, group=None) -> torch.Tensor:
    """Every rank contributes the rows it owns; all ranks get the full [n_total, width] float64 table.
    One all-reduce(SUM) of a zero-filled table (tens of KiB: latency-bound)."""
    import torch.distributed as dist
    table = torch.zeros(n_total, width, dtype=torch.float64, device=device)
    if len(local_idx):
        idx = local_idx if isinstance(local_idx, torch.Tensor) else torch.as_tensor(list(local_idx), device=device)
        table[idx.to(device)] = local_records.to(torch.float64)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        all_reduce_(table, group=group)
    return table


def reduce_table(table: torch.Tensor, group=None) -> torch.Tensor:
    """All-reduce (SUM), in place, of a report table in which every rank has written its own layers' rows and zeros
    everywhere else (report.SweepPlan.set_table): all ranks get the full table.  gather_records without its fill and
    scatter launches -- the table is complete, zeros included, when the sweep's last launch has run.  One rank, or no
    process group: the table as it is."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        all_reduce_(table, group=group)
    return table


def analyze_sharded(layers: Sequence, sizes: Sequence[int], analyze_fn: Callable, width: int, device, group=None):
    """layers[i] is whatever analyze_fn needs to produce layer i's record (a float64[width] tensor on
    `device`).  Returns (full table on every rank, indices this rank processed)."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    rank = dist.get_rank(group) if world > 1 else 0
    mine = lpt_partition(sizes, world)[rank]
    recs = [analyze_fn(layers[i]) for i in mine]
    local = torch.stack(recs) if recs else None
    return gather_records(mine, local, len(layers), width, device, group), mine


# --------------------------------------------------------------------------------------------------
# percentile over several tensors, possibly on several GPUs
# --------------------------------------------------------------------------------------------------
def _bind_select():
    L = _lib.lib()
    if getattr(L, "_sel_bound", False):
        return L
    vp, i64, i32, sz, dbl = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_size_t, ctypes.c_double
    L.mq_percentile_workspace_bytes.restype = sz
    L.mq_select_layout.restype = None
    L.mq_select_layout.argtypes = [i32, ctypes.POINTER(sz), ctypes.POINTER(sz)]
    L.mq_percentile_index.restype = ctypes.c_int
    L.mq_percentile_index.argtypes = [i32, i64, dbl, ctypes.POINTER(i64), ctypes.POINTER(i64), ctypes.POINTER(dbl)]
    L.mq_select_begin.restype = ctypes.c_int
    L.mq_select_begin.argtypes = [vp, sz, vp]
    L.mq_select_accumulate.restype = ctypes.c_int
    L.mq_select_accumulate.argtypes = [vp, i32, i64, i32, vp, sz, vp]
    L.mq_select_accumulate_multi.restype = ctypes.c_int
    L.mq_select_accumulate_multi.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(i64), i32, i32, i32, vp, sz, vp]
    L.mq_select_scan.restype = ctypes.c_int
    L.mq_select_scan.argtypes = [i32, i64, i64, i32, i32, dbl, vp, vp, vp, sz, vp]
    L._sel_bound = True
    return L


_merged_hist_checked = False


@torch.no_grad()
def joint_percentile(tensors: Sequence[torch.Tensor], pct: float = 99.0, group=None, dtype=None,
                     total: Optional[int] = None, wide: bool = False) -> torch.Tensor:
    """np.percentile(np.concatenate(all tensors on all ranks), pct), exact, without moving the data:
    every rank histograms its own tensors, the uint32 histograms are summed with one all-reduce per radix
    level (128 KiB, then 512 KiB per further level), and every rank finishes the same selection.
    All tensors: non-negative, same dtype (float32 or float64), on this rank's device.  A rank may pass
    an empty list.  dtype / total: the common dtype and the total element count over all ranks when the caller
    knows them (a sweep repeats the same shapes): then no metadata is exchanged and the host does not wait for
    the device; otherwise they are agreed on with two small collectives.  wide (float32 tensors): the result
    is np.percentile of the float64 array holding the same values (float32 keys: two histogram levels; float64
    ranks, interpolation and result) -- per-vector MSE lists are float32 means kept in a float64 array by the
    analysis scripts (quant_error_single_weight.py:74-79).  No NaN check: a NaN is ranked as the largest value (numpy
    would return NaN); callers that may hold one test their data themselves."""
    import torch.distributed as dist
    L = _bind_select()
    dist_on = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    dev = torch.device("cuda", torch.cuda.current_device())
    known_dtype = dtype is not None
    dtype = dtype if known_dtype else (tensors[0].dtype if tensors else None)
    if dist_on and not known_dtype:
        dts = [None] * dist.get_world_size(group)
        dist.all_gather_object(dts, str(dtype) if dtype is not None else None, group=group)
        known = [d for d in dts if d is not None]
        if not known or any(d != known[0] for d in known):
            raise ValueError("joint_percentile: tensors must share one dtype and at least one rank must hold data")
        dtype = {"torch.float32": torch.float32, "torch.float64": torch.float64}[known[0]]
    if dtype not in (torch.float32, torch.float64):
        raise ValueError("float32 or float64 tensors expected")
    if wide and dtype != torch.float32:
        raise ValueError("wide=True takes float32 tensors")
    code = _lib.F32_AS_F64 if wide else (_lib.F32 if dtype == torch.float32 else _lib.F64)
    xs = [t.contiguous().reshape(-1) for t in tensors if t.numel()]
    xs = [x.clone() if x.data_ptr() % 16 else x for x in xs]   # (sliced views: the kernels read 16-byte vectors)
    if total is None:
        n = torch.tensor([sum(x.numel() for x in xs)], dtype=torch.int64, device=dev)
        if dist_on:
            all_reduce_(n, group=group)
        n = int(n.item())
    else:
        n = int(total)
    k0, k1, gamma = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_double()
    _lib.check(L.mq_percentile_index(code, n, float(pct), ctypes.byref(k0), ctypes.byref(k1), ctypes.byref(gamma)))
    nbytes = L.mq_percentile_workspace_bytes()
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    wp, wn = ws.data_ptr(), ws.numel()
    out = torch.empty((), dtype=torch.float64 if wide else dtype, device=dev)
    _lib.check(L.mq_select_begin(wp, wn, st))
    levels = 1 if dtype == torch.float32 else 3
    # lists that follow each other in memory without a gap (the per-channel lists of a SweepPlan pass live in one
    # buffer) are histogrammed as one array: a launch takes 64 arrays, a model has hundreds of lists
    runs = []
    for x in xs:
        if runs and runs[-1][0] + runs[-1][1] * x.element_size() == x.data_ptr():
            runs[-1][1] += x.numel()
        else:
            runs.append([x.data_ptr(), x.numel()])
    ptrs = (ctypes.c_void_p * len(runs))(*[r[0] for r in runs])
    lens = (ctypes.c_int64 * len(runs))(*[r[1] for r in runs])
    for level in range(levels + 1):
        _lib.check(L.mq_select_accumulate_multi(ptrs, lens, len(runs), code, level, wp, wn, st))
        if dist_on:
            off, cnt = ctypes.c_size_t(), ctypes.c_size_t()
            L.mq_select_layout(level, ctypes.byref(off), ctypes.byref(cnt))
            # the uint32 counters travel as int32 (the collectives have no unsigned 32-bit sum): exact while every
            # summed bin stays below 2^31, i.e. for fewer than 2^31 values in all -- refused above that, and the
            # first merged histogram of a process is checked once on the host
            if n >= 1 << 31:
                raise OverflowError("joint_percentile: 2^31 or more values in one selection across ranks")
            hist = ws[off.value: off.value + 4 * cnt.value].view(torch.int32)
            all_reduce_(hist, group=group)
            global _merged_hist_checked
            if not _merged_hist_checked:
                _merged_hist_checked = True
                if int(hist.min().item()) < 0:
                    raise OverflowError("joint_percentile: a merged histogram bin passed 2^31")
        _lib.check(L.mq_select_scan(level, k0.value, k1.value, int(level == levels), code, gamma.value,
                                    out.data_ptr(), None, wp, wn, st))
    return out
