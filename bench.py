#!/usr/bin/env python3
"""Benchmark of the quantize -> dequantize -> error hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
                    [--workload llama8b|llama70b|tinyllama|c2|c3|llama70b_shard] [--dist normal|student_t]

Headline workload "llama8b" (BASELINE.json configs[3]): all 225 linear weights of Llama-3-8B (random-init bf16,
15.0 GB), symmetric int8 error statistics per-channel and per-group g = 32 / 64 / 128 -- four passes over the
model per step -- with the layers SHARDED over the N ranks (greedy longest-processing-time on the element
count; no weight crosses GPUs).  Per rank the four groupings of a layer come from ONE read of it (report.SweepPlan:
mq_sweep_multi_launch over chunks of layers, nine launches per step at N = 1; the one-launch-per-pass form, mq_sweep_run,
is timed beside it); per step RCCL carries one all-reduce of the per-layer record table
(distributed.reduce_table) and the histogram all-reduces of one model-wide P99 taken jointly over every
rank's per-channel MSE lists (distributed.joint_percentile).  "llama70b" (configs[4]) is the same over the 561
weights of Llama-3-70B (139 GB) with int8 sym / asym, fp8_e4m3, fp8_e3m4 x {per-channel, g = 32 / 64 / 128};
"tinyllama" (configs[0]) per-channel int8 on fp16 weights.  The total work is fixed as N grows ("strong").

--gpus N without a launcher starts N ranks itself (fresh child processes through torch.distributed.run, before
this process touches a GPU); under torch.distributed.run it must equal WORLD_SIZE.  MQ_BENCH_BACKEND=gloo
rehearses the multi-rank path on a box with fewer GPUs than ranks.

At N = 1 the same JSON line carries the 1-GPU leg of every other BASELINE configuration under "extra", each timed the
same way with its own roofline object: "c1_tinyllama" (configs[0]), "c2" (configs[1]: 16 x 4096 x 4096 fp16,
asymmetric int8 g = 32 / 64 / 128 with dequantized weights, codes, scales, zero-points, the float32 |error| dump and
its exact P99 selected inside the pass), "c3" (configs[2]: three dequantized formats + the compare.py masks),
"llama70b_shard" (one GPU's eighth of configs[4]) and "c5_llama70b" (configs[4] whole on one GPU: 139 GB resident);
and the CPU baseline.  Every step ends with the statistics copied to host memory (inside the timed region).

One JSON line is printed by rank 0.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak
MODEL_OF = {"llama8b": "llama-3-8b", "llama70b": "llama-3-70b", "tinyllama": "tinyllama-1.1b"}
GROUPS = (None, 32, 64, 128)
# Order of the passes in a model sweep: the selections of a pass run under the launches of the next one, only the
# last pass's are exposed -- so the longest lists (g = 32) go first and the per-channel pass, whose lists are the
# rows of a layer, goes last.
SWEEP_ORDER = tuple((None if g == "pc" else int(g)) for g in os.environ.get("MQ_BENCH_ORDER", "32,64,128,pc").split(","))
REFERENCE_BUILD_CONTAINER = {
    # the reference itself (imported quant_weight._int_quant_meta on torch CPU + the scripts' numpy arithmetic) on
    # the c2 unit of work, timed in the build container by scratch/ref_time.py; /root/reference does not exist
    # on the GPU box, so this number cannot be re-measured there
    "value": 0.052, "unit": "GB/s", "cores": 8, "where": "build container (8 cores, torch CPU 8 threads)",
    "sample": "one 4096x4096 fp16 matrix x g=32/64/128, asymmetric int8 + float32 error, MSE, |err| sum and P99"}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="llama8b",
                    choices=["llama8b", "llama70b", "tinyllama", "c2", "c3", "llama70b_shard"])
    ap.add_argument("--dist", default="normal", choices=["normal", "student_t"])
    ap.add_argument("--numpy-order", action="store_true",
                    help="model sweeps: every mean in numpy's summation order (mse_tensor / mse_rep / mse_p99 in the "
                         "reference's bits); about 1.6 x the default step")
    ap.add_argument("--dump-table", metavar="PATH", default=None,
                    help="model sweeps: rank 0 saves the last step's record table and model-wide P99 (numpy .npz)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="N = 1: skip the c2 and llama70b_shard results")
    ap.add_argument("--watchdog", type=int, default=None, metavar="SECONDS",
                    help="every rank: after SECONDS print all Python stacks to stderr and exit non-zero (a hang names "
                         "itself); default 300 with more than one rank, off with one; 0 = off")
    return ap.parse_args()


def spawn_ranks(args):
    """--gpus N from a plain `python bench.py`: start N ranks as fresh children; this process never touches a GPU."""
    import torch
    if os.environ.get("MQ_BENCH_BACKEND", "nccl") == "nccl" and torch.cuda.device_count() < args.gpus:
        print(f"bench.py: --gpus {args.gpus} but only {torch.cuda.device_count()} GPU(s) visible", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def store_host(dst, src):
    """Small device tensors -> pinned host tensors with one kernel on the current stream (mq_store_host)."""
    import torch
    from quantization_amd import _lib
    L = _lib.lib()
    n = len(dst)
    src = [s.contiguous() for s in src]
    vp = ctypes.c_void_p
    _lib.check(L.mq_store_host((vp * n)(*[d.data_ptr() for d in dst]), (vp * n)(*[s.data_ptr() for s in src]),
                               (ctypes.c_size_t * n)(*[s.numel() * s.element_size() for s in src]), n,
                               vp(torch.cuda.current_stream().cuda_stream)))


def algorithmic_bytes_per_elem(in_bytes, lut, asym, gs, outputs):
    """HBM bytes one fused launch must move per weight element: input once, plus each requested output."""
    import torch
    b = float(in_bytes)
    if outputs.get("deq"):
        b += 2
    if outputs.get("codes"):
        b += 2 if lut else 1
    if outputs.get("abs_err") == torch.float32:
        b += 4
    elif outputs.get("abs_err") == torch.float16:
        b += 2
    if gs and gs > 0:
        per_seg = (2 if outputs.get("scale") else 0) * (2 if asym else 1)
        per_seg += 8 if outputs.get("seg_sse") else (4 if outputs.get("seg_mse") else 0)
        b += per_seg / gs
    return b


class ModelSweep:
    """The sharded model sweep (configs [0], [3], [4]): this rank's layers of the model, every pass of the sweep."""

    def __init__(self, name, device, rank, world, dist_name, numpy_order=False, fuse=True):
        import torch
        from quantization_amd import shapes
        from quantization_amd.distributed import lpt_partition
        self.name, self.device, self.rank, self.world = name, device, rank, world
        model = MODEL_OF[name]
        self.layers = shapes.linear_shapes(model)
        self.dtype = torch.float16 if name == "tinyllama" else torch.bfloat16
        if name == "llama70b":
            fmts = [("int8", 8), ("int8_asym", 8), ("fp8_e4m3", 8), ("fp8_e3m4", 8)]
            self.passes = [(f, b, gs) for f, b in fmts for gs in SWEEP_ORDER]
        elif name == "llama8b":
            # (numpy order: the per-channel pass first -- its finishing chain, one thread adding an lm_head's 64 128
            #  chunk sums in order, then runs beside the group passes instead of behind the last launch)
            self.passes = [("int8", 8, gs) for gs in (([None] + [g for g in SWEEP_ORDER if g]) if numpy_order else SWEEP_ORDER)]
        else:
            self.passes = [("int8", 8, None)]
        sizes = [r * c for _, (r, c) in self.layers]
        self.mine = lpt_partition(sizes, world)[rank]
        self.weights = [shapes.synthetic_weight(i, self.layers[i][1], self.dtype, device, dist=dist_name) for i in self.mine]
        esz = 2
        self.local_elems = sum(sizes[i] for i in self.mine)
        self.total_weight_bytes = sum(sizes) * esz * len(self.passes)          # whole job, every pass counts
        self.local_weight_bytes = self.local_elems * esz * len(self.passes)
        # every pass of the sweep over this rank's resident layers: launch descriptors and buffers are prepared once
        from quantization_amd.report import SweepPlan
        self.plan = (SweepPlan(self.weights, self.passes, numpy_order=bool(numpy_order), fuse_groupings=fuse)
                     if self.weights else None)
        # this rank's rows go straight into the whole model's table (every other row zero): the all-reduce needs no
        # fill and scatter in front of it
        self.table_mode = self.plan is not None and not numpy_order
        if self.table_mode:
            self.plan.set_table(len(self.layers), self.mine)
        # algorithmic bytes: every weight once per pass, plus the float32 per-vector list where the pass writes it (the
        # list of a large tensor in a per-group pass is not written: its percentile is selected inside the pass)
        # (the passes of a format that differ only in the grouping read a layer ONCE: report.SweepPlan, mq_sweep_multi_run)
        self.local_algo_bytes = (self.plan.read_bytes + self.plan.list_bytes) if self.plan is not None else 0
        self.fused = self.plan is not None and bool(self.plan.families)
        self.local_read_bytes = self.plan.read_bytes if self.plan is not None else 0
        self.launches_hint = sum(-(-sum(1 for i in self.mine if self.layers[i][1] == sh) // 32)
                                 for sh in {self.layers[i][1] for i in self.mine}) * len(self.passes)
        self.n_layers = len(self.layers)
        self.host_table = self.host_p99 = None
        self.numpy_order = bool(numpy_order)
        self.overlap = True   # selections of pass k on a second stream under the launches of pass k + 1
        # the per-channel lists of the first format feed the model-wide percentile
        self.joint_pass = next(k for k, (_, _, gs) in enumerate(self.passes) if gs is None)
        self.total_vectors = sum(r for _, (r, c) in self.layers)
        self.index = torch.as_tensor([p * self.n_layers + i for p in range(len(self.passes)) for i in self.mine],
                                     dtype=torch.int64, device=device)
        fm = ", ".join(sorted({f for f, _, _ in self.passes}))
        self.desc = (f"{model}: {self.n_layers} linear weights ({sum(sizes) * esz / 1e9:.1f} GB {str(self.dtype).split('.')[-1]}, "
                     f"{dist_name} scale 0.02), {fm} x per-channel / g=32 / 64 / 128 error statistics"
                     if len(self.passes) > 1 else
                     f"{model}: {self.n_layers} linear weights ({sum(sizes) * esz / 1e9:.1f} GB {str(self.dtype).split('.')[-1]}, "
                     f"{dist_name} scale 0.02), int8 per-channel error statistics")
        self.desc += (" (per-layer tensor MSE, mean and exact P99 of the per-vector MSE list; one model-wide P99 over all "
                      f"layers' per-channel lists), layers sharded over {world} rank(s) by longest-processing-time")
        self.kernel_desc = ("statistics-only fused quantize pass (k_stats_row / k_stats_group: ONE launch per pass over all of the "
                            "rank's layers whose segments have one length -- per-group passes the whole shard, per-channel passes "
                            "the layers of one column count)")
        if self.fused:
            npf = max(len(f["slots"]) for f in self.plan.families)
            self.kernel_desc = (f"statistics-only fused quantize pass over {npf} groupings per read (k_stats_multi: the passes of a "
                                "format that differ only in the grouping -- per row, g = 32 / 64 / 128 -- share ONE read of a "
                                "layer; launches per format and column count, in chunks of layers). achieved counts the bytes the "
                                "launch moves (each weight once, every list), not once per grouping: the launch is bound by "
                                "instruction issue, not by HBM")
        elif self.numpy_order:
            self.kernel_desc = ("statistics-only fused quantize pass in numpy's summation order (k_np_group3: g = 32 / 64 / 128 "
                                "from ONE read of a chunk of layers; k_np_row / k_np_rowx: the per-channel pass; both emit the float32 "
                                "pairwise sums of numpy's 8192-element chunks, the finishing launches add them in order)")

    def step(self):
        import torch
        from quantization_amd.distributed import gather_records, joint_percentile, reduce_table
        from quantization_amd.report import RECORD_FIELDS, SweepPlan
        local, vecs = None, []
        if self.weights:
            # the per-channel lists of the first format feed the model-wide percentile
            self.plan.launch(overlap=self.overlap)
            vecs = self.plan.lists(self.joint_pass)
        # RCCL: the histogram all-reduces (one per radix level) of a P99 taken over every rank's lists -- enqueued
        # behind the last quantize launch, beside the last pass's per-layer selections on the second stream ...
        model_p99 = joint_percentile(vecs, 99.0, dtype=torch.float32, total=self.total_vectors, wide=True)
        if self.weights:
            local = self.plan.finish().reshape(-1, len(RECORD_FIELDS))
        # ... and one all-reduce of the zero-filled record table of the whole sweep
        if self.table_mode:
            table = reduce_table(local)
        else:
            table = gather_records(self.index, local, self.n_layers * len(self.passes), len(RECORD_FIELDS), self.device)
        # the statistics leave the device inside the step: the record table (tens of KiB) and the model-wide P99 go
        # to pinned host memory on the step's stream; the timed region ends with a synchronize, so they have arrived
        if self.host_table is None:
            self.host_table = torch.empty(table.shape, dtype=table.dtype, pin_memory=True)
            self.host_p99 = torch.empty((), dtype=model_p99.dtype, pin_memory=True)
        store_host([self.host_table, self.host_p99], [table, model_p99])
        return self.host_table, self.host_p99


class Workload:
    """Single-GPU workloads: passes are grouped into batches of same-shaped tensors: one fused launch (per 32
    tensors) quantizes a whole batch, one set of select launches takes all its P99s."""

    def __init__(self, name, device, rank, dist_name="normal"):
        import torch
        from quantization_amd import _lib
        self.name = name
        g = torch.Generator(device=device)
        TINYLLAMA = [(2048, 2048)] * 2 + [(256, 2048)] * 2 + [(5632, 2048)] * 2 + [(2048, 5632)]
        LLAMA70B = [(8192, 8192)] * 2 + [(1024, 8192)] * 2 + [(28672, 8192)] * 2 + [(8192, 28672)]

        def draw(shape, dtype):
            """N(0, 0.02^2), or Student-t (3 degrees of freedom) scaled by 0.02: the heavy-tailed case."""
            x = torch.randn(*shape, generator=g, device=device)
            if dist_name == "student_t":
                chi = torch.zeros_like(x)
                for _ in range(3):
                    chi += torch.randn(*shape, generator=g, device=device) ** 2
                x = x / torch.sqrt(chi / 3)
            return (x * 0.02).to(dtype)

        self.compare = None
        self.sweep = None     # llama70b_shard: (weights, passes) of a report.SweepPlan
        self.plan = None
        self.host_rec = None
        self.overlap = True
        self.batches = []  # (stacked weights [B, rows, cols], fmt, datatype, gs, outputs dict)
        self.results = {}
        if name == "c2":
            self.desc = (f"c2: 16x 4096x4096 fp16 {dist_name} (scale 0.02), int8-asym g=32/64/128: deq + codes + scale + zp + "
                         "abs-err(f32) dump + per-group/tensor MSE + exact P99 of |err| and of the per-group MSE")
            outs = dict(deq=True, codes=True, scale=True, abs_err=torch.float32, seg_mse=True, totals=True)
            g.manual_seed(1000 * rank)
            w = draw((16, 4096, 4096), torch.float16)
            for gs in (32, 64, 128):
                self.batches.append((w, _lib.INT_ASYM, None, gs, outs))
        else:
            if name == "c3":
                shapes, dt = TINYLLAMA * 22 + [(32000, 2048)], torch.float16
            else:   # one GPU's share of the 80 decoder layers (the 561-tensor model split over 8 GPUs)
                shapes, dt = LLAMA70B * 10, torch.bfloat16
            stats = dict(deq=False, codes=False, scale=False, abs_err=None, seg_mse=True, totals=True)
            if name == "c3":
                passes = [(_lib.INT_SYM, None, None), (_lib.LUT_FP, "fp8_e4m3", None), (_lib.LUT_FP, "fp8_e3m4", None)]
                outs = dict(deq=True, codes=False, scale=False, abs_err=None, seg_sse=False, totals=True)
                self.compare = []
                self.desc = (f"c3: TinyLlama-shaped {dist_name} fp16 linear weights ({len(shapes)} tensors): int8-sym, fp8_e4m3, "
                             "fp8_e3m4 per-channel dequantized weights + tensor MSE, then the element-wise best-format / "
                             "outlier masks of compare.py")
            else:
                passes = [(f, d, gs) for f, d in ((_lib.INT_SYM, None), (_lib.INT_ASYM, None), (_lib.LUT_FP, "fp8_e4m3"),
                                                  (_lib.LUT_FP, "fp8_e3m4")) for gs in SWEEP_ORDER]
                outs = stats
                self.sweep = ([], [(n, 8, gs) for n in ("int8", "int8_asym", "fp8_e4m3", "fp8_e3m4") for gs in SWEEP_ORDER])
                self.desc = (f"llama70b_shard: 1/8 of Llama-3-70B's linear layers ({len(shapes)} tensors, 17.1 GB bf16 {dist_name}): "
                             "int8 sym / asym, fp8_e4m3, fp8_e3m4 x per-channel / g=32 / 64 / 128 error statistics "
                             "(per-vector/tensor MSE, sum|err|, exact P99 of the per-vector MSE)")
            groups = {}
            for sh in shapes:
                groups[sh] = groups.get(sh, 0) + 1
            for k, ((r, c), cnt) in enumerate(groups.items()):
                for lo in range(0, cnt, 32):
                    b = min(32, cnt - lo)
                    g.manual_seed(1000 * rank + 17 * k + lo)
                    w = draw((b, r, c), dt)
                    if self.sweep is not None:
                        self.sweep[0].extend(w.unbind(0))
                    first = len(self.batches)
                    for fmt, table, gs in passes:
                        self.batches.append((w, fmt, table, gs, outs))
                    if self.compare is not None:
                        self.compare.append((w, list(range(first, first + len(passes))),
                                             torch.empty(w.shape, dtype=torch.uint8, device=device)))
        self.kernel_desc = ("fused quantize pass (k_quant_team: one launch per batch of same-shaped tensors, feeding the "
                            "P99 selection)" if name == "c2" else
                            "statistics-only fused quantize pass (k_stats_row / k_stats_group)" if name == "llama70b_shard"
                            else "fused quantize pass (k_quant_wave / k_quant_block: dequantized weights)")
        self.tensors = sum(w.shape[0] for w, *_ in self.batches)
        # weight bytes of a step: every pass over a tensor counts (a sweep quantizes the same weights per format)
        self.weight_bytes = sum(w.numel() * w.element_size() for w, *_ in self.batches)
        self.algo_bytes = sum(w.numel() * algorithmic_bytes_per_elem(w.element_size(), f in (_lib.LUT_FP, _lib.LUT_FP_SUB),
                                                                     f == _lib.INT_ASYM, gs, o)
                              + (4 * w.shape[0] * w.shape[1] if (gs is None and o.get("seg_mse")) else 0)
                              for w, f, _, gs, o in self.batches)
        if self.sweep is not None:
            # the model-sweep entry point over this GPU's resident tensors; its fed lists are not written (see ModelSweep)
            from quantization_amd.report import SweepPlan
            self.plan = SweepPlan(*self.sweep)
            self.algo_bytes = self.plan.read_bytes + self.plan.list_bytes
            if self.plan.families:
                self.kernel_desc = ("statistics-only fused quantize pass over 4 groupings per read (k_stats_multi: per row, "
                                    "g = 32 / 64 / 128 of one format from ONE read of a layer); achieved counts the bytes the "
                                    "launch moves, not once per grouping: the launch is bound by instruction issue, not by HBM")
        self.launches = len(self.batches)
        self.device = device
        self.record = torch.zeros(self.tensors, 5, dtype=torch.float64, device=device)
        counts = []
        for w, fmt, dt, gs, outs in self.batches:
            counts += [w.shape[1] * w.shape[2]] * w.shape[0]
        self.numel = torch.tensor(counts, dtype=torch.float64, device=device)
        self.totals = torch.zeros(self.tensors, 2, dtype=torch.float64, device=device)

    def step(self):
        """All fused quantize launches of the step, then ONE set of select launches per dtype for the P99s of
        every tensor of the step (the chains of short kernels run once per step, not once per batch)."""
        import torch
        from quantization_amd.error_metrics import percentile_rows
        from quantization_amd.fused import QuantResult, quantize_batch
        if self.sweep is not None:
            # records [passes, tensors, fields]
            return self.to_host(self.plan.run(overlap=self.overlap))
        if self.name == "c2":
            # the three group sizes over one stack go to the library together: the short selection kernels of a pass
            # (sampling, brackets, the list percentile) run beside the neighbouring full passes
            from quantization_amd.fused import passes_percentiles, quantize_batch_passes
            w = self.batches[0][0]
            prev = self.results.get("passes")
            if prev is None:
                prev = [QuantResult(0, 0, 0) for _ in self.batches]
                for k, r in enumerate(prev):
                    r._bufs = {"totals": self.totals[k * w.shape[0]:(k + 1) * w.shape[0]]}
            ps = [dict(fmt=fmt, bits=8, group_size=gs, datatype=dt, **{a: b for a, b in outs.items() if a != "abs_err"})
                  for _, fmt, dt, gs, outs in self.batches]
            results = quantize_batch_passes(w, ps, p99=99, list_p99=99, out=prev)
            self.results["passes"] = results
            rec = self.record
            rec[:, 0:2] = self.totals
            torch.div(self.totals[:, 0], self.numel, out=rec[:, 2])
            abs_p99, list_p99 = passes_percentiles(results)
            rec[:, 3] = list_p99
            rec[:, 4] = abs_p99
            return self.to_host(rec)
        results, row = [], 0
        for k, (w, fmt, dt, gs, outs) in enumerate(self.batches):
            res = self.results.get(k)
            if res is None:
                res = QuantResult(0, 0, 0)
                res._bufs = {"totals": self.totals[row:row + w.shape[0]]}
            # the exact P99 of |err| is selected inside the quantize pass (fed selection)
            res = quantize_batch(w, fmt, 8, gs, dt, out=res, p99=99 if outs.get("abs_err") is not None else None, **outs)
            assert res.totals.data_ptr() == self.totals[row].data_ptr()
            self.results[k] = res
            results.append(res)
            row += w.shape[0]
        rec = self.record
        if self.compare is not None:
            # compare.py:99-125 on the three dequantized stacks of every batch: flags per element
            from quantization_amd.error_metrics import compare_formats
            for w, idxs, flags in self.compare:
                compare_formats(w, [results[i].deq for i in idxs], 0.02, out=flags)
            rec[:, 0:2] = self.totals
            return self.to_host(rec)
        rec[:, 0:2] = self.totals
        # mean of a tensor's per-vector MSE list = its tensor MSE (all vectors have one length); the P99 is selected
        # on the float32 list entries and interpolated in float64, as np.percentile of the scripts' float64 list
        rec[:, 2] = self.totals[:, 0] / self.numel
        rec[:, 3] = percentile_rows([res.seg_mse for res in results], 99, wide=True)
        if results[0].abs_p99 is not None:
            rec[:, 4] = torch.cat([res.abs_p99 for res in results])     # exact P99 of |err| (float32)
        return self.to_host(rec)

    def to_host(self, rec):
        """The step's statistics leave the device inside the step (pinned memory, the step's stream)."""
        import torch
        if self.host_rec is None:
            self.host_rec = torch.empty(rec.shape, dtype=rec.dtype, pin_memory=True)
        store_host([self.host_rec], [rec])
        return self.host_rec


def cpu_baseline():
    """The CPU port (oracle/: C restatement of quant_weight.py + the analysis scripts' numpy) on a bounded sample of
    the headline workload: ONE 4096 x 4096 bf16 matrix x {per-channel, g = 32 / 64 / 128}, symmetric int8, tensor
    MSE, per-vector MSE list, its mean and P99 -- timed with one thread and with the CPU share this box gives a
    one-GPU job."""
    import numpy as np
    from oracle import oracle as orc
    rng = np.random.default_rng(0)
    w32 = (rng.standard_normal((4096, 4096)) * 0.02).astype(np.float32)
    wbf = (w32.view(np.uint32) >> 16).astype(np.uint16)            # bfloat16 bits (truncation is fine for a timing input)
    share = int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)

    def sweep(nthreads, budget):
        orc.quantize(wbf[:64], orc.BF16, orc.INT_SYM, 8, 128, nthreads=nthreads)  # load the library
        t0, passes = time.perf_counter(), 0
        while time.perf_counter() - t0 < budget:
            for gs in GROUPS:
                o = orc.quantize(wbf, orc.BF16, orc.INT_SYM, 8, gs, nthreads=nthreads)
                a32 = o["v16"].view(np.float16).astype(np.float32)
                b32 = o["deq"].view(np.float16).astype(np.float32)
                orc.compute_mse(a32, b32)
                d = (a32 - b32).reshape(-1, gs or 4096)
                vec = np.mean(d * d, axis=1, dtype=np.float32).astype(np.float64)   # single_weight.py:61-79, vectorised
                vec.mean(), np.percentile(vec, 99)
                passes += 1
        return passes * wbf.nbytes / (time.perf_counter() - t0) / 1e9, passes

    one, n1 = sweep(1, 6.0)
    many, nm = sweep(share, 8.0)
    return {"value": round(many, 4), "unit": "GB/s", "cores": share, "kind": "port",
            "host_cpu_count": os.cpu_count(), "value_1_thread": round(one, 4),
            "sample": f"one 4096x4096 bf16 matrix x per-channel / g=32 / 64 / 128 symmetric int8: C oracle quantize + numpy "
                      f"tensor MSE + per-vector MSE list, mean, P99; {nm} passes with {share} OpenMP threads (the CPU share "
                      f"of a one-GPU job on this box; the host has {os.cpu_count()} CPUs), {n1} passes with 1 thread",
            "reference_build_container": REFERENCE_BUILD_CONTAINER}


def timed_run(step, steps, warmup, device, dist, world, reserve):
    """W warm-up steps, then K steps between barrier + synchronize pairs; -> (wall max over ranks, GPU event ms,
    summed dominant-kernel ms, dominant-kernel launches)."""
    import torch
    from quantization_amd import _lib
    L = _lib.lib()
    _lib.check(L.mq_profile_reserve(int(reserve * (steps + 1)) + 64))
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    _lib.check(L.mq_profile_begin())
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    tmax = torch.tensor([wall], dtype=torch.float64, device=device)
    if world > 1:
        from quantization_amd.distributed import all_reduce_
        all_reduce_(tmax, op=dist.ReduceOp.MAX)
    kern_total_ms, kern_launches = ctypes.c_double(0), ctypes.c_int64(0)
    _lib.check(L.mq_profile_end(ctypes.byref(kern_total_ms), ctypes.byref(kern_launches)))
    return float(tmax.item()), ev0.elapsed_time(ev1), kern_total_ms.value, kern_launches.value


def roofline(algo_bytes_per_step, steps, kern_ms, launches, kernel_desc, traffic):
    per_launch = algo_bytes_per_step * steps / max(1, launches)
    avg_ms = kern_ms / max(1, launches)
    achieved = per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    return {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "kernel": kernel_desc,
            "launches_timed": int(launches), "avg_launch_us": round(avg_ms * 1e3, 2),
            "algorithmic_bytes_per_launch": int(per_launch)}


def launch_alone(wl, device, dist, world, reserve, algo_bytes):
    """The sweep's quantize launches with nothing else on the chip, outside the timed region: in the timed steps the
    selections of the previous pass run on a second stream and share the HBM with them, which lengthens each launch."""
    wl.overlap = False
    _, _, k_iso, n_iso = timed_run(wl.step, 2, 1, device, dist, world, reserve)
    wl.overlap = True
    if not n_iso:
        return None
    iso = roofline(algo_bytes, 2, k_iso, n_iso, "", None)
    return {"avg_launch_us": iso["avg_launch_us"], "achieved": iso["achieved"], "frac": iso["frac"],
            "note": "the same launches in 2 extra untimed steps with the selections kept on the same stream: in the "
                    "timed steps the previous pass's selections run concurrently on a second stream and share the HBM"}


def traffic_for(name):
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        return json.load(open(tpath)).get(name)
    return None


def issue_for(name):
    """The vector ALUs' busy fraction under the workload's dominant kernel, from the committed SQ counter pass
    (scripts/issue_from_pmc.py): what bounds a launch whose HBM fraction is low on purpose."""
    ipath = os.path.join(ROOT, "profiles", "issue.json")
    if os.path.exists(ipath):
        d = json.load(open(ipath)).get(name)
        if d and d.get("valu_busy") is not None:
            return {"valu_busy": d["valu_busy"], "source": "rocprofv3 --pmc SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE ... "
                    "(profiles/issue.json): SIMD cycles with a vector instruction executing / all SIMD cycles of the launch"}
    return None


def main():
    args = parse_args()
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    world = int(world_env or "1")
    # first contact with RCCL must fail loudly, not hang: with more than one rank every rank arms a watchdog that
    # prints all Python stacks and leaves with a non-zero status (no re-exec; the launcher then ends the others)
    watchdog = args.watchdog if args.watchdog is not None else (300 if world > 1 else 0)
    if watchdog > 0:
        import faulthandler
        faulthandler.dump_traceback_later(watchdog, exit=True)
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} does not match WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    import torch
    from quantization_amd import _lib
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("MQ_BENCH_BACKEND", "nccl")
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            if torch.cuda.device_count() < world:
                print(f"bench.py: {world} ranks but {torch.cuda.device_count()} GPU(s)", file=sys.stderr)
                sys.exit(2)
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:   # rehearsal: ranks may share a GPU
            torch.cuda.set_device(local_rank % max(1, torch.cuda.device_count()))
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
    device = torch.device("cuda", torch.cuda.current_device())
    _lib.lib()

    sharded = args.workload in MODEL_OF
    if sharded:
        wl = ModelSweep(args.workload, device, rank, world, args.dist, numpy_order=args.numpy_order)
        total_weight_bytes, local_algo, reserve = wl.total_weight_bytes, wl.local_algo_bytes, wl.launches_hint
    else:
        if world > 1:
            print(f"bench.py: workload {args.workload} is a single-GPU workload", file=sys.stderr)
            sys.exit(2)
        wl = Workload(args.workload, device, rank, args.dist)
        total_weight_bytes, local_algo, reserve = wl.weight_bytes, wl.algo_bytes, wl.launches
    if sharded and world > 1:
        # every rank names its shard and what it sees before the first collective of the data path
        print(f"bench.py rank {rank}/{world}: device {device} ({torch.cuda.get_device_name(device)}), backend {backend}, "
              f"dist world {dist.get_world_size()}, {len(wl.mine)} layers, {wl.local_weight_bytes} weight bytes per step, "
              f"watchdog {watchdog} s", file=sys.stderr, flush=True)
    wall, gpu_ms, kern_ms, kern_launches = timed_run(wl.step, args.steps, args.warmup, device, dist, world, reserve)
    isolated = None
    if sharded or getattr(wl, "sweep", None) is not None:
        isolated = launch_alone(wl, device, dist, world, reserve, local_algo)

    if args.dump_table and sharded and rank == 0:
        import numpy as np
        table, p99 = wl.step()
        torch.cuda.synchronize()
        np.savez(args.dump_table, table=table.numpy(), p99=p99.numpy(), layers_mine=np.asarray(wl.mine))
    elif args.dump_table and sharded:
        wl.step()   # (the collectives of rank 0's extra step need every rank)

    # per-rank shares, for the record
    shares = None
    if sharded:
        # (device tensors for RCCL, host tensors for the gloo rehearsal)
        mine = torch.tensor([wl.local_weight_bytes, len(wl.mine)], dtype=torch.int64,
                            device=device if backend == "nccl" else "cpu")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        if world > 1:
            torch.cuda.synchronize()
            dist.all_gather(allr, mine)
        else:
            allr = [mine]
        shares = [[int(v) for v in t.tolist()] for t in allr]

    out = None
    if rank == 0:
        out = {
            "metric": "weight GB/s quant+dequant+MSE/P99",
            "value": round(total_weight_bytes * args.steps / wall / 1e9, 2),
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(wall / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong" if sharded else "weak",
            "vs_baseline": None,
            "dtype": "f16",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {wl.desc}",
                       "passes_per_step": len(wl.passes) if sharded else wl.launches,
                       **({"weight_bytes_read_per_step_rank0": int(wl.local_read_bytes)} if sharded else {}),
                       "weight_bytes_per_step_whole_job": int(total_weight_bytes),
                       "gpu_ms_per_step_events_rank0": round(gpu_ms / args.steps, 4),
                       **({"numpy_order": True} if args.numpy_order else {})},
            "roofline": roofline(local_algo, args.steps, kern_ms, kern_launches, wl.kernel_desc + " (rank 0)",
                                 # (the PMC passes measured the one-GPU launches: a shard's launches are smaller)
                                 traffic_for(args.workload) if world == 1 and not args.numpy_order else None),
        }
        if isolated:
            out["roofline"]["launch_alone"] = isolated
        if world == 1 and not args.numpy_order and issue_for(args.workload):
            out["roofline"]["issue"] = issue_for(args.workload)
        if sharded and world == 1 and getattr(wl, "fused", False) and args.workload == "llama8b" and not args.no_extra:
            # the same sweep with ONE launch per pass (every pass reads the weights itself): the HBM-bound form, for
            # comparison -- its own weights and lists, 3 steps outside the headline's timed region
            w1 = ModelSweep(args.workload, device, rank, world, args.dist, fuse=False)
            wall1, _, k1, n1 = timed_run(w1.step, 3, 1, device, None, 1, w1.launches_hint)
            r1 = roofline(w1.local_algo_bytes, 3, k1, n1, "", None)
            a1 = launch_alone(w1, device, None, 1, w1.launches_hint, w1.local_algo_bytes)
            out["roofline"]["one_launch_per_pass"] = {
                "ms_per_step": round(wall1 / 3 * 1e3, 4), "value": round(w1.total_weight_bytes * 3 / wall1 / 1e9, 2),
                "avg_launch_us": r1["avg_launch_us"], "achieved": r1["achieved"], "frac": r1["frac"],
                "launch_alone_frac": a1["frac"] if a1 else None,
                "note": "k_stats_row / k_stats_group, one launch per (format, grouping) pass: each pass streams the weights "
                        "from HBM again"}
            del w1
            torch.cuda.empty_cache()
        if sharded:
            out["config"].update({
                "backend": backend if world > 1 else "none (one rank)",
                "rccl_ranks": dist.get_world_size() if (world > 1 and backend == "nccl") else 0,
                "collectives_per_step": "1 all-reduce of the record table + 2 histogram all-reduces (joint P99 over float32 keys)"
                if world > 1 else "none",
                "weight_bytes_per_step_per_rank": [s[0] for s in shares], "layers_per_rank": [s[1] for s in shares]})

    # N = 1: the 1-GPU leg of every other configuration, driver-timed in the same run, and the CPU baseline
    if world == 1 and rank == 0 and sharded and args.workload == "llama8b" and not args.no_extra:
        del wl
        torch.cuda.empty_cache()
        out["extra"] = {}
        extras = (("c1_tinyllama", "tinyllama", 20, 3), ("c2", "c2", 20, 3), ("c3", "c3", 10, 2),
                  ("llama70b_shard", "llama70b_shard", 3, 1), ("c5_llama70b", "llama70b", 2, 1))
        for key, name, steps, warm in extras:
            if name in MODEL_OF:
                if name == "llama70b" and torch.cuda.mem_get_info(device)[0] < 230e9:
                    out["extra"][key] = {"skipped": "needs 139 GB of weights + 61 GB of per-vector lists resident"}
                    continue
                w2 = ModelSweep(name, device, 0, 1, args.dist)
                wbytes, algo, hint, desc = w2.total_weight_bytes, w2.local_algo_bytes, w2.launches_hint, w2.desc
            else:
                w2 = Workload(name, device, 0, args.dist)
                wbytes, algo, hint, desc = w2.weight_bytes, w2.algo_bytes, w2.launches, w2.desc
            wall2, gpu2, k2, n2 = timed_run(w2.step, steps, warm, device, None, 1, hint)
            out["extra"][key] = {
                "value": round(wbytes * steps / wall2 / 1e9, 2), "unit": "GB/s", "steps": steps, "warmup": warm,
                "ms_per_step": round(wall2 / steps * 1e3, 4), "workload": f"{name}: {desc}",
                "weight_bytes_per_step": int(wbytes),
                "roofline": roofline(algo, steps, k2, n2, w2.kernel_desc, traffic_for(name))}
            if name in MODEL_OF or w2.sweep is not None:
                out["extra"][key]["roofline"]["launch_alone"] = launch_alone(w2, device, None, 1, hint, algo)
            if issue_for(name):
                out["extra"][key]["roofline"]["issue"] = issue_for(name)
            del w2
            torch.cuda.empty_cache()
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
