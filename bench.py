#!/usr/bin/env python3
"""Benchmark of the quantize -> dequantize -> error hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|tinyllama|llama8b|llama70b]
                    [--dist normal|student_t]

For N > 1 the driver starts one rank per GPU with torch.distributed.run; every rank processes its own
copy of the workload (weak scaling: the units are independent weight tensors, no data-path collective)
and the per-tensor statistics records are all-gathered once per step over RCCL.

Default workload "c2" (BASELINE.json configs[1]): 4096x4096 fp16 matrices, asymmetric int8 with
group sizes 32 / 64 / 128, returning dequantized weights, codes, scales, zero-points and the float32
element-wise absolute-error dump, plus per-group and per-tensor squared-error sums, the mean and exact
P99 of the per-group MSE vector and the exact P99 of the absolute error.  A step is one
pass over a batch of 16 distinct matrices (512 MiB of weights, 5.4 GB of outputs -- more than the
256 MiB Infinity Cache, so inputs come from HBM) x 3 group sizes = 48 tensor passes in 3 batched launches.

Other workloads (BASELINE.json configs[0], [2], [3], [4]; not the headline line): "tinyllama" / "llama8b" --
all linear weights of that shape list, int8-sym per-channel, statistics only; "c3" -- TinyLlama shapes, int8-sym /
fp8_e4m3 / fp8_e3m4 per-channel dequantized + the element-wise best-format masks of elwise_error/compare.py;
"llama70b" -- one GPU's eighth of the Llama-3-70B linear layers (10 decoder layers, 17 GB bf16), int8 sym / asym,
fp8_e4m3, fp8_e3m4 x {per-channel, g = 32 / 64 / 128}, statistics only.

One JSON line is printed by rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import quantization_amd as qa  # noqa: E402
from quantization_amd import _lib  # noqa: E402
from quantization_amd.error_metrics import percentile_multi, true_div  # noqa: E402
from quantization_amd.fused import QuantResult, quantize_batch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak

TINYLLAMA = [(2048, 2048)] * 2 + [(256, 2048)] * 2 + [(5632, 2048)] * 2 + [(2048, 5632)]
LLAMA8B = [(4096, 4096)] * 2 + [(1024, 4096)] * 2 + [(14336, 4096)] * 2 + [(4096, 14336)]
LLAMA70B = [(8192, 8192)] * 2 + [(1024, 8192)] * 2 + [(28672, 8192)] * 2 + [(8192, 28672)]


def model_shapes(per_layer, layers, head):
    return per_layer * layers + [head]


def algorithmic_bytes_per_elem(in_bytes, fmt, gs, outputs):
    """HBM bytes one fused launch must move per weight element: input once, plus each requested output."""
    b = float(in_bytes)
    if outputs.get("deq"):
        b += 2
    if outputs.get("codes"):
        b += 2 if fmt == _lib.LUT_FP else 1
    if outputs.get("abs_err") == torch.float32:
        b += 4
    elif outputs.get("abs_err") == torch.float16:
        b += 2
    if gs and gs > 0:
        per_seg = (2 if outputs.get("scale") else 0) * (2 if fmt == _lib.INT_ASYM else 1)
        per_seg += 8 if outputs.get("seg_sse") else 0
        b += per_seg / gs
    return b


class Workload:
    """Passes are grouped into batches of same-shaped tensors: one fused launch (per 32 tensors) quantizes a
    whole batch, one set of select launches takes all its P99s."""

    def __init__(self, name, device, rank, dist_name="normal"):
        self.name = name
        g = torch.Generator(device=device)

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
        self.batches = []  # (stacked weights [B, rows, cols], fmt, datatype, gs, outputs dict)
        self.results = {}
        if name == "c2":
            self.desc = (f"16x 4096x4096 fp16 {dist_name} (scale 0.02), int8-asym g=32/64/128: deq + codes + scale + zp + "
                         "abs-err(f32) dump + per-group/tensor MSE + exact P99 of |err| and of the per-group MSE")
            outs = dict(deq=True, codes=True, scale=True, abs_err=torch.float32, seg_sse=True, totals=True)
            g.manual_seed(1000 * rank)
            w = draw((16, 4096, 4096), torch.float16)
            for gs in (32, 64, 128):
                self.batches.append((w, _lib.INT_ASYM, None, gs, outs))
        else:
            if name in ("tinyllama", "c3"):
                shapes, dt = model_shapes(TINYLLAMA, 22, (32000, 2048)), torch.float16
            elif name == "llama8b":
                shapes, dt = model_shapes(LLAMA8B, 32, (128256, 4096)), torch.bfloat16
            else:   # one GPU's share of the 80 decoder layers (the 561-tensor model split over 8 GPUs)
                shapes, dt = LLAMA70B * 10, torch.bfloat16
            stats = dict(deq=False, codes=False, scale=True, abs_err=None, seg_sse=True, totals=True)
            if name == "c3":
                passes = [(_lib.INT_SYM, None, None), (_lib.LUT_FP, "fp8_e4m3", None), (_lib.LUT_FP, "fp8_e3m4", None)]
                outs = dict(deq=True, codes=False, scale=False, abs_err=None, seg_sse=False, totals=True)
                self.compare = []
                self.desc = (f"TinyLlama-shaped {dist_name} fp16 linear weights ({len(shapes)} tensors): int8-sym, fp8_e4m3, "
                             "fp8_e3m4 per-channel dequantized weights + tensor MSE, then the element-wise best-format / "
                             "outlier masks of compare.py")
            elif name == "llama70b":
                passes = [(f, d, gs) for f, d in ((_lib.INT_SYM, None), (_lib.INT_ASYM, None), (_lib.LUT_FP, "fp8_e4m3"),
                                                  (_lib.LUT_FP, "fp8_e3m4")) for gs in (None, 32, 64, 128)]
                outs = stats
                self.desc = (f"1/8 of Llama-3-70B's linear layers ({len(shapes)} tensors, bf16 {dist_name}): int8 sym / asym, "
                             "fp8_e4m3, fp8_e3m4 x per-channel / g=32 / 64 / 128, stats only (per-vector/tensor MSE, "
                             "sum|err|, exact P99 of the per-vector MSE)")
            else:
                passes = [(_lib.INT_SYM, None, None)]
                outs = stats
                self.desc = (f"{name}-shaped random-init {str(dt).split('.')[-1]} linear weights ({len(shapes)} tensors), "
                             "int8-sym per-channel, stats only (per-row/tensor MSE, sum|err|, exact P99 of the per-row MSE)")
            groups = {}
            for sh in shapes:
                groups[sh] = groups.get(sh, 0) + 1
            for k, ((r, c), cnt) in enumerate(groups.items()):
                for lo in range(0, cnt, 32):
                    b = min(32, cnt - lo)
                    g.manual_seed(1000 * rank + 17 * k + lo)
                    w = draw((b, r, c), dt)
                    first = len(self.batches)
                    for fmt, table, gs in passes:
                        self.batches.append((w, fmt, table, gs, outs))
                    if self.compare is not None:
                        self.compare.append((w, list(range(first, first + len(passes))),
                                             torch.empty(w.shape, dtype=torch.uint8, device=device)))
        self.kernel_desc = ("fused quantize pass (k_quant_team/wave/block: one launch per batch of same-shaped tensors"
                            + (", feeding the P99 selection)" if name == "c2" else ")"))
        self.tensors = sum(w.shape[0] for w, *_ in self.batches)
        # weight bytes of a step: every pass over a tensor counts (a sweep quantizes the same weights per format)
        self.weight_bytes = sum(w.numel() * w.element_size() for w, *_ in self.batches)
        self.algo_bytes = sum(w.numel() * algorithmic_bytes_per_elem(w.element_size(), f, gs, o)
                              for w, f, _, gs, o in self.batches)
        self.launches = len(self.batches)
        self.device = device
        self.record = torch.zeros(self.tensors, 5, dtype=torch.float64, device=device)
        # per-tensor constants of the statistics step, resident on the device
        lens, counts = [], []
        for w, fmt, dt, gs, outs in self.batches:
            L = gs if gs else w.shape[2]
            lens += [L] * w.shape[0]
            counts += [w.shape[1] * w.shape[2]] * w.shape[0]
        self.post_div = torch.tensor([L if L & (L - 1) == 0 else 1 for L in lens], dtype=torch.float64, device=device)
        self.numel = torch.tensor(counts, dtype=torch.float64, device=device)
        self.totals = torch.zeros(self.tensors, 2, dtype=torch.float64, device=device)

    def step(self):
        """All fused quantize launches of the step, then ONE set of select launches per dtype for the P99s of
        every tensor of the step (the chains of short kernels run once per step, not once per batch)."""
        results, row = [], 0
        for k, (w, fmt, dt, gs, outs) in enumerate(self.batches):
            # one preallocated output set per batch: outputs cycle through > 256 MiB so they reach HBM; the
            # [sum err^2, sum |err|] pairs of all batches land in one table
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
            return [rec]
        rec[:, 0:2] = self.totals
        # mean of the per-(row, group) / per-row MSE list = sum of squares / (L * nseg): equal lengths
        rec[:, 2] = self.totals[:, 0] / self.numel
        # exact P99 of every per-vector MSE list (float64 select).  A power-of-two L divides after the select
        # (an exponent shift commutes with it); any other length divides its list first (IEEE division, as the
        # scripts divide), per batch.
        vecs = []
        for (w, fmt, dt, gs, outs), res in zip(self.batches, results):
            L = gs if gs else w.shape[2]
            per_vec = res.seg_sse if L & (L - 1) == 0 else true_div(res.seg_sse, L)
            vecs += [per_vec[i] for i in range(per_vec.shape[0])]
        rec[:, 3] = percentile_multi(vecs, 99) / self.post_div
        if results[0].abs_p99 is not None:
            rec[:, 4] = torch.cat([res.abs_p99 for res in results])     # exact P99 of |err| (float32)
        return [rec]


def cpu_baseline(nthreads):
    """The CPU port (oracle/: C restatement of quant_weight.py + the analysis scripts' numpy) on a bounded
    sample of the c2 workload: ONE 4096x4096 fp16 matrix x g = 32 / 64 / 128, asymmetric int8 with
    deq/codes/scale/zp, float32 error, tensor MSE, abs-error sum and P99, per-group MSE vector + P99."""
    from oracle import oracle as orc
    rng = np.random.default_rng(0)
    w = (rng.standard_normal((4096, 4096)) * 0.02).astype(np.float16)
    orc.quantize(w[:64], orc.F16, orc.INT_ASYM, 8, 128, nthreads=nthreads)  # load the library
    t0 = time.perf_counter()
    passes = 0
    while time.perf_counter() - t0 < 10.0:   # bounded sample: whole 3-pass sweeps until 10 s have gone
        for gs in (32, 64, 128):
            o = orc.quantize(w, orc.F16, orc.INT_ASYM, 8, gs, nthreads=nthreads)
            deq = o["deq"].view(np.float16)
            a32, b32 = orc.proto_to_array(w), orc.proto_to_array(deq)
            orc.compute_mse(a32, b32)
            orc.elementwise_f32(w, deq)            # float32 |err|, its sum and np.percentile(., 99)
            # per-(row, group) MSE vector + mean + P99 (quant_error_single_weight.py:61-79), vectorised
            d = (a32 - b32).reshape(-1, gs)
            vec = np.mean(d * d, axis=1, dtype=np.float32).astype(np.float64)
            vec.mean(), np.percentile(vec, 99)
            passes += 1
    dt = time.perf_counter() - t0
    return {"value": round(passes * w.nbytes / dt / 1e9, 4), "unit": "GB/s", "cores": nthreads, "kind": "port",
            "sample": f"one 4096x4096 fp16 matrix x g=32/64/128, {passes} passes in {dt:.1f} s: C oracle quantize "
                      f"({nthreads} OpenMP threads) + numpy MSE / abs-err sum / P99 / per-group MSE vector + P99"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "tinyllama", "llama8b", "llama70b"])
    ap.add_argument("--dist", default="normal", choices=["normal", "student_t"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        # one rank per GPU over RCCL; MQ_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs
        backend = os.environ.get("MQ_BENCH_BACKEND", "nccl")
        local_rank %= max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
    device = torch.device("cuda", torch.cuda.current_device())
    _lib.lib()

    wl = Workload(args.workload, device, rank, args.dist)
    gathered = torch.zeros(world * wl.tensors, 5, dtype=torch.float64, device=device) if world > 1 else None

    def one_step():
        stats = wl.step()
        if world > 1:
            # per-tensor statistics records of every rank -> every rank, one RCCL collective per step
            dist.all_gather_into_tensor(gathered, torch.cat(stats))

    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    L = _lib.lib()
    _lib.check(L.mq_profile_begin())
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        one_step()
    ev1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    gpu_ms = ev0.elapsed_time(ev1)
    tmax = torch.tensor([wall], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    wall = float(tmax.item())

    # dominant kernel: the full quantize launches of the timed region itself, each bracketed by a pair of HIP
    # events on the launch stream inside the library (mq_profile_begin / mq_profile_end)
    kern_total_ms, kern_launches = ctypes.c_double(0), ctypes.c_int64(0)
    _lib.check(L.mq_profile_end(ctypes.byref(kern_total_ms), ctypes.byref(kern_launches)))
    kern_ms = kern_total_ms.value / max(1, kern_launches.value)
    algo_per_launch = wl.algo_bytes / wl.launches
    achieved = algo_per_launch / (kern_ms * 1e-3) / 1e9

    if rank == 0:
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get(args.workload)
        out = {
            "metric": "weight GB/s quant+dequant+MSE/P99",
            "value": round(world * wl.weight_bytes * args.steps / wall / 1e9, 2),
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(wall / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {wl.desc}", "fused_launches_per_step": wl.launches,
                       "tensor_passes_per_step": wl.tensors,
                       "weight_bytes_per_step_per_gpu": wl.weight_bytes,
                       "algorithmic_bytes_per_step_per_gpu": int(wl.algo_bytes),
                       "gpu_ms_per_step_events": round(gpu_ms / args.steps, 4)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": wl.kernel_desc, "launches_timed": int(kern_launches.value), "avg_launch_us": round(kern_ms * 1e3, 2),
                         "algorithmic_bytes_per_launch": int(algo_per_launch)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(min(int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1)), 16))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
