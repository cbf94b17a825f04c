#!/usr/bin/env python3
"""HBM traffic per launch of the quantize / select kernels from two rocprofv3 --pmc passes
(FETCH_SIZE in one run, WRITE_SIZE in another; both counters are in KiB).

    scripts/traffic_from_pmc.py <fetch counter_collection.csv> <write counter_collection.csv> [workload]

gfx950 correction: FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 bytes, so the
read side is doubled (checked here on torch's float32 multiply over the 1 GiB weight stack, whose read
size is known); WRITE_SIZE is exact for 16-byte-per-lane stores.  Writes profiles/traffic.json."""
import csv
import json
import os
import sys
from collections import defaultdict


def per_kernel(path, counter):
    """Counter values per kernel name.  The quantize kernels that feed a selection (<..., true>) run twice per
    batch: the sampling pass over 1/32 of the weight and the full pass; the two are told apart by duration."""
    acc = defaultdict(list)
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].split("(")[0][:60]
            if "k_quant" in name and name.rstrip().endswith(("true>", " 1>", " 2>")):
                dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                name += " [full pass]" if dur > 150_000 else " [sampling pass]"
            acc[name].append(float(r["Counter_Value"]))
    return acc


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    workload = sys.argv[3] if len(sys.argv) > 3 else "c2"
    rows = {}
    for k in sorted(set(fetch) | set(write)):
        if not (k.startswith("void k_") or k.startswith("k_")):
            continue
        f = sum(fetch.get(k, [0])) / max(1, len(fetch.get(k, [0])))
        w = sum(write.get(k, [0])) / max(1, len(write.get(k, [0])))
        rows[k] = {"launches": len(fetch.get(k, [])), "fetch_bytes": 2 * f * 1024, "write_bytes": w * 1024,
                   "hbm_bytes": 2 * f * 1024 + w * 1024}
        print(f"{k:62s} n={rows[k]['launches']:4d} read {2*f*1024/1e6:10.1f} MB  write {w*1024/1e6:10.1f} MB")
    # the dominant kernel of the workload: the statistics-only kernels of the model sweeps, else the full quantize
    # passes; bytes per launch averaged over every such launch (as bench.py averages its durations)
    if any("k_stats_" in k for k in rows):
        quant = [v for k, v in rows.items() if "k_stats_" in k]
    else:
        quant = [v for k, v in rows.items() if "k_quant" in k and "[sampling pass]" not in k
                 and ("[full pass]" in k or not any("[full pass]" in q for q in rows))]
    per_launch = sum(v["hbm_bytes"] * v["launches"] for v in quant) / max(1, sum(v["launches"] for v in quant))
    out_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "traffic.json")
    data = json.load(open(out_path)) if os.path.exists(out_path) else {}
    data[workload] = round(per_launch)
    data[workload + "_detail"] = rows
    json.dump(data, open(out_path, "w"), indent=1)
    print("fused quantize HBM bytes per launch:", round(per_launch))


if __name__ == "__main__":
    main()
