#!/usr/bin/env python3
"""How busy the vector ALUs are under the dominant kernel, from one rocprofv3 --pmc pass with
SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE:

    scripts/issue_from_pmc.py <counter_collection.csv> [workload]

Units on gfx950 (checked against each other here): SQ_ACTIVE_INST_VALU counts quad-cycles summed over every SIMD of
the chip; GRBM_GUI_ACTIVE counts cycles summed over the 8 XCDs; the chip has 1024 SIMDs.  So
    valu_busy = 4 * SQ_ACTIVE_INST_VALU / (1024 * GRBM_GUI_ACTIVE / 8)
is the fraction of all SIMD cycles of the launch in which a vector instruction was executing, and
4 * SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU the mean issue cost of one (about 4 cycles for a wave of 64 on 16 lanes).
The counter charges every vector instruction one quad-cycle (the quotient comes out at 4.01 .. 4.04 for every kernel
here), so a stream of instructions that really issue in ~2.4 cycles (v_and_b32, plain float32) reads a little high and
one with 8-cycle instructions (v_rcp_f32) a little low: valu_busy can pass 1.0 by a few percent.  A launch near 1.0 is
bound by instruction issue whatever its HBM fraction says.  Writes profiles/issue.json."""
import csv
import json
import os
import sys
from collections import defaultdict

SIMDS, XCDS = 1024, 8


def main():
    path = sys.argv[1]
    workload = sys.argv[2] if len(sys.argv) > 2 else "llama8b"
    acc = defaultdict(lambda: defaultdict(list))
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            acc[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    names = [k for k in acc if "k_stats_" in k] or [k for k in acc if "k_quant" in k or "k_np_" in k]
    rows, tot = {}, defaultdict(float)
    for k in sorted(names):
        c = {n: sum(v) for n, v in acc[k].items()}
        n = len(acc[k]["GRBM_GUI_ACTIVE"])
        if n == 0 or c["GRBM_GUI_ACTIVE"] <= 0:
            continue
        if c["GRBM_GUI_ACTIVE"] / n / XCDS < 50_000:     # (sampling passes and other short launches: not the dominant kernel)
            continue
        simd_cycles = SIMDS * c["GRBM_GUI_ACTIVE"] / XCDS
        rows[k] = {"launches": n, "valu_busy": round(4 * c["SQ_ACTIVE_INST_VALU"] / simd_cycles, 4),
                   "cycles_per_valu_inst": round(4 * c["SQ_ACTIVE_INST_VALU"] / c["SQ_INSTS_VALU"], 3),
                   "valu_insts_per_launch": round(c["SQ_INSTS_VALU"] / n),
                   "wave_cycles_waiting_on_memory_or_barrier": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 4),
                   "wave_cycles_waiting_to_issue": round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 4)}
        for n2 in ("SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE"):
            tot[n2] += c[n2]
        print(k, json.dumps(rows[k]))
    busy = round(4 * tot["SQ_ACTIVE_INST_VALU"] / (SIMDS * tot["GRBM_GUI_ACTIVE"] / XCDS), 4) if tot["GRBM_GUI_ACTIVE"] else None
    out_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "issue.json")
    data = json.load(open(out_path)) if os.path.exists(out_path) else {}
    data[workload] = {"valu_busy": busy, "kernels": rows}
    json.dump(data, open(out_path, "w"), indent=1)
    print("VALU busy over the dominant kernel's launches:", busy)


if __name__ == "__main__":
    main()
