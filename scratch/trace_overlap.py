"""From a rocprofv3 kernel trace CSV: time during which kernels of two different queues ran concurrently."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), 1, r["Queue_Id"]))
    ev.append((int(r["End_Timestamp"]), -1, r["Queue_Id"]))
ev.sort()
active = {}
last = None
both = one = 0
for t, d, q in ev:
    if last is not None:
        n = sum(1 for v in active.values() if v > 0)
        if n >= 2: both += t - last
        elif n == 1: one += t - last
    active[q] = active.get(q, 0) + d
    last = t
per_q = {}
for r in rows:
    per_q[r["Queue_Id"]] = per_q.get(r["Queue_Id"], 0) + int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("busy time per queue (ms):", {q: round(v / 1e6, 2) for q, v in per_q.items()})
print(f"time with >= 2 queues running: {both / 1e6:.2f} ms; with exactly one: {one / 1e6:.2f} ms")
