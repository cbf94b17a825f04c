"""Per kernel name: mean of each counter over the dispatches of a rocprofv3 --pmc counter_collection.csv."""
import sys, csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if "k_stats" not in k and "k_quant" not in k and "k_np" not in k: continue
        acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()): print(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}  first={v[0]:.4g} last={v[-1]:.4g}")
