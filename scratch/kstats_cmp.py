"""Average duration of the quantize / selection kernels in a rocprofv3 kernel_stats.csv (argv: files)."""
import csv, sys
for path in sys.argv[1:]:
    print("==", path)
    for r in csv.DictReader(open(path)):
        if "k_quant" in r["Name"] or "k_sel" in r["Name"]:
            print(f"  {r['Name'][:62]:62s} calls {r['Calls']:>4s}  avg {float(r['AverageNs']) / 1e3:8.1f} us")
