"""Prints the kernel timeline of the last step in a rocprofv3 --kernel-trace csv: start, duration, queue, name."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marker = sys.argv[2] if len(sys.argv) > 2 else 'k_store_host'
idx = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]['Start_Timestamp'])
for r in rows[a:b + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{(s - t0) / 1000:9.1f} +{(e - s) / 1000:7.1f}  q{r['Queue_Id']} s{r['Stream_Id']} {r['Kernel_Name'][:70]}")
