"""Timeline of the last step in a rocprofv3 kernel trace CSV: kernel, stream-ish (queue), start offset, duration."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 80
rows = rows[-nlast:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  q{r.get('Queue_Id', '?'):>3}  {r['Kernel_Name'][:70]}")
