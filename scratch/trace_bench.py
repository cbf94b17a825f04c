"""Summarise a rocprofv3 kernel trace (rocpd sqlite) of bench.py: time per kernel name inside the timed steps."""
import glob
import sqlite3
import sys

db = sorted(glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True))[-1]
con = sqlite3.connect(db)
rows = con.execute("select name, start, end from kernels order by start").fetchall()
tot = {}
for name, s, e in rows:
    k = name.split("(")[0][:90]
    t = tot.setdefault(k, [0, 0.0])
    t[0] += 1
    t[1] += (e - s) / 1e6
span = (rows[-1][2] - rows[0][1]) / 1e6
print(f"{len(rows)} launches, span {span:.1f} ms, sum of kernel time {sum(v[1] for v in tot.values()):.1f} ms")
for k, (n, ms) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print(f"{ms:10.2f} ms {n:7d}  {ms / n * 1e3:9.1f} us  {k}")
