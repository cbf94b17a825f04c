"""Per-kernel counter averages from a rocprofv3 --pmc run (rocpd sqlite)."""
import glob, sqlite3, sys
db = sorted(glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True))[-1]
con = sqlite3.connect(db)
rows = con.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
for k, c, v, n in rows:
    if sys.argv[2] in k:
        print(f"{k.split('(')[0][:50]:50s} {c:28s} {v:16.1f}  (n={n})")
