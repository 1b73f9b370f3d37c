"""Summarise a rocprofv3 rocpd database (kernel-trace): per-kernel calls / total / average, optional CSV."""
import csv, sqlite3, sys
db = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 else None
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
if out:
    w = csv.writer(open(out, "w"))
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([r[0], r[1], r[2], round(r[3], 1), round(100 * r[2] / tot, 3), r[4], r[5]])
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"{r[2]/1e6:9.3f} ms {r[1]:6d} calls {r[3]/1e3:9.1f} us avg  {r[0][:100]}")
print(f"total {tot/1e6:.3f} ms")
