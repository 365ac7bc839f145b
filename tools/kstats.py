"""Aggregate a rocprofv3 rocpd database (kernel trace) into per-kernel totals.  python tools/kstats.py DB [OUT.csv] [nsteps]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); c = db.cursor()
rows = c.execute("select name, count(*), sum(end-start), avg(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
if len(sys.argv) > 2 and sys.argv[2] != "-":
    with open(sys.argv[2], "w") as f:
        f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage\n")
        for r in rows: f.write(f'"{r[0]}",{r[1]},{r[2]},{r[3]:.1f},{100*r[2]/tot:.2f}\n')
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1
print(f"total {tot/1e6:.2f} ms over {n} steps = {tot/1e6/n:.2f} ms/step")
for r in rows[:32]:
    nm = re.sub(r'\(anonymous namespace\)::', '', r[0])[:64]
    print(f"{nm:64s} {r[1]:6d} {r[2]/1e6:9.2f} {r[3]/1e3:9.1f} {100*r[2]/tot:5.1f}")
