"""Per-kernel summary of a rocprofv3 --kernel-trace sqlite db.
python scratch/kstats.py db n_steps [top]            whole trace divided by n_steps
python scratch/kstats.py db @kernel_substr [top]     only the last 60 % of the dispatches; steps = launches of that kernel"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 45
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
sym = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = db.execute(f"select s.kernel_name, k.start, k.end from {kd} k join {sym} s on k.kernel_id=s.id order by k.start").fetchall()
if sys.argv[2].startswith("@"):
    rows = rows[int(len(rows) * 0.4):]
    marks = [i for i, r in enumerate(rows) if sys.argv[2][1:] in r[0]]
    rows = rows[marks[0]:marks[-1]]
    n = float(len(marks) - 1)
    print("window:", n, "steps,", (rows[-1][1] - rows[0][1]) / n / 1e3, "us wall per step")
else:
    n = float(sys.argv[2])
agg = {}
for name, s, e in rows:
    a = agg.setdefault(name, [0, 0]); a[0] += 1; a[1] += e - s
print("sum of kernel time per step, us:", sum(a[1] for a in agg.values()) / n / 1e3, " launches per step:", sum(a[0] for a in agg.values()) / n)
for name, a in sorted(agg.items(), key=lambda t: -t[1][1])[:top]:
    print(f"{name[:78]:78s} n/step={a[0]/n:6.2f} avg={a[1]/a[0]/1e3:8.2f}us per-step={a[1]/n/1e3:8.2f}us")
