"""Per-kernel summary of a rocprofv3 --kernel-trace sqlite db: python scratch/kstats.py db n_steps [top]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); n = float(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 45
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
sym = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = db.execute(f"select s.kernel_name, count(*), avg(k.end-k.start), sum(k.end-k.start) from {kd} k join {sym} s on k.kernel_id=s.id group by s.kernel_name order by 4 desc").fetchall()
print("sum of kernel time per step, us:", sum(r[3] for r in rows) / n / 1e3, " launches per step:", sum(r[1] for r in rows) / n)
for r in rows[:top]:
    print(f"{r[0][:78]:78s} n/step={r[1]/n:6.2f} avg={r[2]/1e3:8.2f}us per-step={r[3]/n/1e3:8.2f}us")
