"""Top kernels by total time in a rocprofv3 --kernel-trace CSV dir (names cut at 80 chars), restricted to the last `frac` of the trace."""
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
t0, t1 = rows[0][0], rows[-1][1]
cut = t1 - (t1 - t0) * frac
acc = collections.defaultdict(lambda: [0.0, 0])
for s, e, n in rows:
    if s >= cut:
        k = n.split("(")[0][:80]
        acc[k][0] += (e - s) / 1e6
        acc[k][1] += 1
tot = sum(v[0] for v in acc.values())
print("window %.1f ms, kernel time %.1f ms" % ((t1 - cut) / 1e6, tot))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][0])[:22]:
    print("%9.2f ms  n=%5d  %s" % (v[0], v[1], k))
