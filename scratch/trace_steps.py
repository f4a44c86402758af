"""Timeline of the kernels of a few hipGraph-replayed steps of any bench workload: rocprofv3 --kernel-trace CSV dir, the
kernel-name prefix that starts a step, how many steps to print from the middle of the densest window."""
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
anchor = sys.argv[2]
nshow = int(sys.argv[3]) if len(sys.argv) > 3 else 2
rows = []
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0].replace("void fr::", "").replace("fr::", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
g = [k for k, r in enumerate(rows) if r[2].startswith(anchor)]
K = min(100, len(g) - 1)
best = min(range(len(g) - K), key=lambda a: rows[g[a + K]][0] - rows[g[a]][0])
print("window of %d steps: %.2f us per step" % (K, (rows[g[best + K]][0] - rows[g[best]][0]) / K / 1e3))
mid = best + K // 2
base = rows[g[mid]][0]
prev_end = base
for s, e, n in rows[g[mid]: g[mid + nshow]]:
    print("  %8.2f -> %8.2f  (%6.2f us, gap %5.2f)  %s" % ((s - base) / 1e3, (e - base) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, n[:90]))
    prev_end = max(prev_end, e)
