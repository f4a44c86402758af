"""The launches of ONE training step inside a rocprofv3 --kernel-trace CSV: python profiles/step_window.py <dir> <anchor> [K]
`anchor` = a substring of the kernel that opens a step (one launch per step).  Prints the densest window of K consecutive
steps (wall per step, kernel time and launches per step) and the timeline of one step from its middle."""
import collections
import csv
import glob
import sys

f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
anchor = sys.argv[2]
K = int(sys.argv[3]) if len(sys.argv) > 3 else 16
rows = []
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("fr::", "")
    if "<" in name:
        name = name.split("<")[0] + "<>"
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
g = [k for k, r in enumerate(rows) if anchor in r[2]]
best = min(range(len(g) - K), key=lambda a: rows[g[a + K]][0] - rows[g[a]][0])
t0, t1 = rows[g[best]][0], rows[g[best + K]][0]
inside = [r for r in rows if t0 <= r[0] < t1]
print("window: %d steps, %.2f us wall per step, %.2f us of kernel time per step, %.1f launches per step"
      % (K, (t1 - t0) / K / 1e3, sum(e - s for s, e, _ in inside) / K / 1e3, len(inside) / K))
acc = collections.defaultdict(list)
for s, e, n in inside:
    acc[n].append((e - s) / 1e3)
for n, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print("%-44s n/step %5.2f  avg %7.2f us  per step %7.2f us" % (n[:44], len(v) / K, sum(v) / len(v), sum(v) / K))
mid = g[best + K // 2]
nxt = g[best + K // 2 + 1]
base = rows[mid][0]
print("one step (start -> end, us; gap to the previous launch's end):")
prev = None
for s, e, n in rows[mid:nxt + 1]:
    print("  %8.2f -> %8.2f  (%5.2f)  %s" % ((s - base) / 1e3, (e - base) / 1e3, 0.0 if prev is None else (s - prev) / 1e3, n[:60]))
    prev = e
