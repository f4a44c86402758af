"""Per-kernel durations inside the hipGraph replay window of a rocprofv3 --kernel-trace CSV of bench.py."""
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = []
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void fr::", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
g = [k for k, r in enumerate(rows) if r[2].startswith("focf_step_kernel") or r[2].startswith("focf_gather")]
# densest window of K consecutive gather launches
best = min(range(len(g) - K + 1), key=lambda a: rows[g[a + K - 1]][0] - rows[g[a]][0])
t0, t1 = rows[g[best]][0], rows[g[best + K - 1]][1]
print("window: %d step (gather) launches, %.2f us per step" % (K, (rows[g[best + K - 1]][0] - t0) / (K - 1) / 1e3))
acc = collections.defaultdict(list)
for s, e, n in rows:
    if t0 <= s <= t1:
        acc[n].append((e - s) / 1e3)
for n, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print("%-32s n=%4d avg %7.2f us  min %7.2f  max %7.2f" % (n, len(v), sum(v) / len(v), min(v), max(v)))
# timeline of 2 steps in the middle
mid = g[best + K // 2]
base = rows[mid][0]
for s, e, n in rows[mid - 1: mid + 9]:
    print("  %8.2f -> %8.2f  %s" % ((s - base) / 1e3, (e - base) / 1e3, n))
