"""Per-kernel-name durations in a rocprofv3 --kernel-trace CSV dir, for names matching argv[2] (substring)."""
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
pat = sys.argv[2]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if pat in n:
        acc[n.split("(")[0][:70] + " grid=" + r.get("Grid_Size", "?")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print("%-95s n=%4d median %9.1f us  max %9.1f" % (n, len(v), v2[len(v2) // 2], v2[-1]))
