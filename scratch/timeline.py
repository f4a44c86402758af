"""Timeline of the step kernels and side kernels inside bench.py's hipGraph replay from a rocprofv3 --kernel-trace CSV."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void fr::", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
rows.sort()
st = [k for k, r in enumerate(rows) if r[2].startswith("focf_step_kernel")]
lo, hi = int(sys.argv[2]), int(sys.argv[3])
a, b = st[lo], st[hi]
base = rows[a][0]
prev_end, tot_gap, n = None, 0.0, 0
for s, e, nm in rows[a:b + 1]:
    tag = "STEP" if nm.startswith("focf_step") else nm[:20]
    gap = ""
    if nm.startswith("focf_step"):
        if prev_end is not None:
            g = (s - prev_end) / 1e3
            tot_gap += g
            gap = " gap %.1f" % g if g > 0.05 else ""
        prev_end = e
        n += 1
    print("%8.1f -> %8.1f  %-20s dur %5.1f%s" % ((s - base) / 1e3, (e - base) / 1e3, tag, (e - s) / 1e3, gap))
print("steps", n, "per step %.2f us, gaps %.2f us per step" % ((rows[b][1] - base) / 1e3 / n, tot_gap / n))
