"""Torch-native (non-fr::) kernels in a rocprofv3 --stats CSV: calls and average duration."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True))[-1]
rows = list(csv.reader(open(f)))
tot = {r[0]: (int(r[1]), float(r[3]) / 1000) for r in rows[1:]}
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1
for n, (c, a) in sorted(tot.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
    if 'fr::' in n:
        continue
    print(f"{c / steps:7.2f}/step {a:7.2f} us  {n[:150]}")
