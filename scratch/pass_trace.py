"""Kernel by kernel through one captured pass: python scratch/pass_trace.py <rocprofv3 db> <anchor kernel substring> [nth-from-last]
(the window between two consecutive launches of the anchor kernel, late in the run: graph replays)."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name,start,end from kernels order by start"))
def short(n):
    n = re.sub(r'void ', '', n)
    m = re.match(r'(fr::\w+)(<[^(]*>)?', n)
    if m:
        return m.group(1) + (m.group(2) or '')
    m = re.search(r'(\w+Functor\w*<\w+>|CatArrayBatchedCopy|neg_kernel|copyBuffer)', n)
    return ('torch:' + m.group(1)) if m else n[:60]
names = [short(r[0]) for r in rows]
idx = [i for i, n in enumerate(names) if sys.argv[2] in n]
k = int(sys.argv[3]) if len(sys.argv) > 3 else 300
a, b = idx[-k], idx[-k + 1]
t0 = rows[a][1]
for i in range(a, b + 1):
    n, s, e = rows[i]
    print(f"{(s - t0) / 1000:8.1f} {(e - s) / 1000:6.1f}  {names[i]}")
