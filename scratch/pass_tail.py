"""The last <ms> milliseconds of a rocprofv3 kernel trace (sqlite): kernels of at least <min_us> one by one, the rest summed.
python scratch/pass_tail.py <db> <ms> [min_us]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name,start,end from kernels order by start"))
span, min_us = float(sys.argv[2]) * 1e6, float(sys.argv[3]) if len(sys.argv) > 3 else 100.0
t1 = rows[-1][2]
def short(n):
    n = re.sub(r'void ', '', n)
    m = re.match(r'(fr::\w+)(<[^(]*>)?', n)
    if m:
        return m.group(1) + (m.group(2) or '')
    m = re.search(r'(\w+Functor\w*<\w+>|CatArrayBatchedCopy|neg_kernel|copyBuffer|reduce_kernel|index\w*|nonzero\w*)', n)
    return ('torch:' + m.group(1)) if m else n[:70]
small, small_t, big_t = 0, 0.0, 0.0
for n, s, e in rows:
    if s < t1 - span:
        continue
    d = (e - s) / 1000
    if d >= min_us:
        big_t += d
        print(f"{(s - (t1 - span)) / 1000:9.1f} {d:9.1f}  {short(n)}")
    else:
        small += 1
        small_t += d
print(f"kernels >= {min_us} us: {big_t / 1000:.2f} ms; {small} smaller ones: {small_t / 1000:.2f} ms")
