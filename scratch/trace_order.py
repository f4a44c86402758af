"""From a graph_trace npz: duration of the sweeper waves against their place in the start order."""
import sys, numpy as np
z = np.load(sys.argv[1])
b, wq = z["buf"], z["wq"]
t0 = b[:, 0].min()
st, en, role = (b[:, 0] - t0) / 100.0, (b[:, 1] - t0) / 100.0, b[:, 2]
for rl, name in ((2, "interaction"), (1, "sweeper")):
    m = role == rl
    o = np.argsort(wq[m])
    d = (en[m] - st[m])[o]
    s = st[m][o]
    n = len(d)
    print(name, "waves", n)
    for k in range(8):
        sl = slice(k * n // 8, (k + 1) * n // 8)
        print(f"  octile {k}: start med {np.median(s[sl]):6.2f}  dur med {np.median(d[sl]):6.2f} p90 {np.percentile(d[sl], 90):6.2f} max {d[sl].max():6.2f}")
