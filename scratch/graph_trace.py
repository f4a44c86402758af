"""Wave timeline of ONE step kernel in the middle of bench.py's hipGraph replay (trace build, FAIRREC_HIP_LIB set)."""
import ctypes, io, os, sys, contextlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "recbole-fairrec_amd")]
import bench
from fairrec import _C
lib = ctypes.CDLL(_C.LIB_PATH)
target = int(os.environ.get("TRACE_STEP", "250"))
assert lib.fr_debug_set_trace_step(target) == 0
sys.argv = ["bench.py", "--no-cpu-baseline", "--graph-only"] + sys.argv[1:]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
n = 16384
both = np.zeros((2 * n, 4), dtype=np.uint64)
assert lib.fr_debug_step_trace(both.ctypes.data_as(ctypes.c_void_p), n) == 0
b, phs = both[:n], both[n:]
keep = b[:, 1] > 0
b, phs = b[keep], phs[keep]
if os.environ.get("TRACE_OUT"):
    np.savez_compressed(os.environ["TRACE_OUT"], buf=b, phs=phs, wq=np.nonzero(keep)[0])
t0 = b[:, 0].min()
st, en, role = (b[:, 0] - t0) / 100.0, (b[:, 1] - t0) / 100.0, b[:, 2]
print("step", target, "waves", len(b), "span %.2f us" % en.max())
for rl, name in ((0, "loss"), (1, "sweeper"), (2, "interaction"), (3, "stage")):
    m = role == rl
    if m.any():
        print(f"{name:12s} n={m.sum():5d} start med {np.median(st[m]):6.2f} max {st[m].max():6.2f} | end med {np.median(en[m]):6.2f} p90 {np.percentile(en[m], 90):6.2f} max {en[m].max():6.2f}")
hw = b[:, 3]
simd = ((hw >> 4) & 3) | (((hw >> 8) & 0xf) << 2) | (((hw >> 13) & 7) << 7) | (((hw >> 32) & 0xf) << 10)
ids, inv = np.unique(simd, return_inverse=True)
last = np.zeros(len(ids)); np.maximum.at(last, inv, en)
print("SIMD last end p10 %.1f med %.1f p90 %.1f max %.1f" % tuple(np.percentile(last, [10, 50, 90, 100])))
cyc = (phs[:, 0] & np.uint64((1 << 40) - 1)).astype(np.float64)
dur = (b[:, 1] - b[:, 0]).astype(np.float64) * 10.0
ok = dur > 2000
print("shader clock: median %.3f GHz" % np.median(cyc[ok] / dur[ok]))
import json
d = json.loads([l for l in buf.getvalue().splitlines() if l.startswith("{")][-1])
print("bench: %.2f us/step, kernel %s" % (d["ms_per_step"] * 1e3, d["roofline"]["kernel_us"]))
