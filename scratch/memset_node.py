"""Is a hipMemsetAsync captured into a hipGraph (a memset NODE) honoured on every replay?  Inside ONE captured stream:
a = fill(7.0); free a; x = empty (takes a's block); hipMemsetAsync(x, 0); y = x + 0.  Every replay must give y == 0.
Variants: the memset issued from the capturing thread or from another thread (torch's autograd engine launches backward
kernels from its own worker thread), small and large buffers."""
import ctypes
import glob
import os
import sys
import threading

import torch

libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
hip = ctypes.CDLL(glob.glob(os.path.join(libdir, "libamdhip64.so*"))[0])
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetAsync.restype = ctypes.c_int


def raw_stream():
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def run(n, other_thread, mode):
    outs = {}

    def body():
        a = torch.full((n,), 7.0, device="cuda")
        del a
        x = torch.empty(n, device="cuda")
        st = raw_stream()
        if other_thread:
            rc = []

            def w():
                torch.cuda.set_device(0)
                rc.append(hip.hipMemsetAsync(x.data_ptr(), 0, n * 4, st))
            t = threading.Thread(target=w)
            t.start()
            t.join()
            assert rc == [0], rc
        else:
            assert hip.hipMemsetAsync(x.data_ptr(), 0, n * 4, st) == 0
        outs["y"] = x + 0
        outs["x_ptr"] = x.data_ptr()

    body()      # eager warm-up
    torch.cuda.synchronize()
    assert float(outs["y"].abs().max()) == 0.0
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode=mode):
        body()
    bad = 0
    for r in range(20):
        g.replay()
        torch.cuda.synchronize()
        m = float(outs["y"].abs().max())
        bad += m != 0.0
    print(f"n={n:>9d} other_thread={other_thread!s:5s} mode={mode:12s}: {bad}/20 replays left y != 0", flush=True)
    return bad


if __name__ == "__main__":
    total = 0
    for mode in ("thread_local", "global"):
        for n in (1120, 70 * 16, 1 << 16, 1 << 22):
            for ot in (False, True):
                try:
                    total += run(n, ot, mode)
                except Exception as e:      # noqa: BLE001
                    print(f"n={n} other_thread={ot} mode={mode}: {type(e).__name__}: {str(e)[:200]}", flush=True)
    print("bad replays in total:", total)
    os._exit(0)
