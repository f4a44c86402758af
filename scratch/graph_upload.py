"""Does hipGraphUpload take the first launch's upload out of a graph's first replay?  20 small kernels captured, first replay
timed with and without an upload before it."""
import ctypes, time, torch
hip = ctypes.CDLL("libamdhip64.so")
x = torch.zeros(1 << 22, device="cuda")
def build():
    g = torch.cuda.CUDAGraph(keep_graph=True)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
            for _ in range(20):
                x.add_(1.0)
    torch.cuda.current_stream().wait_stream(s)
    g.instantiate()
    return g
for upload in (False, True, False, True):
    g = build()
    torch.cuda.synchronize()
    if upload:
        st = torch.cuda.current_stream().cuda_stream
        rc = hip.hipGraphUpload(ctypes.c_void_p(g.raw_cuda_graph_exec()), ctypes.c_void_p(st))
        torch.cuda.synchronize()
    t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); t1 = time.perf_counter()
    g.replay(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"upload {upload} rc {rc if upload else '-'}: first replay {1e6 * (t1 - t0):.0f} us, second {1e6 * (t2 - t1):.0f} us")
