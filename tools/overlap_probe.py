"""Do a wgrad and a dgrad of the same layer overlap when issued on two streams?  Eager and as a captured graph."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_amd import _lib as L, ops, autotune
autotune.enable(True)
def bufs(n, h, w, cb, cs):
    g = ops._conv_geom(n, h, w, cb, cs, 5, 5, 2, 2, 'SAME')
    autotune.tune_geometry(g)
    return g, torch.randn(n, h, w, cb, device='cuda'), torch.randn(n, g.hs, g.ws, cs, device='cuda'), torch.randn(5, 5, cb, cs, device='cuda') * .05
ws1 = torch.empty(256 << 20, dtype=torch.uint8, device='cuda'); ws2 = torch.empty(256 << 20, dtype=torch.uint8, device='cuda')
side = torch.cuda.Stream()
for (n, h, w, cb, cs) in [(64, 32, 32, 64, 128), (64, 16, 16, 128, 256), (64, 8, 8, 256, 512)]:
    g, big, small, wt = bufs(n, h, w, cb, cs)
    dw = torch.empty_like(wt); dx = torch.empty_like(big)
    gp = ctypes.byref(g)
    def seq():
        st = L.stream()
        L.lib.gan_conv2d_wgrad(gp, L.ptr(big), L.ptr(small), L.ptr(dw), None, 0, ws1.data_ptr(), ws1.numel(), st)
        L.lib.gan_conv2d_dgrad(gp, L.ptr(small), L.ptr(wt), L.ptr(dx), ws1.data_ptr(), ws1.numel(), st)
    def par():
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            L.lib.gan_conv2d_wgrad(gp, L.ptr(big), L.ptr(small), L.ptr(dw), None, 0, ws2.data_ptr(), ws2.numel(), side.cuda_stream)
        L.lib.gan_conv2d_dgrad(gp, L.ptr(small), L.ptr(wt), L.ptr(dx), ws1.data_ptr(), ws1.numel(), main.cuda_stream)
        main.wait_stream(side)
    def timeit(fn, reps=30):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    res = {"eager seq": timeit(seq), "eager 2-stream": timeit(par)}
    for name, fn in (("graph seq", seq), ("graph 2-stream", par)):
        gr = torch.cuda.CUDAGraph()
        cap = torch.cuda.Stream()
        with torch.cuda.stream(cap):
            fn(); torch.cuda.synchronize()
            with torch.cuda.graph(gr, stream=cap):
                for _ in range(10): fn()
        res[name] = timeit(gr.replay, 10) / 10
    print((n, h, w, cb, cs), "  ".join("%s %.1f us" % kv for kv in res.items()), flush=True)
