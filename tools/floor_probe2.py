"""Where do the ~4.5 us of a tiny kernel inside the training graph come from?  Wall time of graphs that
interleave tiny kernels with (a) a large streaming kernel of small code, (b) a contraction kernel."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_amd import _lib
from gan_amd._lib import lib, ptr, stream, check

def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

def graph_us(body, reps=100):
    body(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): body()
    return timeit(g.replay) / reps

c = 128
mm, mv, sm, sr = [torch.ones(c, device='cuda') for _ in range(4)]
x64, y64 = torch.zeros(64, device='cuda'), torch.zeros(64, device='cuda')
big, big2 = torch.zeros(4 << 20, device='cuda'), torch.zeros(4 << 20, device='cuda')
tiny1 = lambda: check(lib.gan_bn_moving_update(ptr(mm), ptr(mv), ptr(sm), ptr(sr), 4096, c, 1e-5, 0.9, 0, stream()))
tiny2 = lambda: check(lib.gan_act_fwd(ptr(x64), ptr(y64), 64, 1, 0.2, 1.0, 0.0, stream()))
tiny3 = lambda: check(lib.gan_add_fwd(ptr(x64), ptr(y64), ptr(y64), 64, stream()))
stream_k = lambda: check(lib.gan_act_fwd(ptr(big), ptr(big2), big.numel(), 1, 0.2, 1.0, 0.0, stream()))
g = _lib.conv_geom(64, 32, 32, 64, 16, 16, 128, 5, 5, 2, 2, 1, 1)
xin = torch.randn(64, 32, 32, 64, device='cuda'); w = torch.randn(5, 5, 64, 128, device='cuda') * 0.02
b = torch.zeros(128, device='cuda'); yout = torch.empty(64, 16, 16, 128, device='cuda')
nbytes = int(lib.gan_conv_workspace_bytes(ctypes.byref(g)))
wp, wn = _lib.workspace(nbytes)
conv = lambda: check(lib.gan_conv2d_fwd(ctypes.byref(g), ptr(xin), ptr(w), ptr(b), ptr(yout), wp, wn, stream()))

base_s = graph_us(stream_k)
print("streaming kernel alone                : %6.2f us" % base_s)
print("streaming + tiny1                     : +%5.2f us for the tiny one" % (graph_us(lambda: (stream_k(), tiny1())) - base_s))
print("streaming + tiny1 + tiny2 + tiny3     : +%5.2f us per tiny one" % ((graph_us(lambda: (stream_k(), tiny1(), tiny2(), tiny3())) - base_s) / 3))
base_c = graph_us(conv)
print("contraction (+ slab reduce) alone     : %6.2f us" % base_c)
print("contraction + tiny1                   : +%5.2f us for the tiny one" % (graph_us(lambda: (conv(), tiny1())) - base_c))
print("contraction + tiny1 + tiny2 + tiny3   : +%5.2f us per tiny one" % ((graph_us(lambda: (conv(), tiny1(), tiny2(), tiny3())) - base_c) / 3))
print("tiny1, tiny2, tiny3 alone             : %6.2f us per tiny one" % (graph_us(lambda: (tiny1(), tiny2(), tiny3())) / 3))

# cold data: every tiny launch works on arrays nothing has touched since a 256 MB sweep evicted the caches
sets = [[torch.ones(c, device='cuda') for _ in range(4)] for _ in range(50)]
huge, huge2 = torch.zeros(64 << 20, device='cuda'), torch.zeros(64 << 20, device='cuda')     # 256 MB each
sweep = lambda: check(lib.gan_act_fwd(ptr(huge), ptr(huge2), huge.numel(), 1, 0.2, 1.0, 0.0, stream()))
def cold_body():
    sweep()
    for s in sets:
        check(lib.gan_bn_moving_update(ptr(s[0]), ptr(s[1]), ptr(s[2]), ptr(s[3]), 4096, c, 1e-5, 0.9, 0, stream()))
def hot_body():
    sweep()
    for s in sets:
        check(lib.gan_bn_moving_update(ptr(mm), ptr(mv), ptr(sm), ptr(sr), 4096, c, 1e-5, 0.9, 0, stream()))
t_sweep = graph_us(sweep, reps=5)
print("tiny kernel on 50 cold array sets     : %6.2f us per launch" % ((graph_us(cold_body, reps=5) - t_sweep) / 50))
print("tiny kernel on one hot array set      : %6.2f us per launch" % ((graph_us(hot_body, reps=5) - t_sweep) / 50))
