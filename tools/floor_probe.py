import torch, time
x = torch.zeros(64, device='cuda')
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for K in (100, 400):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): x.add_(1)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(K): x.add_(1)
    print("graph of %d dependent tiny kernels: %.2f us per kernel" % (K, timeit(g.replay) * 1e3 / K))
big = torch.zeros(4 << 20, device='cuda')   # 16 MB
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(100): big.add_(1)
t = timeit(g.replay) * 1e3 / 100
print("graph of 100 x 16 MB add_: %.2f us per kernel = %.2f TB/s" % (t, 2 * 16.8e6 / t / 1e6))

# the library's own small kernels, same treatment
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_amd._lib import lib, ptr, stream, check
c = 128
mm, mv = torch.zeros(c, device='cuda'), torch.ones(c, device='cuda')
sm, sr = torch.zeros(c, device='cuda'), torch.ones(c, device='cuda')
y = torch.zeros(64, device='cuda')
tests = {
    "gan_bn_moving_update c=128": lambda: check(lib.gan_bn_moving_update(ptr(mm), ptr(mv), ptr(sm), ptr(sr), 4096, c, 1e-5, 0.9, 0, stream())),
    "gan_act_fwd 64 elements": lambda: check(lib.gan_act_fwd(ptr(x), ptr(y), 64, 1, 0.2, 1.0, 0.0, stream())),
    "gan_act_fwd 4M elements (16 MB)": lambda: check(lib.gan_act_fwd(ptr(big), ptr(big2), big.numel(), 1, 0.2, 1.0, 0.0, stream())),
}
big2 = torch.zeros_like(big)
for name, fn in tests.items():
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(100): fn()
    print("%s: %.2f us per kernel" % (name, timeit(g.replay) * 1e3 / 100))
