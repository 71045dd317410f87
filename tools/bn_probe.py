"""Time gan_bn_fwd_train / gan_bn_bwd on the small-activation shapes, 100 calls per HIP graph."""
import os, sys, ctypes
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_amd import _lib
from gan_amd._lib import lib, ptr, stream, check

def graph_us(body, reps=100):
    body(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): body()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 / reps * 1e3

for groups, rows, c in [(1, 1024, 512), (1, 4096, 256), (2, 1024, 512), (2, 4096, 256), (1, 64, 1024), (1, 16384, 128)]:
    x = torch.randn(groups * rows, c, device='cuda'); y = torch.empty_like(x); dy = torch.randn_like(x); dx = torch.empty_like(x)
    ga, be = torch.ones(c, device='cuda'), torch.zeros(c, device='cuda')
    mm, mv = torch.zeros(c, device='cuda'), torch.ones(c, device='cuda')
    sm, sr = torch.empty(groups * c, device='cuda'), torch.empty(groups * c, device='cuda')
    dg, db = torch.empty(c, device='cuda'), torch.empty(c, device='cuda')
    wp, wn = _lib.workspace(int(lib.gan_bn_workspace_bytes(groups, rows, c)))
    fwd = lambda: check(lib.gan_bn_fwd_train(ptr(x), ptr(ga), ptr(be), ptr(mm), ptr(mv), None, ptr(y), ptr(sm), ptr(sr), groups, rows, c,
                                             1e-5, 0.9, 0, 2, 0.2, wp, wn, stream()))
    bwd = lambda: check(lib.gan_bn_bwd(ptr(x), ptr(ga), ptr(be), ptr(sm), ptr(sr), ptr(dy), ptr(dx), ptr(dg), ptr(db), 0, groups, rows, c,
                                       2, 0.2, wp, wn, stream()))
    print("groups %d rows %5d c %4d (%5.1f MB): fwd %6.2f us  bwd %6.2f us" % (groups, rows, c, x.numel() * 4 / 1e6, graph_us(fwd), graph_us(bwd)))
