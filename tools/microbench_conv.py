"""Time one conv / data-gradient / weight-gradient geometry (for rocprofv3 --pmc passes too).
usage: microbench_conv.py D H W CI CO KS STRIDE [n=8] [iters=50]"""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rt_pose_amd.backend import HipBackend
from rt_pose_amd.graph import Geom, View, wgrad_split
a = [int(v) for v in sys.argv[1:]]
d, h, w, ci, co, ks, st = a[:7]
n = a[7] if len(a) > 7 else 8
it = a[8] if len(a) > 8 else 50
be = HipBackend('cuda:0')
pad = ks // 2
do, ho, wo = [(s + 2 * pad - ks) // st + 1 for s in (d, h, w)]
g = Geom(n, d, h, w, do, ho, wo, ci, co, ks, st, pad)
def mk(shape, dt=torch.bfloat16): return torch.randn(shape, device='cuda').to(dt)
x = View(mk((n, d, h, w, ci)), n, d, h, w, ci, 0, ci)
y = View(mk((n, do, ho, wo, co)), n, do, ho, wo, co, 0, co)
dx = View(mk((n, d, h, w, ci)), n, d, h, w, ci, 0, ci)
wf = mk((n, ks ** 3, co, ci)) * 0.05
wd = mk((ks ** 3, ci, co)) * 0.05
bt = torch.randn(n, 64, co, device='cuda')
S = be.wgrad_nsplit(g) or wgrad_split(do * ho * wo)
gp = torch.zeros(n, S, ks ** 3, co, ci, device='cuda')
fs = {"conv": be.conv(x, wf, True, bt, None, y, g, True, False, False),
      "dgrad": be.conv(y, wd, False, None, None, dx, g, False, True, False),
      "wgrad": be.wgrad(y, x, g, S, gp)}
flops = 2.0 * n * do * ho * wo * co * ci * ks ** 3
for name, f in fs.items():
    s = be.stream()
    for _ in range(20): f(s)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): f(s)
    torch.cuda.synchronize(); us = (time.perf_counter() - t0) / it * 1e6
    print("%s %s: %.1f us  %.1f TFLOP/s" % (name, a[:7], us, flops / us / 1e6))
