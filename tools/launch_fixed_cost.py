"""Fixed cost of one launch of the persistent LDS-tiled kernels vs their per-brick cost: the 32 -> 32 full-resolution conv and weight
gradient at n = 4 / 8 / 16 / 32 samples of [16, 64, 160] (10 / 20 / 40 / 80 bricks per workgroup on 256 workgroups), launched back to
back on one stream; least-squares line through (bricks per workgroup, us per launch).  tools/launch_fixed_cost.py"""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rt_pose_amd.backend import HipBackend
from rt_pose_amd.graph import Geom, View
be = HipBackend('cuda:0')
d, h, w, c = 16, 64, 160, 32
def mk(shape, dt=torch.bfloat16): return torch.randn(shape, device='cuda').to(dt)
def t(f, it=200):
    s = be.stream()
    for _ in range(200): f(s)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): f(s)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e6
rows = []
for n in (4, 8, 16, 32):
    g = Geom(n, d, h, w, d, h, w, c, c, 3, 1, 1)
    x = View(mk((n, d, h, w, c)), n, d, h, w, c, 0, c)
    y = View(mk((n, d, h, w, c)), n, d, h, w, c, 0, c)
    res = View(mk((n, d, h, w, c)), n, d, h, w, c, 0, c)
    wf = mk((n, 27, c, c)) * 0.05
    bt = torch.randn(n, 64, c, device='cuda')
    S = be.wgrad_nsplit(g)
    gp = torch.zeros(n, S, 27, c, c, device='cuda')
    bricks = n * 640 / 256
    rows.append((bricks, t(be.conv(x, wf, True, bt, res, y, g, True, False, False)), t(be.wgrad(y, x, g, S, gp))))
    print("n %2d  bricks per workgroup %5.1f  conv_tiled %7.1f us  wgrad_tiled %7.1f us" % (n, *rows[-1]), flush=True)
    del x, y, res, gp
    torch.cuda.empty_cache()
b = np.array([r[0] for r in rows])
for k, name in ((1, "conv_tiled"), (2, "wgrad_tiled")):
    v = np.array([r[k] for r in rows])
    slope, icpt = np.polyfit(b, v, 1)
    print("%s: %.2f us fixed per launch + %.3f us per brick of a workgroup (fit through %s)" % (name, icpt, slope, [round(float(q), 1) for q in v]))
