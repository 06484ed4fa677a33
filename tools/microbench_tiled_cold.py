"""conv_tiled on ONE buffer set (what tools/microbench_tiled.py times: x / residual / y = 252 MB, about the size of the 256-MB
Infinity Cache) against the same launch cycling through K buffer sets (every operand cold, as in the step, where each launch reads
tensors other kernels wrote hundreds of microseconds and > 1 GB of traffic earlier).  python tools/microbench_tiled_cold.py [K]"""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rt_pose_amd.backend import HipBackend
from rt_pose_amd.graph import Geom, View
be = HipBackend('cuda:0')
K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n, d, h, w, c = 8, 16, 64, 160, 32
g = Geom(n, d, h, w, d, h, w, c, c, 3, 1, 1)
def mk(shape): return torch.randn(shape, device='cuda').to(torch.bfloat16)
def view(): return View(mk((n, d, h, w, c)), n, d, h, w, c, 0, c)
wf = mk((n, 27, c, c)) * 0.05
bt = torch.randn(n, 64, c, device='cuda')
sets = [(view(), view(), view()) for _ in range(K)]
fns = [be.conv(x, wf, True, bt, res, y, g, True, False, False) for x, res, y in sets]
S = be.wgrad_nsplit(g)
gps = [torch.zeros(n, S, 27, c, c, device='cuda') for _ in range(K)]
wgs = [be.wgrad(y, x, g, S, gp) for (x, res, y), gp in zip(sets, gps)]
def t(fl, it=304):
    s = be.stream()
    for i in range(304): fl[i % len(fl)](s)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(it): fl[i % len(fl)](s)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e6
print('conv_tiled: one buffer set %.1f us | %d sets round-robin %.1f us' % (t(fns[:1]), K, t(fns)))
print('wgrad_tiled: one buffer set %.1f us | %d sets round-robin %.1f us' % (t(wgs[:1]), K, t(wgs)))
