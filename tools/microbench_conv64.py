"""64 -> 64-channel 3x3x3 stride-1 conv at the native shape (B = 8, 16 x 64 x 160) and at level 1 (8 x 32 x 80): forward with class
bias + residual + ReLU + statistics, data gradient with P / Q statistics.  RTP_CONV64=0 in the environment: the four-slice route."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rt_pose_amd.backend import HipBackend
from rt_pose_amd.graph import Geom, View
be = HipBackend('cuda:0')
def mk(shape, dt=torch.bfloat16): return torch.randn(shape, device='cuda').to(dt)
def t(f, it=100):
    s = be.stream()
    for _ in range(100): f(s)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): f(s)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e6
n, c = 8, 64
for (d, h, w) in (((16, 64, 160),) if 'full' in sys.argv else ((16, 64, 160), (8, 32, 80), (4, 16, 40), (2, 8, 20))):   # full: the native shape only (PMC passes)
    g = Geom(n, d, h, w, d, h, w, c, c, 3, 1, 1)
    x = View(mk((n, d, h, w, c)), n, d, h, w, c, 0, c)
    y = View(mk((n, d, h, w, c)), n, d, h, w, c, 0, c)
    res = View(mk((n, d, h, w, c)), n, d, h, w, c, 0, c)
    wf = mk((n, 27, c, c)) * 0.05
    wd = mk((27, c, c)) * 0.05
    bt = torch.randn(n, 64, c, device='cuda')
    ws = torch.zeros(n * d * h * w * 32, device='cuda')
    S = be.conv_stats_nsplit(x, g, False, ws=True)
    st = torch.zeros(n, S, c, 2, device='cuda')
    Sb = be.conv_stats_nsplit(y, g, True, ws=True)
    pq = torch.zeros(n, Sb, c, 2, device='cuda')
    fwd = be.conv(x, wf, True, bt, res, y, g, True, False, False, (None, st), ws=ws)
    bwd = be.conv(y, wd, False, None, None, x, g, False, True, False, (res, pq), ws=ws)
    gf = 2.0 * n * d * h * w * c * c * 27 / 1e9
    a, b = t(fwd), t(bwd)
    print('RTP_CONV64=%s %dx%dx%d: forward %.1f us (%.0f TFLOP/s)  data gradient %.1f us (%.0f TFLOP/s)  [%d workgroups per sample]'
          % (os.environ.get('RTP_CONV64', '1'), d, h, w, a, gf / a * 1e3, b, gf / b * 1e3, S))
