"""Per-variant timing of the LDS-tiled conv (forward / data-gradient, with and without the fused statistics) at the
full-resolution and level-1 shapes of the hr3d step.  Sustained back-to-back launches (clocks ramp under load)."""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rt_pose_amd.backend import HipBackend
from rt_pose_amd.graph import Geom, View
be = HipBackend('cuda:0')
c, n = 32, 8
def mk(shape, dt=torch.bfloat16): return torch.randn(shape, device='cuda').to(dt)
def t(f, it=300):
    s = be.stream()
    for _ in range(200): f(s)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): f(s)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e6
has_stats = hasattr(be, "conv_stats_nsplit")
for (d, h, w) in ((16, 64, 160), (8, 32, 80)):
    g = Geom(n, d, h, w, d, h, w, c, c, 3, 1, 1)
    x = View(mk((n, d, h, w, c)), n, d, h, w, c, 0, c)
    y = View(mk((n, d, h, w, c)), n, d, h, w, c, 0, c)
    res = View(mk((n, d, h, w, c)), n, d, h, w, c, 0, c)
    wf = mk((n, 27, c, c)) * 0.05
    wd = mk((27, c, c)) * 0.05
    bt = torch.randn(n, 64, c, device='cuda')
    gf = 2 * n * d * h * w * c * c * 27 / 1e9
    rows = [("fwd btab+res+relu", lambda st: be.conv(x, wf, True, bt, res, y, g, True, False, False, *st)),
            ("fwd btab+relu", lambda st: be.conv(x, wf, True, bt, None, y, g, True, False, False, *st)),
            ("dgrad", lambda st: be.conv(x, wd, False, None, None, y, g, False, True, False, *st))]
    for name, mkf in rows:
        line = "%-10s %-18s plain %6.1f us (%5.0f TF/s)" % ("%dx%dx%d" % (d, h, w), name, 0, 0)
        tp = t(mkf(()))
        line = "%-10s %-18s plain %6.1f us (%5.0f TF/s)" % ("%dx%dx%d" % (d, h, w), name, tp, gf / tp * 1e-3 * 1e3)
        if has_stats:
            S = be.conv_stats_nsplit(x, g, name == "dgrad")
            so = torch.zeros(n, S, c, 2, device='cuda')
            ts = t(mkf(((res if name == "dgrad" else None, so),)))
            line += "   +stats %6.1f us (%5.0f TF/s)" % (ts, gf / ts * 1e-3 * 1e3)
        print(line)
