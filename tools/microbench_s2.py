"""The four full-resolution stride-2 layers' geometry (t1 / s*.f10.0 / s*.f20.0: 32 -> 32, [8,16,64,160] -> [8,8,32,80]), forward conv +
weight gradient + data gradient, on whatever kernels the environment selects (RTP_DISABLE_S2_FWD / RTP_DISABLE_S2_WGRAD = the
generic gather kernels of round 2).  Used unprofiled (prints us per launch) and under rocprofv3 --pmc (tools/pmc_s2.sh)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rt_pose_amd.backend import HipBackend
from rt_pose_amd.graph import Geom, View
be = HipBackend('cuda:0')
n, d, h, w, c = 8, 16, 64, 160, 32
g = Geom(n, d, h, w, d // 2, h // 2, w // 2, c, c, 3, 2, 1)
mk = lambda shape: torch.randn(shape, device='cuda').to(torch.bfloat16)
x = View(mk((n, d, h, w, c)), n, d, h, w, c, 0, c)
y = View(mk((n, d // 2, h // 2, w // 2, c)), n, d // 2, h // 2, w // 2, c, 0, c)
dx = View(mk((n, d, h, w, c)), n, d, h, w, c, 0, c)
wf = mk((n, 27, c, c)) * 0.05
wd = mk((27, c, c)) * 0.05
bt = torch.randn(n, 64, c, device='cuda')
S = be.conv_stats_nsplit(x, g, False)
st = torch.zeros(n, max(S, 1), c, 2, device='cuda')
f_fwd = be.conv(x, wf, True, bt, None, y, g, True, False, False, (None, st) if S else None)
Sw = be.wgrad_nsplit(g) or 64
gp = torch.zeros(n, Sw, 27, c, c, device='cuda')
f_wg = be.wgrad(y, x, g, Sw, gp)
f_dg = be.conv(y, wd, False, None, None, dx, g, False, True, False)
it = 30 if 'pmc' in sys.argv else 300
def t(f):
    s = be.stream()
    for _ in range(it): f(s)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): f(s)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e6
print('stride-2 32->32 [8,16,64,160]: forward %.1f us | weight gradient %.1f us | data gradient %.1f us  (S2_FWD off=%s, S2_WGRAD off=%s)' % (
    t(f_fwd), t(f_wg), t(f_dg), os.environ.get('RTP_DISABLE_S2_FWD', '0'), os.environ.get('RTP_DISABLE_S2_WGRAD', '0')))
