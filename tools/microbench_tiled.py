import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rt_pose_amd.backend import HipBackend
from rt_pose_amd.graph import Geom, View
be = HipBackend('cuda:0')
n, d, h, w, c = 8, 16, 64, 160, 32
g = Geom(n, d, h, w, d, h, w, c, c, 3, 1, 1)
def mk(shape, dt=torch.bfloat16): return torch.randn(shape, device='cuda').to(dt)
x = View(mk((n, d, h, w, c)), n, d, h, w, c, 0, c)
y = View(mk((n, d, h, w, c)), n, d, h, w, c, 0, c)
res = View(mk((n, d, h, w, c)), n, d, h, w, c, 0, c)
wf = mk((n, 27, c, c)) * 0.05
bt = torch.randn(n, 64, c, device='cuda')
S = be.wgrad_nsplit(g)
gp = torch.zeros(n, S, 27, c, c, device='cuda')
f_conv = be.conv(x, wf, True, bt, res, y, g, True, False, False)
f_wg = be.wgrad(y, x, g, S, gp)
def t(f, it=300):
    s = be.stream()
    for _ in range(300): f(s)   # long warm-up: the clocks ramp only under sustained load
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): f(s)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e6
print('dbg', os.environ.get('RTP_TILED_DBG', '0'), 'conv_tiled %.1f us' % t(f_conv), 'wgrad_tiled %.1f us' % t(f_wg))
if 'fold' in sys.argv:
    Wm = torch.randn(c, c, 3, 3, 3, device='cuda') * 0.05
    gam = torch.rand(c, device='cuda') + 0.5; bet = torch.randn(c, device='cuda') * 0.1
    st = torch.rand(n, 32, c, 2, device='cuda') * 1e4 + 1e5; st[..., 1] *= 2
    mr = torch.zeros(n, 8, 2, device='cuda'); so = torch.zeros(n, 32, c, 2, device='cuda')
    wf2 = mk((n, 27, c, c)); bt2 = torch.zeros(n, 64, c, device='cuda')
    wt_ = torch.zeros(27, c, c, device='cuda'); be.tail([('pack_wt', Wm, c, c, c, 27, wt_)])(be.stream())
    f_fused = be.conv_gn_fused(x, wt_, None, gam, bet, st, 32, 8, 1e-5, c, mr, res, y, g, True, so)
    f_fold = be.fold_fwd(Wm, None, gam, bet, st, 32, 8, 1e-5, g, c, c, wf2, bt2, mr, None)
    f_plain = be.conv(x, wf2, True, bt2, res, y, g, True, False, False, (None, so))
    def both(s): f_fold(s); f_plain(s)
    print('conv+stats %.1f us | fold %.1f us | fold+conv %.1f us | conv with fold in prologue %.1f us' % (t(f_plain), t(f_fold), t(both), t(f_fused)))
    sys.exit(0)
if 'full' not in sys.argv:
    # the fused backward pair: weight gradient + slab contraction (+ subset sums of gy), then the data gradient that writes the
    # finished gradient (coefficients in its prologue; 0 / 1 / 2 extra terms)
    wd = mk((27, c, c)) * 0.05
    qp = torch.zeros(n, S, c, device='cuda'); tg = torch.zeros(n, S, 27, 32, device='cuda')
    mr = torch.rand(n, 8, 2, device='cuda') + 0.5; gam = torch.rand(c, device='cuda') + 0.5
    cf = torch.zeros(n * c * 5, device='cuda'); cs = torch.zeros(n, 64, c, device='cuda')
    dx = View(mk((n, d, h, w, c)), n, d, h, w, c, 0, c)
    e1 = View(mk((n, d, h, w, c)), n, d, h, w, c, 0, c); e2 = View(mk((n, d, h, w, c)), n, d, h, w, c, 0, c)
    gn = dict(qpart=qp, q_nsplit=S, p=None, tg=tg, csum_out=cs, mr=mr, gamma=gam, groups=8, coeff_out=cf)
    print('wgrad_q %.1f us' % t(be.wgrad_q(y, x, g, S, gp, wd, qp, None)), 'wgrad_q+tg %.1f us' % t(be.wgrad_q(y, x, g, S, gp, wd, qp, tg)))
    print('dgrad plain+stats %.1f us' % t(be.conv(y, wd, False, None, None, dx, g, False, True, False, (x, torch.zeros(n, 32, c, 2, device='cuda')))),
          'fused(no gn) %.1f' % t(be.conv_dgrad_fused(y, wd, x, None, [], True, dx, g)),
          'fused(gn) %.1f' % t(be.conv_dgrad_fused(y, wd, x, None, [], True, dx, g, None, gn)),
          'fused(gn,+1) %.1f' % t(be.conv_dgrad_fused(y, wd, x, None, [(e1, None)], True, dx, g, None, gn)),
          'fused(gn,+2) %.1f' % t(be.conv_dgrad_fused(y, wd, x, None, [(e1, None), (e2, None)], True, dx, g, None, gn)))
if 'full' in sys.argv: sys.exit(0)   # PMC passes: only the full-resolution launches
# level-1 sized problem (8 x 32 x 80): few bricks per workgroup, ragged W
d1, h1, w1 = 8, 32, 80
g1 = Geom(n, d1, h1, w1, d1, h1, w1, c, c, 3, 1, 1)
x1 = View(mk((n, d1, h1, w1, c)), n, d1, h1, w1, c, 0, c)
y1 = View(mk((n, d1, h1, w1, c)), n, d1, h1, w1, c, 0, c)
S1 = be.wgrad_nsplit(g1)
gp1 = torch.zeros(n, S1, 27, c, c, device='cuda')
print('level-1: conv_tiled %.1f us' % t(be.conv(x1, wf, True, bt, None, y1, g1, True, False, False)),
      'wgrad_tiled %.1f us' % t(be.wgrad(y1, x1, g1, S1, gp1)))
