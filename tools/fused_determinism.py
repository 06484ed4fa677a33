#!/usr/bin/env python
"""Bitwise repeatability of the fused-backward kernels on identical inputs (B=8 full-resolution layer)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rt_pose_amd.backend import HipBackend
from rt_pose_amd.graph import Geom, View
be = HipBackend('cuda:0')
n, d, h, w, c = 8, 16, 64, 160, 32
g = Geom(n, d, h, w, d, h, w, c, c, 3, 1, 1)
torch.manual_seed(0)
mk = lambda shape: torch.randn(shape, device='cuda').to(torch.bfloat16)
V = lambda t: View(t, n, d, h, w, c, 0, c)
x, gy, e1, e2 = V(torch.relu(mk((n, d, h, w, c))).contiguous()), V(mk((n, d, h, w, c))), V(mk((n, d, h, w, c))), V(mk((n, d, h, w, c)))
wd = mk((27, c, c)) * 0.05
S = be.wgrad_nsplit(g)
s = be.stream()
def rep(name, fn, outs, k=5):
    ref = None
    worst = 0.0
    for i in range(k):
        for o in outs: o.zero_()
        fn(s); torch.cuda.synchronize()
        cur = [o.clone() for o in outs]
        if ref is None: ref = cur
        else:
            for a, b in zip(ref, cur):
                dd = (a.float() - b.float()).abs().max().item()
                worst = max(worst, dd / (a.float().abs().max().item() + 1e-30))
    print("%-34s max rel diff between runs %.3e" % (name, worst))
gp = torch.zeros(n, S, 27, c, c, device='cuda'); qp = torch.zeros(n, S, c, device='cuda'); tg = torch.zeros(n, S, 27, 32, device='cuda')
rep("wgrad_q (slabs, qpart)", be.wgrad_q(gy, x, g, S, gp, wd, qp, None), [gp, qp])
rep("wgrad_q + tg", be.wgrad_q(gy, x, g, S, gp, wd, qp, tg), [gp, qp, tg])
be.wgrad_q(gy, x, g, S, gp, wd, qp, tg)(s); torch.cuda.synchronize()   # leave valid qp / tg behind (tg accumulated once more: fine)
mr = torch.rand(n, 8, 2, device='cuda') + 0.5; gam = torch.rand(c, device='cuda') + 0.5
cf = torch.zeros(n * c * 5, device='cuda'); cs = torch.zeros(n, 64, c, device='cuda')
dx = V(torch.zeros(n, d, h, w, c, device='cuda', dtype=torch.bfloat16))
ts = be.conv_stats_nsplit(gy, g, True); tot = torch.zeros(n, ts, 32, device='cuda')
gn = dict(qpart=qp, q_nsplit=S, p=None, tg=tg, csum_out=cs, mr=mr, gamma=gam, groups=8, coeff_out=cf)
rep("dgrad fused (no gn, mask)", be.conv_dgrad_fused(gy, wd, x, None, [], True, dx, g), [dx.buf])
rep("dgrad fused (no gn, mask, totals)", be.conv_dgrad_fused(gy, wd, x, None, [], True, dx, g, tot), [dx.buf, tot])
rep("dgrad fused (gn from tg)", be.conv_dgrad_fused(gy, wd, x, None, [], True, dx, g, None, gn), [dx.buf, cf, cs])
rep("dgrad fused (gn, +1 extra)", be.conv_dgrad_fused(gy, wd, x, None, [(e1, None)], True, dx, g, None, gn), [dx.buf, cf])
rep("dgrad fused (gn, +2 extras)", be.conv_dgrad_fused(gy, wd, x, None, [(e1, None), (e2, cf.clone())], False, dx, g, None, gn), [dx.buf, cf])
