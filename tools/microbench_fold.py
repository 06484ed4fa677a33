import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rt_pose_amd.backend import HipBackend
from rt_pose_amd.graph import Geom
be = HipBackend('cuda:0')
def t(f, it=300):
    s = be.stream()
    for _ in range(100): f(s)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): f(s)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e6
n, d, h, w = 8, 16, 64, 160
for c, S in ((32, 32), (64, 40)):
    g = Geom(n, d, h, w, d, h, w, c, c, 3, 1, 1)
    W = torch.randn(c, c, 3, 3, 3, device='cuda') * 0.05
    gam, bet = torch.ones(c, device='cuda'), torch.zeros(c, device='cuda')
    stats = torch.rand(n, S, c, 2, device='cuda') * 1000 + 5000
    wf = torch.zeros(n, 27, c, c, device='cuda', dtype=torch.bfloat16)
    bt = torch.zeros(n, 64, c, device='cuda'); mr = torch.zeros(n, 8, 2, device='cuda')
    f = be.fold_fwd(W, None, gam, bet, stats, S, 8, 1e-5, g, c, c, wf, bt, mr, None)
    print('dbg', os.environ.get('RTP_FOLD_DBG', '0'), 'c', c, 'fold_fwd %.1f us' % t(f))
pq = torch.rand(n, 32, 32, 2, device='cuda'); coeff = torch.zeros(n * 32 * 5, device='cuda')
print('gn_bwd_coeffs %.1f us' % t(be.gn_bwd_coeffs(pq, 32, mr, gam[:32].contiguous(), n, 32, 8, d * h * w, coeff, None, None, 0)))
