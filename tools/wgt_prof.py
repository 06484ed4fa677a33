"""Phase cycle breakdown of wgrad_tiled_kernel (workgroup 0), from a -DRTP_WGT_PROF build:
    tools/variant.sh wgrad_tiled wgtprof -DRTP_WGT_PROF && RTP_LIB=rt_pose_amd/lib/librtp_hip_wgtprof.so python tools/wgt_prof.py
"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rt_pose_amd.backend import HipBackend
from rt_pose_amd.graph import Geom, View
be = HipBackend('cuda:0')
n, d, h, w, c = 8, 16, 64, 160, 32
g = Geom(n, d, h, w, d, h, w, c, c, 3, 1, 1)
mk = lambda shape: torch.randn(shape, device='cuda').to(torch.bfloat16)
x, gy = (View(mk((n, d, h, w, c)), n, d, h, w, c, 0, c) for _ in range(2))
ns = 32
slabs = torch.zeros(n, ns, 27, c, c, device='cuda')
f = be.wgrad(gy, x, g, ns, slabs)
s = be.stream()
for _ in range(100): f(s)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100): f(s)
e1.record(); torch.cuda.synchronize()
print('launch-to-launch %.2f us' % (e0.elapsed_time(e1) * 10))
buf = (ctypes.c_longlong * 16)()
fn = ctypes.CDLL(os.environ['RTP_LIB']).rtp_wgt_prof_read; fn.argtypes = [ctypes.c_void_p]
assert fn(buf) == 0
nb, ni = max(buf[2], 1), max(buf[7], 1)
print('consumer per brick: mfma code %d  barrier wait %d  (%d bricks)' % (buf[0] / nb, buf[1] / nb, buf[2]))
print('loader per iteration: dma issue %d  subset sums %d  barrier wait %d  (%d iterations)' % (buf[4] / ni, buf[5] / ni, buf[6] / ni, buf[7]))
print('kernel %d cycles, %.2f us -> %.2f GHz' % (buf[8], buf[9] / 100.0, buf[8] / (buf[9] * 10.0)))
