"""Phase-loop cycle breakdown of conv_tiled_kernel (workgroup 0), from the -DRTP_TILED_PROF build (tools/tiled_prof.sh):
    RTP_LIB=rt_pose_amd/lib/librtp_hip_prof.so python tools/tiled_prof.py
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
x, y, res = (View(mk((n, d, h, w, c)), n, d, h, w, c, 0, c) for _ in range(3))
wf = mk((n, 27, c, c)) * 0.05
bt = torch.randn(n, 64, c, device='cuda')
f = be.conv(x, wf, True, bt, res, y, g, True, False, False)
s = be.stream()
for _ in range(200): f(s)
torch.cuda.synchronize()
fn0 = ctypes.CDLL(os.environ['RTP_LIB']).rtp_tiled_prof_read; fn0.argtypes = [ctypes.c_void_p]
b0 = (ctypes.c_longlong * 72)(); fn0(b0); t_a = b0[67]
for _ in range(100): f(s)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 72)()
fn = be.lib.rtp_tiled_prof_read if hasattr(be.lib, 'rtp_tiled_prof_read') else ctypes.CDLL(os.environ['RTP_LIB']).rtp_tiled_prof_read
fn.argtypes = [ctypes.c_void_p]
assert fn(buf) == 0
names = ['pre', 'mfma', 'bar_c', 'issue', 'epi', 'bar_l', 'nC', 'nL']
print('dbg', os.environ.get('RTP_TILED_DBG', '0'), 'sync', os.environ.get('RTP_TILED_SYNC', '0'), '(cycles per phase; s_memtime ticks)')
for wv in range(8):
    v = buf[wv * 8:(wv + 1) * 8]
    nc, nl = max(v[6], 1), max(v[7], 1)
    print('wave %d  ' % wv + '  '.join('%s %6.0f' % (names[k], v[k] / (nc if k < 3 else nl)) for k in range(6)) + '   nC %d nL %d' % (v[6], v[7]))
print('workgroup 0 in-kernel %.2f us = %d cycles -> %.2f GHz' % (buf[68] / 100.0, buf[64] + buf[65] + buf[66], (buf[64] + buf[65] + buf[66]) / (buf[68] * 10.0)))
print('weights staged %d  setup %d  phase loop %d cycles;  launch-to-launch period %.2f us (s_memrealtime over 100 launches)' % (buf[64], buf[65], buf[66], (buf[67] - t_a) / 100.0 / 100.0))
wg = (ctypes.c_longlong * 1024)()
fw = ctypes.CDLL(os.environ['RTP_LIB']).rtp_tiled_prof_wgs; fw.argtypes = [ctypes.c_void_p]; fw(wg)
import numpy as np
a = np.array(list(wg), dtype=np.int64).reshape(512, 2)[:256]
t0 = a[:, 0].min()
st, en = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0
print('workgroup start us: min %.2f max %.2f | end us: min %.2f median %.2f max %.2f | duration: min %.2f median %.2f max %.2f' % (
    st.min(), st.max(), en.min(), np.median(en), en.max(), (en - st).min(), np.median(en - st), (en - st).max()))
xcd = np.arange(256) % 8
print('end by XCD (max):', ' '.join('%.1f' % en[xcd == k].max() for k in range(8)), ' duration by XCD (median):', ' '.join('%.1f' % np.median((en - st)[xcd == k]) for k in range(8)))
