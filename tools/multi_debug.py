"""Why does rtp_multi_end refuse a pair?  RTP_MERGE_DEBUG=1 python tools/multi_debug.py <n>"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rt_pose_amd.backend import HipBackend
from rt_pose_amd.graph import Geom, View
hip = HipBackend("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ci = co = 32
fns, fw = [], []
for k, (d, h, w) in enumerate(((8, 64, 128), (4, 32, 64))):
    g = Geom(n, d, h, w, d, h, w, ci, co, 3, 1, 1)
    mk = lambda: View(torch.randn(n, d, h, w, ci, device="cuda").bfloat16(), n, d, h, w, ci, 0, ci)
    x, y = mk(), mk()
    wf = torch.randn(n, 27, co, ci, device="cuda").bfloat16()
    fns.append(hip.conv(x, wf, True, None, None, y, g, True, False, False))
    S = hip.wgrad_nsplit(g)
    print("problem", k, "conv slots", hip.conv_stats_nsplit(x, g, False), "wgrad slots", S)
    fw.append(hip.wgrad(y, x, g, S, torch.zeros(n, S, 27, co, ci, device="cuda")))
print("conv multi:", hip.multi(fns) is not None)
print("wgrad multi:", hip.multi(fw) is not None)
