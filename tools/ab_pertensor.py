#!/usr/bin/env python
"""Per-parameter comparison of two tools/ab_grads.py outputs: which tensors differ (names from configs.param_shapes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rt_pose_amd import configs
a, b = np.load(sys.argv[1])["g"].astype(np.float64), np.load(sys.argv[2])["g"].astype(np.float64)
shapes = configs.param_shapes(sys.argv[3] if len(sys.argv) > 3 else "hr3d")
off = 0
rows = []
for k, sh in shapes.items():
    n = int(np.prod(sh))
    n_al = n
    x, y = a[off:off + n], b[off:off + n]
    d = np.linalg.norm(x - y) / max(np.linalg.norm(x), 1e-30)
    if d > 0:
        rows.append((d, k))
    off += n_al
print("%d tensors differ" % len(rows))
for d, k in sorted(rows, reverse=True)[:25]:
    print("%.3e  %s" % (d, k))
