#!/usr/bin/env python
"""List order, lane and dependencies of the hr3d launch plans, built on the CPU with the tests' emulated backend (small dims: the
order does not depend on them).  tools/plan_order.py [fwd|bwd] [config]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.emu_backend import EmuBackend
from rt_pose_amd import configs, lanes
from rt_pose_amd.engine import PoseEngine, FlatParams
from rt_pose_amd.trainer import init_state_dict
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
name = sys.argv[2] if len(sys.argv) > 2 else "hr3d"
be = EmuBackend()
s = configs.spec(name)
shapes = configs.param_shapes(name)
flat = FlatParams(shapes, be.alloc)
flat.load_state_dict(init_state_dict(shapes, 0))
eng = PoseEngine(be, flat.values, s["arch"], s["final_fuse"], s["heads"], s["weight"], s["code_weights"], 1, (8, 16, 32), train=True, pgrads=flat.grads)
L = eng.fwd if which == "fwd" else eng.bwd
preds = lanes._order_preds(L)
for i, x in enumerate(L):
    cross = [L[j].tag for j in sorted(preds[i]) if L[j].lane != x.lane]
    print("%3d lane %d  %-28s <- %s" % (i, x.lane, x.tag, ", ".join(cross)))
