"""Print the training plan's launches per lane (tags in issue order) at the bench workload -- the chains whose length sets how
long the main lane waits at the stage boundaries."""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rt_pose_amd import configs
from rt_pose_amd.trainer import DataParallelTrainer
name = sys.argv[1] if len(sys.argv) > 1 else "hr3d"
tr = DataParallelTrainer(name, 8, configs.NATIVE_DIMS, total_steps=100, use_graph=False)
for phase, ls in (("fwd", tr.engine.fwd), ("bwd", tr.engine.bwd)):
    by = collections.defaultdict(list)
    for L in ls:
        by[L.lane].append(L.tag)
    for lane in sorted(by):
        print(phase, "lane", lane, len(by[lane]), "launches:", " ".join(by[lane]))
