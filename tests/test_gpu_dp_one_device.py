"""The multi-rank PRODUCT path on ONE GPU: two fresh child processes on cuda:0, gloo with device tensors, each
a DataParallelTrainer(world_size=2) on the HIP backend, one gradient bucket and two.

What a 1-GPU box cannot show is RCCL's transport (tests/test_gpu_rccl.py needs two GPUs); everything else of the data-parallel
step runs here on hardware: the batch split by rank, ONE flat SUM all-reduce of the fp32 gradient buffer -- or, with two buckets,
the transition2 .. pose_head suffix kicked from INSIDE the backward launch list on the lane's stream (torch.cuda.ExternalStream,
trainer._install_early_bucket) while the sweep goes on -- and 1 / world folded into the optimiser kernels.  Checked like
tests/test_dp_gloo.py: replicas stay bit-identical, and the result equals a single-process replay that sums the two ranks' gradients
by hand (reference: det3d/torchie/apis/train.py:284-291 DDP wrap, det3d/core/utils/dist_utils.py:31-57).

The children are started with subprocess from a fresh interpreter -- never forked from, or exec'd by, a process that has touched
the GPU."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
DIMS, B, STEPS = (8, 16, 32), 2, 3
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
@pytest.mark.parametrize("buckets", [1, 2])
def test_two_ranks_on_one_gpu_match_manual_sum(tmp_path, buckets):
    from rt_pose_amd import configs, synth
    from rt_pose_amd.engine import one_cycle
    from rt_pose_amd.trainer import DataParallelTrainer
    from tests.util import run_ranks
    world = 2
    # (files for the children's output, one deadline for both, both killed on expiry: tests/util.py)
    run_ranks([sys.executable, os.path.join(ROOT, "tests", "dp_one_device_child.py"), str(tmp_path), str(buckets)], world, str(tmp_path),
              {"PYTHONPATH": ROOT + os.pathsep + os.environ.get("PYTHONPATH", "")})
    r0, r1 = [torch.load(os.path.join(tmp_path, "rank%d.pt" % r)) for r in range(world)]
    assert torch.equal(r0["p"], r1["p"]), "replicas diverged"
    assert all(torch.equal(a, b) for a, b in zip(r0["g"], r1["g"])), "all-reduced gradients differ between ranks"
    assert r0["loss"] != r1["loss"], "ranks must see different shards"
    assert r0["allreduce_ms"] is not None and r0["allreduce_ms"] > 0
    if buckets == 2:
        assert r0["split"] is not None and r0["split"] == r1["split"]
    # single-process replay on this GPU: the two shards' gradients summed by hand, same optimiser rule (grad_scale = 1 / world)
    trs = [DataParallelTrainer("hr3d", B, DIMS, total_steps=10, device="cuda:0", use_graph=False, seed=0) for _ in range(world)]
    for step in range(STEPS):
        for r, tr in enumerate(trs):
            tr.load(synth.make_batch(B, 1, DIMS, seed=100 + step, rank=r))
            with tr._on_stream():
                tr._fwd_bwd()
        torch.cuda.synchronize()
        tot = sum(tr.flat.g for tr in trs)
        d = (tot.float().cpu() - r0["g"][step]).norm() / r0["g"][step].norm()
        assert float(d) < 1e-5, (step, float(d))        # class-sum atomics reorder: 1e-9-level noise, amplified by Adam later
        lr, b1 = one_cycle(step, 10, configs.spec("hr3d")["lr_max"])
        for tr in trs:
            tr.flat.g.copy_(tot)
            with tr._on_stream():
                tr.opt.set_hyper(lr, b1, grad_scale=1.0 / world)
                tr.opt.run()
            tr.step_idx += 1
        torch.cuda.synchronize()
    # parameters: Adam's first steps are ~lr * sign(g), so last-bit gradient noise in near-zero entries can move single parameters by
    # up to 2 * lr; norm-wise the replicas and the replay agree
    p = trs[0].flat.p.float().cpu()
    assert float((p - r0["p"]).norm() / r0["p"].norm()) < 1e-4
