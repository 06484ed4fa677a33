"""Data-parallel path on CPU: 2 processes, gloo, the emulated kernels.  Checks the sharding + single flat all-reduce
+ optimiser wiring of rt_pose_amd.trainer.DataParallelTrainer (on the GPU box the same code runs over RCCL/xGMI):
  * both ranks hold identical parameters after every step;
  * the step equals a single-process replay that averages the two ranks' gradients by hand (per-rank loss
    normalisers, exactly as the reference's DDP does -- SURVEY.md 8e)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rt_pose_amd import configs, synth
from rt_pose_amd.engine import one_cycle
from rt_pose_amd.trainer import DataParallelTrainer
from tests.emu_backend import EmuBackend

DIMS, B, STEPS = (8, 16, 16), 2, 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    tr = DataParallelTrainer("hr3d", B, DIMS, total_steps=10, rank=rank, world_size=world, backend=EmuBackend(exact=True), seed=0)
    for step in range(STEPS):
        tr.step(synth.make_batch(B, 1, DIMS, seed=100 + step, rank=rank))
    torch.save({"p": tr.flat.p.clone(), "loss": float(tr.losses()["loss"])}, os.path.join(out, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_step_matches_manual_average(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = [torch.load(os.path.join(tmp_path, "rank%d.pt" % r)) for r in range(world)]
    assert torch.equal(r0["p"], r1["p"]), "replicas diverged"
    assert r0["loss"] != r1["loss"], "ranks must see different shards"
    # single-process replay: two engines with shared weights, gradients averaged by hand
    # (same intra-op thread count as the workers: Adam's first steps are ~lr*sign(g), so a last-bit difference in a
    # near-zero gradient element would otherwise show up as a 2*lr parameter difference)
    nthreads = torch.get_num_threads()
    torch.set_num_threads(2)
    trs = [DataParallelTrainer("hr3d", B, DIMS, total_steps=10, backend=EmuBackend(exact=True), seed=0) for _ in range(world)]
    for step in range(STEPS):
        for r, tr in enumerate(trs):
            tr.load(synth.make_batch(B, 1, DIMS, seed=100 + step, rank=r))
            tr._fwd_bwd()
        avg = sum(tr.flat.g for tr in trs) / world
        lr, b1 = one_cycle(step, 10, configs.spec("hr3d")["lr_max"])
        for tr in trs:
            tr.flat.g.copy_(avg)
            tr.opt.set_hyper(lr, b1)
            tr.opt.run()
    torch.set_num_threads(nthreads)
    d = (trs[0].flat.p - r0["p"]).abs()
    frac = float((d > 1e-6).float().mean())
    assert frac < 1e-3, ("fraction of parameters off by more than 1e-6", frac, float(d.max()))


def _worker_buckets(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    res = {}
    for nb in (1, 2):
        tr = DataParallelTrainer("hr3d", B, DIMS, total_steps=10, rank=rank, world_size=world, backend=EmuBackend(exact=True), seed=0,
                                 ar_buckets=nb)
        assert tr.ar_buckets == nb
        if nb == 2:
            tags = [L.tag for L in tr.engine.bwd]
            i = tags.index("allreduce:early")
            assert tags[i - 1] == "tail" and tags[-1] == "tail" and 0 < tr.ar_split < tr.flat.numel
            # the early bucket is exactly transition2 .. pose_head: everything the sweep has finished when it leaves stage 3
            first = [k for k, o in tr.flat.offsets.items() if o == tr.ar_split][0]
            assert first.startswith("backbone.backbone.transition2."), first
        for step in range(STEPS):
            tr.step(synth.make_batch(B, 1, DIMS, seed=100 + step, rank=rank))
        res[nb] = tr.flat.p.clone()
    torch.save(res, os.path.join(out, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_gradient_buckets_equal_one(tmp_path):
    """ar_buckets=2 (early tail flush + all-reduce of the transition2 .. pose_head suffix queued inside the backward list, the rest
    after the sweep) trains exactly like the single all-reduce: same parameters on both ranks after two steps."""
    world = 2
    mp.spawn(_worker_buckets, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = [torch.load(os.path.join(tmp_path, "rank%d.pt" % r)) for r in range(world)]
    assert torch.equal(r0[2], r1[2]), "replicas diverged with two buckets"
    d = (r0[1] - r0[2]).abs()
    assert float((d > 1e-6).float().mean()) < 1e-3, ("two buckets vs one", float(d.max()))
