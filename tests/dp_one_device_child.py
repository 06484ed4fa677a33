"""Child process of tests/test_gpu_dp_one_device.py: one rank of a DataParallelTrainer job whose ranks all sit on cuda:0 and talk
over gloo with DEVICE tensors (a 1-GPU box cannot run RCCL between ranks).  Everything the multi-rank product path does beyond the
collective's transport runs for real: rank-sharded batches, the flat fp32 gradient buffer, `ar_buckets` 1 / 2 (the early bucket is
kicked from inside the backward launch list under torch.cuda.ExternalStream), grad_scale = 1 / world inside the optimiser kernels.
    RANK=r WORLD_SIZE=n MASTER_ADDR=127.0.0.1 MASTER_PORT=p python tests/dp_one_device_child.py <out dir> <buckets>
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out, buckets = sys.argv[1], int(sys.argv[2])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])   # MASTER_ADDR / MASTER_PORT come with them (run_ranks)
    import datetime
    import faulthandler
    faulthandler.dump_traceback_later(float(os.environ.get("RTP_HANG_DUMP_S", "180")), exit=True)   # a stuck rank names its line
    from rt_pose_amd import pin_hw_queues
    pin_hw_queues()   # before torch selects a device: GPU_MAX_HW_QUEUES is read when HIP initialises (set_device does that)
    import torch
    import torch.distributed as dist
    from rt_pose_amd import synth
    from rt_pose_amd.trainer import DataParallelTrainer
    from tests.test_gpu_dp_one_device import B, DIMS, STEPS
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
    tr = DataParallelTrainer("hr3d", B, DIMS, total_steps=10, device="cuda:0", rank=rank, world_size=world, use_graph=False, seed=0,
                             ar_buckets=buckets)
    assert tr.be.name == "hip" and tr.ar_buckets == buckets, (tr.be.name, tr.ar_buckets)
    if buckets == 2:   # the early bucket is a launch of the backward list, right behind a tail flush
        tags = [L.tag for L in tr.engine.bwd_plan.launches]
        k = tags.index("allreduce:early")
        assert tags[k - 1].startswith("tail") and 0 < tr.ar_split < tr.flat.numel, (tags[k - 1], tr.ar_split)
    grads = []
    for step in range(STEPS):
        tr.step(synth.make_batch(B, 1, DIMS, seed=100 + step, rank=rank))
        torch.cuda.synchronize()
        grads.append(tr.flat.g.float().cpu().clone())   # after the all-reduce: the SUM over ranks
    torch.save({"p": tr.flat.p.float().cpu(), "g": grads, "loss": float(tr.losses()["loss"]), "allreduce_ms": tr.allreduce_ms(),
                "split": getattr(tr, "ar_split", None)}, os.path.join(out, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
