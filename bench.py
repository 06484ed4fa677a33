#!/usr/bin/env python
"""HRRadarPose training throughput on MI355X (radar frames/s), BASELINE.json's metric.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--model hr3d] [--batch 8]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = forward + losses + backward + gradient all-reduce + clip/Adam on one batch of synthetic radar tensors of
the dataset-native shape [B,Cin,16,64,160] per GPU (weak scaling), inputs resident in HBM.  Rank 0 prints ONE JSON
line.  At N=1 the line also carries:
  roofline     -- the dominant kernel family's algorithmic FLOP/s (HIP events around every launch of that family,
                  on the launch stream, over K eagerly launched steps right after the timed region) vs the dense
                  bf16 MFMA peak;
  cpu_baseline -- the oracle (oracle/hrradarpose_ref.py, PyTorch CPU fp32) timed on this host on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from rt_pose_amd import pin_hw_queues  # noqa: E402
HW_QUEUES = pin_hw_queues()   # the lane plan's measured optimum (HIP's default), pinned before torch initialises HIP

PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA, MI355X_MICROARCH.md chip table
PEAK_HBM_GBPS = 8000.0     # HBM3E peak (same guide; ~6.3 TB/s achievable)


def cpu_baseline(name, seconds_budget=90.0):
    """Oracle train step (fwd + loss + bwd + clip + Adam) on the host cores; reported, never the target.
    SURVEY.md 8d: B = 2 frames at the native shape, fp32, one warm-up step, median of 3 timed steps, at the host's physical
    core count (stated); plus an 8-thread datapoint (the authoring container's size) and, on hosts with more than 32
    physical cores, a 32-thread datapoint (PyTorch's CPU conv stops scaling well before 100+ threads).  `value` is the best
    datapoint."""
    import statistics
    import torch
    from oracle import hrradarpose_ref as O
    arch, fin, fout, fuse, heads, weight, cw = O.MODEL_CONFIGS[name]
    shapes = O.param_shapes(arch, fin, fout, fout, heads)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or avail
    except Exception:
        phys = avail
    phys = max(1, min(phys, avail))
    batch, dims = 2, (16, 64, 160)
    ex = O.synth_example(batch, O.ARCHS[arch]["inplanes"], dims, seed=1234, one_hm=heads["hm"] == 1)

    def run(threads, warm, timed, budget):
        torch.set_num_threads(threads)
        sd = {k: v.requires_grad_(True) for k, v in O.seeded_state_dict(shapes, seed=1).items()}
        opt = O.AdamTrueWD(list(sd.values()))
        ts, t_begin = [], time.perf_counter()
        for i in range(warm + timed):
            t0 = time.perf_counter()
            for p in sd.values():
                p.grad = None
            O.radar_pose_net(sd, ex, fuse, weight, cw)["loss"][0].backward()
            opt.step(1e-4, 0.95)
            if i >= warm:
                ts.append(time.perf_counter() - t0)
            if time.perf_counter() - t_begin > budget and ts:
                break
        return statistics.median(ts), len(ts)

    # SURVEY 8d protocol for EVERY datapoint: one warm-up step, then the median of 3 timed steps (a time box per datapoint only
    # guards against a pathological host: it can cut the timed steps short, never the warm-up).  PyTorch's CPU conv3d stops scaling
    # long before 100+ threads (measured on the GPU box's host: 0.24 frames/s at 128 threads, 0.88-0.92 at 8-32), so the line's
    # `value` is the BEST datapoint with its thread count in `cores`; every datapoint, the physical-core one included, is listed.
    points = {}
    points[phys] = run(phys, 1, 3, seconds_budget * 0.5)
    if phys > 32:
        points[32] = run(32, 1, 3, seconds_budget * 0.25)
    if 8 not in points and avail >= 8:
        points[8] = run(8, 1, 3, seconds_budget * 0.25)
    best = min(points, key=lambda t: points[t][0])
    return dict(value=round(batch / points[best][0], 4), unit="frames/s", cores=best, kind="port", physical_cores=phys,
                datapoints={str(t): {"frames_per_s": round(batch / m, 4), "timed_steps": k} for t, (m, k) in points.items()},
                sample="train steps (fwd + loss + bwd + clip + Adam) of batch %d at [B,%d,16,64,160], fp32, oracle/hrradarpose_ref.py; "
                       "every thread count: 1 warm-up + median of 3 timed steps; value = the best thread count's datapoint (cores = "
                       "that count; the host has %d physical cores)" % (batch, O.ARCHS[arch]["inplanes"], phys))


def torch_gpu_child_run(name="hr3d", batch=8, amp=False, steps=3, warm=2, dev="cuda:0"):
    import torch
    from oracle import hrradarpose_ref as O
    arch, fin, fout, fuse, heads, weight, cw = O.MODEL_CONFIGS[name]
    shapes = O.param_shapes(arch, fin, fout, fout, heads)
    sd = {k: v.to(dev).requires_grad_(True) for k, v in O.seeded_state_dict(shapes, seed=1).items()}
    opt = O.AdamTrueWD(list(sd.values()))
    ex = O.synth_example(batch, O.ARCHS[arch]["inplanes"], (16, 64, 160), seed=1234, one_hm=heads["hm"] == 1)

    def to_dev(v):
        if torch.is_tensor(v):
            return v.to(dev)
        if isinstance(v, dict):
            return {k: to_dev(x) for k, x in v.items()}
        if isinstance(v, (list, tuple)):
            return type(v)(to_dev(x) for x in v)
        return v
    ex = to_dev(ex)

    def step():
        for p in sd.values():
            p.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            loss = O.radar_pose_net(sd, ex, fuse, weight, cw)["loss"][0]
        loss.backward()
        opt.step(1e-4, 0.95)
        return loss

    t0 = time.perf_counter()
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t_warm = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    return dict(value=round(batch * steps / el, 2), unit="frames/s", ms_per_step=round(1e3 * el / steps, 2),
                dtype="bf16 autocast" if amp else "fp32", batch=batch, warmup_s=round(t_warm, 1), loss=float(loss),
                peak_mem_GB=round(torch.cuda.max_memory_allocated() / 1e9, 1))



def torch_gpu_baseline(name, batch, limit_s=300):
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--torch-gpu-child", name, str(batch)]
    out = {"what": "oracle/hrradarpose_ref.py (the reference's model + train step in plain PyTorch) with its tensors on this GPU: "
                   "ATen / MIOpen kernels, eager autograd, %d frames" % batch, "unit": "frames/s"}
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=limit_s)
        for ln in r.stdout.splitlines():
            if ln.startswith("TORCH_GPU "):
                d = json.loads(ln[len("TORCH_GPU "):])
                out[d["dtype"].replace(" ", "_")] = {"value": d["value"], "ms_per_step": d["ms_per_step"]}
        if len(out) == 2:
            out["error"] = "no result (exit code %d)" % r.returncode
    except subprocess.TimeoutExpired as e:
        for ln in (e.stdout or b"").decode(errors="replace").splitlines() if isinstance(e.stdout, bytes) else (e.stdout or "").splitlines():
            if ln.startswith("TORCH_GPU "):
                d = json.loads(ln[len("TORCH_GPU "):])
                out[d["dtype"].replace(" ", "_")] = {"value": d["value"], "ms_per_step": d["ms_per_step"]}
        out["note"] = "stopped after %d s" % limit_s
    except Exception as e:  # informational leg: never fail the bench line
        out["error"] = repr(e)
    return out


def dcn_op_bench(dev, batch, iters=10):
    import torch
    from rt_pose_amd.dcn import deform_conv
    n, c, h, w, co = batch * 16, 32, 64, 160, 32
    g = torch.Generator(device=dev).manual_seed(7)
    x = torch.randn(n, c, h, w, device=dev, generator=g).requires_grad_(True)
    off = (torch.randn(n, 4 * 18, h, w, device=dev, generator=g) * 0.5).requires_grad_(True)
    wt = (torch.randn(co, c, 3, 3, device=dev, generator=g) * 0.05).requires_grad_(True)
    gy = torch.randn(n, co, h, w, device=dev, generator=g)

    def fwd():
        return deform_conv(x, off, wt, 1, 1, 1, 1, 4, 64)

    def fb():
        for v in (x, off, wt):
            v.grad = None
        fwd().backward(gy)

    def timed(f):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            f()
        torch.cuda.synchronize()
        return round((time.perf_counter() - t0) / iters * 1e3, 3)

    out = {"forward_ms": timed(fwd), "forward_backward_ms": timed(fb)}
    out["backward_ms"] = round(out["forward_backward_ms"] - out["forward_ms"], 3)
    # the same call with every sample within a pixel of its tap (what the DCN head's zero-initialised offset conv produces): the
    # one-pass backward takes its 3 x 3-footprint branch
    with torch.no_grad():
        off.mul_(0.4)
    out["offsets_within_a_pixel"] = {"offset_sigma_px": 0.2, "forward_ms": timed(fwd), "forward_backward_ms": timed(fb)}
    out["offset_sigma_px"] = 0.5
    alg = (x.numel() + off.numel() + gy.numel()) * 4
    out.update(workload="DCNv1 3x3 [%d,%d,%d,%d] -> %d, deformable_groups 4, im2col_step 64; fp32 tensors in / out; forward: exact fp32 MFMA; "
                        "backward (one-pass kernel): both matrix products on the bf16 matrix core with (hi, lo) operand splits, lo x lo "
                        "dropped = 3.9e-6 norm-wise (RTP_DCN_FP32_MFMA=1: exact fp32 products) (BASELINE config 4, op level)" % (n, c, h, w, co),
               arithmetic={"forward": "fp32 (v_mfma_f32_32x32x2_f32)", "backward": "bf16x3 split products, fp32 accumulation"},
               forward_algorithmic_GBps=round(alg / out["forward_ms"] / 1e6, 1),
               forward_tflops_fp32=round(2.0 * n * h * w * co * c * 9 / out["forward_ms"] / 1e9, 2))
    return out


def lidar_bench(dev, batch, npts=200000, iters=5):
    import numpy as np
    import torch
    from rt_pose_amd.lidar import DynamicVoxelEncoder, lidar_to_radar
    enc = DynamicVoxelEncoder([0.0, -10.05, -5.8, 11.6, 10.05, 5.8], [0.0725, 0.314, 0.725])
    g = torch.Generator(device=dev).manual_seed(11)
    frames = [torch.rand(npts, 4, device=dev, generator=g) * torch.tensor([14.0, 24.0, 14.0, 1.0], device=dev)
              - torch.tensor([1.0, 12.0, 7.0, 0.0], device=dev) for _ in range(batch)]
    P = np.eye(4)
    P[:3, 3] = [0.1, -0.05, 0.2]

    def run():
        nv = 0
        for f in frames:
            pts = f.clone()
            lidar_to_radar(pts, P)   # in place
            v, c = enc.voxelize(pts)
            enc.to_dense(v, c)
            nv += v.shape[0]
        return nv

    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        nv = run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    out = {"workload": "%d LiDAR frames x %d points: extrinsic transform + dynamic voxelisation (stable radix sort, in-order means) "
                       "+ dense scatter (BASELINE config 5 pieces)" % (batch, npts),
           "ms_per_batch": round(ms, 3), "points_per_s": round(batch * npts / ms * 1e3, 1), "voxels": int(nv)}
    # ... and the fused model: hr3d whose head towers read concat(radar feature, dense LiDAR grid) (configs.LIDAR_VARIANTS;
    # this repo's composition, the reference ships no fusion detector) -- the same train step as the headline, LiDAR grid resident
    try:
        from rt_pose_amd import configs, synth
        from rt_pose_amd.trainer import DataParallelTrainer
        spec = configs.spec("hr3d_lidar")
        trf = DataParallelTrainer("hr3d_lidar", batch, configs.NATIVE_DIMS, total_steps=100, device=dev, use_graph=False)
        trf.load(synth.make_batch(batch, spec["cin"], configs.NATIVE_DIMS, seed=4321, lidar_channels=spec["lidar_channels"]))
        for _ in range(3):
            trf.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nst = 10
        for _ in range(nst):
            trf.step()
        torch.cuda.synchronize()
        msf = (time.perf_counter() - t0) / nst * 1e3
        out["fusion_model"] = {"workload": "hr3d_lidar train step (radar feature ++ %d-channel dense LiDAR grid in front of the towers), "
                                           "%d frames/GPU" % (spec["lidar_channels"], batch),
                               "ms_per_step": round(msf, 3), "frames_per_s": round(batch / msf * 1e3, 1)}
        del trf
    except Exception as e:   # informational leg
        out["fusion_model"] = {"error": repr(e)[:200]}
    return out


def main():
    if len(sys.argv) >= 4 and sys.argv[1] == "--torch-gpu-child":   # child process of the torch_gpu_baseline leg
        for amp in (True, False):
            print("TORCH_GPU " + json.dumps(torch_gpu_child_run(sys.argv[2], int(sys.argv[3]), amp, steps=3 if amp else 2,
                                                                warm=2 if amp else 1)), flush=True)
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="hr3d")
    ap.add_argument("--batch", type=int, default=8, help="frames per GPU")
    ap.add_argument("--graph", action="store_true", help="replay fwd+loss+bwd as one captured HIP graph (eager two-stream "
                    "launching is faster while the step is GPU-bound: DESIGN.md 8)")
    ap.add_argument("--no-graph", action="store_true", help="(default) kept for compatibility")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-torch-gpu", action="store_true", help="skip the PyTorch-eager-on-this-GPU baseline leg")
    ap.add_argument("--no-lidar", action="store_true", help="skip the LiDAR voxelisation (config 5 pieces) leg")
    ap.add_argument("--no-dcn", action="store_true", help="skip the DCN operator (config 4) leg")
    ap.add_argument("--no-forward", action="store_true", help="skip the forward-only (config 2) leg")
    ap.add_argument("--no-other-models", action="store_true", help="skip the driver-timed legs of the other shipped configs")
    ap.add_argument("--dims", default="", help="Z,Y,X of the radar tensor -- TEST RIGS ONLY (tests/test_gpu_bench_two_ranks.py checks this "
                    "file's multi-rank control flow at a small size); the metric is defined at the dataset-native 16,64,160, the default")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # a rank that stops making progress names its own line on stderr and exits (its peers then fail in the collective instead
        # of waiting for it): re-armed at every phase below, so the limit is per phase, not for the whole run
        import datetime
        import faulthandler
        hang_s = float(os.environ.get("RTP_HANG_DUMP_S", "600"))

        def watchdog(phase):
            faulthandler.cancel_dump_traceback_later()
            if phase is not None and hang_s > 0:
                sys.stderr.write("[bench rank %d] %s\n" % (rank, phase))
                sys.stderr.flush()
                faulthandler.dump_traceback_later(hang_s, exit=True)
        watchdog("rendezvous")
        pg_timeout = datetime.timedelta(seconds=max(60.0, hang_s))
        # RTP_BENCH_ONE_DEVICE=1 (test rigs with a single GPU): every rank on cuda:0 over gloo -- exercises this file's N > 1
        # control flow (barriers, max over ranks, rank-0 JSON) where RCCL cannot run; the numbers mean nothing
        if os.environ.get("RTP_BENCH_ONE_DEVICE"):
            local = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo", timeout=pg_timeout)
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=pg_timeout)
    else:
        torch.cuda.set_device(0)

        def watchdog(phase):
            pass
    dev = "cuda:%d" % local
    watchdog("building the trainer")

    from rt_pose_amd import _lib, configs, synth
    from rt_pose_amd.trainer import DataParallelTrainer

    spec = configs.spec(args.model)
    dims = tuple(int(v) for v in args.dims.split(",")) if args.dims else tuple(configs.NATIVE_DIMS)
    native = dims == tuple(configs.NATIVE_DIMS)
    if not native:   # a test rig's size: only the train-step and forward-only control flow, none of the measurement legs
        args.no_roofline = args.no_other_models = args.no_dcn = args.no_lidar = args.no_torch_gpu = args.no_cpu_baseline = True
    tr = DataParallelTrainer(args.model, args.batch, dims, total_steps=max(100, args.steps + args.warmup),
                             device=dev, rank=rank, world_size=world, use_graph=args.graph)
    ex = synth.make_batch(args.batch, spec["cin"], dims, seed=1234, one_hm=spec["heads"]["hm"] == 1, rank=rank,
                          lidar_channels=spec.get("lidar_channels", 0))
    tr.load(ex)  # inputs resident in HBM before the timed region
    torch.cuda.synchronize()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    watchdog("warm-up")
    for _ in range(args.warmup):
        tr.step()
    # THREE consecutive timed segments of exactly --steps steps each, every one bracketed by barrier + synchronize on both sides; the
    # line reports the MEDIAN segment (box-to-box and run-to-run spread of a 0.1-s region is a few percent) and
    # lists all three.  Multi-rank: a segment's time is the slowest rank's.
    seg_s, seg_rank_ms = [], []
    for _seg in range(3):
        watchdog("timed segment %d" % _seg)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            tr.step()
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            # every rank's own clock around the same K steps: the job's time is the slowest rank's; the spread is reported so that a
            # scaling run checks itself (a straggler GPU, or a rank that never joined the collective, shows here)
            ts = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(ts, torch.tensor([el], device=dev, dtype=torch.float64))
            seg_rank_ms.append([round(1e3 * float(x) / args.steps, 3) for x in ts])
            el = max(float(x) for x in ts)
        seg_s.append(el)
    mid = sorted(range(3), key=lambda i: seg_s[i])[1]
    elapsed = seg_s[mid]
    rank_ms = seg_rank_ms[mid] if world > 1 else None
    loss = float(tr.losses()["loss"])
    frames = world * args.batch * args.steps
    g = tr.engine.graph
    train_flops_per_frame = (g.flops["conv_fwd"] + g.flops["conv_dgrad"] + g.flops["wgrad"]) / args.batch

    line = {
        "metric": "radar frames/sec (train) HRRadarPose", "value": round(frames / elapsed, 3), "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "%s train step (fwd+loss+bwd+allreduce+clip+Adam), %d frames/GPU of [%d,%d,%d,%d], "
                               "random-init weights%s" % (args.model, args.batch, spec["cin"], *dims, "" if native else " -- NOT the native shape (test rig)"),
                   "global_batch": world * args.batch, "parallelism": "dp%d" % world, "hip_graph": bool(args.graph),
                   "hw_queues": HW_QUEUES},
        "allreduce_ms": (round(tr.allreduce_ms(), 4) if world > 1 else None),   # flat fp32 gradient all-reduce, events on the step stream
        "allreduce_MB": round(tr.flat.numel * 4 / 1e6, 2),
        "allreduce_buckets": getattr(tr, "ar_buckets", 1),
        "collective": ({"backend": dist.get_backend(), "ranks": dist.get_world_size(),
                        "ms_per_step_by_rank": rank_ms, "rank_spread_ms": round(max(rank_ms) - min(rank_ms), 3)} if world > 1 else None),
        "segments_ms_per_step": [round(1e3 * x / args.steps, 3) for x in seg_s],   # three timed segments of `steps` steps; ms_per_step = their median
        "final_loss": round(loss, 5),
        "train_gflop_per_frame": round(train_flops_per_frame / 1e9, 2),
        "mfma_frac_whole_step": round(frames / elapsed * train_flops_per_frame / (world * PEAK_BF16_TFLOPS * 1e12), 4),
    }

    # BASELINE config 2, reported beside the metric: forward + key-point decode only (the same plan's forward list; in
    # training mode it also writes the statistics the backward needs), same batch, timed after the train steps
    if not args.no_forward:
        watchdog("forward-only leg")
        from rt_pose_amd.engine import PoseEngine
        inf = PoseEngine(tr.be, tr.flat.values, spec["arch"], spec["final_fuse"], spec["heads"], spec["weight"],
                         spec["code_weights"], args.batch, dims, train=False, test_cfg=configs.test_cfg())
        with tr._on_stream():
            inf.load_input(ex["rdr"]["rdr_tensor"])
            for _ in range(3):
                inf.run_forward()
                inf.run_decode()
        barrier()
        t1 = time.perf_counter()
        nf = max(10, args.steps)
        for _ in range(nf):
            with tr._on_stream():
                inf.run_forward()
                inf.run_decode()
        barrier()
        el_f = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([el_f], device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el_f = float(t)
        line["forward_only"] = {"workload": "%s backbone + head forward + key-point decode (inference plan), %d frames/GPU, bf16 "
                                            "(BASELINE config 2)" % (args.model, args.batch),
                                "value": round(world * args.batch * nf / el_f, 2), "unit": "frames/s",
                                "ms_per_batch": round(1e3 * el_f / nf, 3),
                                "mfma_frac": round(world * args.batch * nf / el_f * g.flops["conv_fwd"] / args.batch
                                                   / (world * PEAK_BF16_TFLOPS * 1e12), 4)}
        del inf

    if world == 1 and not args.no_roofline:
        be = tr.be
        # the LDS-tiled conv is timed in two groups: its dominant geometry (32 -> 32 channels, full resolution) and the rest
        # (level-1 tensors, 16-channel head outputs); graph.py's conv_tiled totals include both
        rest_f = g.flops["conv_tiled"] - g.flops["conv_tiled_full"]
        rest_b = g.alg_bytes["conv_tiled"] - g.alg_bytes["conv_tiled_full"]
        fb_f, fb_b = g.flops["conv_tiled_full_bwd"], g.alg_bytes["conv_tiled_full_bwd"]
        fams = {"conv_tiled_kernel<2,...> 32->32ch 3x3x3 at full resolution (fwd+dgrad)":
                    ((_lib.FAM_CONV_TILED_FULL, _lib.FAM_CONV_TILED_FULL_BWD), g.flops["conv_tiled_full"], g.alg_bytes["conv_tiled_full"]),
                # the same launches split: forward (class bias / residual / ReLU / statistics epilogue, GroupNorm fold prologue) and
                # data gradient (most of them the fused variants that also do the gradient fan-in and GroupNorm-backward apply)
                "  of which forward": (_lib.FAM_CONV_TILED_FULL, g.flops["conv_tiled_full"] - fb_f, g.alg_bytes["conv_tiled_full"] - fb_b),
                "  of which data gradient": (_lib.FAM_CONV_TILED_FULL_BWD, fb_f, fb_b),
                "conv_tiled other geometries (level 1, 16-channel outputs)": (_lib.FAM_CONV_TILED, rest_f, rest_b),
                "conv64_kernel 64-wide 3x3x3 (64->64 layers, paired head towers; fwd+dgrad)": (_lib.FAM_CONV64, g.flops["conv64"], g.alg_bytes["conv64"]),
                "conv_igemm generic (fwd+dgrad)": (_lib.FAM_CONV, g.flops["conv_generic"], g.alg_bytes["conv_generic"]),
                "wgrad_tiled (32ch 3x3x3)": (_lib.FAM_WGRAD_TILED, g.flops["wgrad_tiled"], g.alg_bytes["wgrad_tiled"]),
                "wgrad generic": (_lib.FAM_WGRAD, g.flops["wgrad_generic"], g.alg_bytes["wgrad_generic"])}
        tr.use_graph = False
        tr.engine.use_lanes = False  # one stream while timing kernels: concurrent side-stream work would inflate the events
        all_fams = sorted({f for fam, _, _ in fams.values() for f in (fam if isinstance(fam, tuple) else (fam,))})
        for fam in all_fams:
            be.prof_enable(fam, True)
        ksteps = min(args.steps, 5)
        for _ in range(ksteps):
            tr.step()
        torch.cuda.synchronize()
        best = None
        detail = {}
        collected = {}
        for fam in all_fams:
            collected[fam] = be.prof_collect(fam)
            be.prof_enable(fam, False)
        for kname, (fam, flops, nbytes) in fams.items():
            parts = [collected[f] for f in (fam if isinstance(fam, tuple) else (fam,))]
            ms, cnt = sum(p_[0] for p_ in parts), sum(p_[1] for p_ in parts)
            if cnt == 0:
                continue
            tf = flops * ksteps / (ms * 1e-3) / 1e12
            gbps = nbytes * ksteps / (ms * 1e-3) / 1e9
            detail[kname] = {"launches_per_step": cnt // ksteps, "ms_per_step": round(ms / ksteps, 3),
                             "avg_us_per_launch": round(1e3 * ms / cnt, 2), "tflops": round(tf, 2),
                             "algorithmic_GBps": round(gbps, 1), "hbm_frac": round(gbps / PEAK_HBM_GBPS, 4)}
            if not kname.startswith("  ") and (best is None or ms > best[1]):
                best = (kname, ms, tf, gbps)
        # HBM bytes per launch of the dominant kernel: PMC counters cannot be read inside this process, so the separate
        # rocprofv3 --pmc passes (tools/pmc_tiled.sh: FETCH_SIZE doubled per MI355X_MICROARCH.md, + WRITE_SIZE, B=8
        # full-resolution layer) write profiles/pmc_traffic.json and this line quotes that artefact (null when absent)
        traffic, traffic_src = None, None
        tj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tj):
            try:
                with open(tj) as f:
                    tdata = json.load(f)
                key = "conv_tiled_full" if best[0].startswith("conv_tiled_kernel") else "wgrad_tiled" if best[0].startswith("wgrad_tiled") else None
                if key in tdata.get("bytes_per_launch", {}):
                    traffic = tdata["bytes_per_launch"][key]
                    traffic_src = "profiles/pmc_traffic.json, recorded %s (%s)" % (tdata.get("recorded", "round 2"), tdata.get("source", ""))
            except Exception:
                pass
        # The 32-channel 3x3x3 layers sit at the ridge (288-431 algorithmic flop/B against 2500/8 = 312): the MFMA bound is
        # the one SURVEY 8d names; the same launches against the HBM roof are reported beside it.
        line["roofline"] = {"bound": "mfma", "kernel": best[0], "achieved": round(best[2], 2), "peak": PEAK_BF16_TFLOPS,
                            "unit": "TFLOP/s", "frac": round(best[2] / PEAK_BF16_TFLOPS, 4),
                            "traffic": traffic if args.batch == 8 and args.model == "hr3d" else None,
                            "traffic_source": traffic_src,
                            "hbm_view": {"achieved": round(best[3], 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                                         "frac": round(best[3] / PEAK_HBM_GBPS, 4)},
                            "families": detail}
        # ... and the same family at FULL WIDTH: the plan's width hints give the main lane's launches 192 of 256 workgroups so that the
        # side lanes' chains run beside them -- in the single-stream timing above nothing runs beside them, so the hints
        # only cost.  A second plan built without hints, same kernels, same events, 3 single-stream steps: what the kernel does with
        # the chip to itself.  Reported beside `frac` (which stays the figure of the plan as shipped).
        try:
            from rt_pose_amd.options import PlanOptions
            opt = PlanOptions.from_env()
            opt.width_hints = ""
            ftr = DataParallelTrainer(args.model, args.batch, dims, total_steps=100, device=dev, use_graph=False, backend=be,
                                      stream=tr.stream, options=opt)
            ftr.load(ex)
            ftr.engine.use_lanes = False
            for _ in range(2):
                ftr.step()
            torch.cuda.synchronize()
            fw = (_lib.FAM_CONV_TILED_FULL, _lib.FAM_CONV_TILED_FULL_BWD)
            for fam in fw:
                be.prof_collect(fam)          # (drop what the warm-up steps recorded)
                be.prof_enable(fam, True)
            for _ in range(3):
                ftr.step()
            torch.cuda.synchronize()
            parts = [be.prof_collect(f) for f in fw]
            for fam in fw:
                be.prof_enable(fam, False)
            ms, cnt = sum(p_[0] for p_ in parts), sum(p_[1] for p_ in parts)
            fg = ftr.engine.graph
            if cnt:
                tf = fg.flops["conv_tiled_full"] * 3 / (ms * 1e-3) / 1e12
                line["roofline"]["full_width"] = {"what": "the same launches from a plan WITHOUT width hints (256 workgroups), single stream",
                                                  "avg_us_per_launch": round(1e3 * ms / cnt, 2), "tflops": round(tf, 2),
                                                  "frac": round(tf / PEAK_BF16_TFLOPS, 4)}
            del ftr
            torch.cuda.empty_cache()
        except Exception as e:   # informational
            line["roofline"]["full_width"] = {"error": repr(e)[:200]}
    # The other shipped configs (configs/cruw_pose/hr3d_one_hm_doppler*.py) and the DCN-head variant of the headline model (BASELINE
    # config 4), driver-timed in the same run: 3 warm-up + 10 timed train steps each at the same 8 frames per GPU (their plans are
    # built beside the headline model's)
    if world == 1 and not args.no_other_models and args.model == "hr3d":
        line["other_models"] = {}
        for oname in ("hr3d_one_hm_doppler", "hr3d_one_hm_doppler_phase", "hr3d_dcn"):
            ospec = configs.spec(oname)
            # (same backend object and step stream as the headline model: the lanes keep their streams / hardware queues)
            otr = DataParallelTrainer(oname, args.batch, configs.NATIVE_DIMS, total_steps=100, device=dev, use_graph=False,
                                      backend=tr.be, stream=tr.stream)
            otr.load(synth.make_batch(args.batch, ospec["cin"], configs.NATIVE_DIMS, seed=1234, one_hm=ospec["heads"]["hm"] == 1))
            for _ in range(3):
                otr.step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(10):
                otr.step()
            torch.cuda.synchronize()
            el_o = time.perf_counter() - t1
            og = otr.engine.graph
            ofl = (og.flops["conv_fwd"] + og.flops["conv_dgrad"] + og.flops["wgrad"]) / args.batch
            line["other_models"][oname] = {
                "value": round(args.batch * 10 / el_o, 2), "unit": "frames/s", "ms_per_step": round(1e2 * el_o, 3), "steps": 10,
                "frames_per_gpu": args.batch, "input": "[%d,%d,16,64,160]" % (args.batch, ospec["cin"]),
                "train_gflop_per_frame": round(ofl / 1e9, 2),
                "mfma_frac_whole_step": round(args.batch * 10 / el_o * ofl / (PEAK_BF16_TFLOPS * 1e12), 4),
                "final_loss": round(float(otr.losses()["loss"]), 5)}
            del otr
            torch.cuda.empty_cache()
    # The headline model at other frames-per-GPU counts (the reference's own hr3d batch is 16 per GPU, configs/cruw_pose/hr3d.py:7): the
    # same plan rules (four-stream lane map, width hints, shared head launches) for every batch; 3 warm-up + 10 timed steps each
    if world == 1 and not args.no_other_models and args.model == "hr3d" and args.batch == 8:
        line["other_batches"] = {}
        for ob in (4, 16):
            otr = DataParallelTrainer("hr3d", ob, configs.NATIVE_DIMS, total_steps=100, device=dev, use_graph=False, backend=tr.be,
                                      stream=tr.stream)
            otr.load(synth.make_batch(ob, spec["cin"], configs.NATIVE_DIMS, seed=1234, one_hm=False))
            for _ in range(3):
                otr.step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(10):
                otr.step()
            torch.cuda.synchronize()
            el_o = time.perf_counter() - t1
            line["other_batches"][str(ob)] = {"value": round(ob * 10 / el_o, 2), "unit": "frames/s", "ms_per_step": round(1e2 * el_o, 3),
                                              "steps": 10, "frames_per_gpu": ob,
                                              "mfma_frac_whole_step": round(ob * 10 / el_o * train_flops_per_frame / (PEAK_BF16_TFLOPS * 1e12), 4),
                                              "final_loss": round(float(otr.losses()["loss"]), 5)}
            del otr
            torch.cuda.empty_cache()
    # BASELINE config 4, op level (the reference's DCN head cannot run on its own 5-D feature): DCNv1 3x3,
    # deformable_groups 4, im2col_step 64 on the level-0 feature with Z folded into the batch, [B*16, 32, 64, 160] fp32
    if world == 1 and not args.no_dcn:
        line["dcn_op"] = dcn_op_bench(dev, args.batch)
    # BASELINE config 5, the pieces the reference defines: extrinsic transform + dynamic voxelisation +
    # dense scatter of one 200 000-point LiDAR frame per radar frame
    if world == 1 and not args.no_lidar:
        line["lidar_stream"] = lidar_bench(dev, args.batch)
    # The reference-style PyTorch path on this same GPU (the oracle with its tensors on the device: stock ATen / MIOpen
    # kernels, eager autograd -- what the reference itself would execute here), in a child process with a time limit so
    # that nothing it does can touch this process; informational like cpu_baseline.
    if world == 1 and not args.no_torch_gpu:
        line["torch_gpu_baseline"] = torch_gpu_baseline(args.model, args.batch)
    # MPJPE proxy: an ARTEFACT of tests/keypoint_agreement.py (run on an MI355X with the oracle as the checker, which this
    # process may only use for cpu_baseline) -- quoted with its file name, not measured by this run
    ka_files = {"hr3d": "r05_keypoint_agreement_hr3d.json", "hr3d_one_hm_doppler": "r05_keypoint_agreement_doppler.json"}
    kj = os.path.join(ROOT, "profiles", ka_files.get(args.model, "-"))   # (keyed on the exact model; omitted when none was recorded for it)
    if world == 1 and os.path.exists(kj):
        try:
            with open(kj) as f:
                ka = json.load(f)
            ka.pop("per_seed", None)
            line["keypoint_agreement_artefact"] = dict(ka, source="profiles/" + os.path.basename(kj))
        except Exception:
            pass
    # ... and its round-5 companions (artefacts as well, quoted by file name): the same comparison for the Doppler configuration, and
    # the TRAJECTORY check (tests/trajectory_check.py: the same seeds trained by the HIP bf16 step and by the oracle's fp32 step at
    # reduced dims, each model decoding the same held-out frames with its own forward)
    if world == 1:
        extra = {}
        for key, kname in (("keypoint_agreement_doppler", "r05_keypoint_agreement_doppler.json"),
                           ("trajectory_hr3d", "r05_trajectory_hr3d.json"), ("trajectory_doppler", "r05_trajectory_doppler.json")):
            kj = os.path.join(ROOT, "profiles", kname)
            if os.path.exists(kj):
                try:
                    with open(kj) as f:
                        ka = json.load(f)
                    extra[key] = dict({k: ka[k] for k in ("model", "dims", "train_steps", "seeds", "frames", "mpjpe_cm", "abs_mpjpe_cm",
                                                          "argmax_agreement", "argmax_within_1_voxel", "worst_seed_abs_mpjpe_delta_cm") if k in ka},
                                      source="profiles/" + kname)
                except Exception:
                    pass
        if extra:
            line["numerics_artefacts"] = extra
    # whole-step HBM traffic from the separate PMC passes (tools/pmc_step.sh), against the fused-minimum algorithmic bytes
    pj = next((q for q in (os.path.join(ROOT, "profiles", f) for f in ("r06_pmc_step_traffic.json", "r05_pmc_step_traffic.json", "r04_pmc_step_traffic.json")) if os.path.exists(q)),
              os.path.join(ROOT, "profiles", "r06_pmc_step_traffic.json"))
    if world == 1 and "roofline" in line and os.path.exists(pj) and args.model == "hr3d" and args.batch == 8:
        try:
            with open(pj) as f:
                pt = json.load(f)
            fused_min = 3 * 553e6 * args.batch   # 553 MB per frame and pass, three passes
            line["roofline"]["whole_step_traffic_GB"] = round(pt["bytes_per_step"] / 1e9, 2)
            line["roofline"]["whole_step_traffic_ratio"] = round(pt["bytes_per_step"] / fused_min, 3)
            line["roofline"]["whole_step_traffic_source"] = "profiles/%s (recorded %s)" % (os.path.basename(pj), pt.get("recorded", "?"))
        except Exception:
            pass
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(args.model)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        watchdog("shutdown")
        dist.destroy_process_group()
        watchdog(None)


if __name__ == "__main__":
    main()
