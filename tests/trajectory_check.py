"""Training-trajectory check: does TRAINING in bf16 on the HIP kernels land where fp32 training lands?

tests/keypoint_agreement.py compares two decodes of the SAME weights.  Round 4's gate measurement showed that two correct bf16
evaluations of one plan differ by up to ~20 % on layer-1 gradient tensors of the Doppler configs, so the question left open was
whether a whole bf16 TRAINING RUN of such a config still reaches the fp32 run's accuracy.  Here, at reduced dims, the same seeded
initial weights are trained on the same seeded batches
    (a) by the product path: DataParallelTrainer (HIP kernels, bf16 activations / gradients, fp32 master weights), and
    (b) by the oracle: oracle/hrradarpose_ref.py forward + autograd in fp32 with the restated optimiser rule (AdamTrueWD + one_cycle,
        clip 35) -- plain torch code, its tensors on the GPU so that hundreds of steps take minutes,
and each model then decodes the same held-out synthetic frames with ITS OWN forward (HIP inference plan / oracle fp32 forward).
Reported: root-relative MPJPE and absolute MPJPE of both against the synthetic ground truth (rt_pose_amd.evaluate.pjpe / abs_pjpe =
eval_util.py:5-10), per seed and pooled.  North star: MPJPE within 0.5 cm of the reference's.

    python -m tests.trajectory_check --model hr3d_one_hm_doppler --steps 400 --seeds 2 --out gpurun_out/trajectory.json
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import hrradarpose_ref as O  # noqa: E402
from rt_pose_amd import configs  # noqa: E402
from tests.keypoint_agreement import decoded_points, make_pose_batch  # noqa: E402


def to_dev(v, dev):
    if torch.is_tensor(v):
        return v.to(dev)
    if isinstance(v, dict):
        return {k: to_dev(x, dev) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return type(v)(to_dev(x, dev) for x in v)
    return v


def run(name="hr3d", steps=300, batch=8, eval_batches=4, seed=0, dims=(8, 32, 80), dev="cuda:0", log=None):
    from rt_pose_amd.engine import PoseEngine, one_cycle
    from rt_pose_amd.evaluate import abs_pjpe, pjpe
    from rt_pose_amd.trainer import DataParallelTrainer, init_state_dict
    spec = configs.spec(name)
    tcfg = configs.test_cfg()
    arch, fin, fout, fuse, heads, weight, cw = O.MODEL_CONFIGS[name]
    pool = [make_pose_batch(batch, dims, 10_000 + i, name)[0] for i in range(min(steps, 40))]
    held = [make_pose_batch(batch, dims, 900_000 + i, name) for i in range(eval_batches)]
    # ---- (a) the product path
    t0 = time.time()
    tr = DataParallelTrainer(name, batch, dims, total_steps=steps, device=dev, use_graph=False, seed=seed)
    hist_h = []
    for it in range(steps):
        tr.step(pool[it % len(pool)])
        if it % 50 == 0 or it == steps - 1:
            hist_h.append((it, float(tr.losses()["loss"])))
            if log:
                log("hip    step %d loss %.4f" % hist_h[-1])
    torch.cuda.synchronize()
    t_hip = time.time() - t0
    inf = PoseEngine(tr.be, tr.flat.values, spec["arch"], spec["final_fuse"], spec["heads"], spec["weight"], spec["code_weights"], batch,
                     dims, train=False, test_cfg=tcfg)
    # ---- (b) the oracle, fp32, same initial weights (trainer.init_state_dict(seed)), same batches, same schedule
    t0 = time.time()
    sd = {k: v.to(dev).requires_grad_(True) for k, v in init_state_dict(configs.param_shapes(name), seed).items()}
    opt = O.AdamTrueWD(list(sd.values()))
    pool_d = [to_dev(ex, dev) for ex in pool]
    hist_o = []
    for it in range(steps):
        for p in sd.values():
            p.grad = None
        loss = O.radar_pose_net(sd, pool_d[it % len(pool_d)], fuse, weight, cw)["loss"][0]
        loss.backward()
        lr, b1 = one_cycle(it, steps, spec["lr_max"])
        opt.step(lr, b1)
        if it % 50 == 0 or it == steps - 1:
            hist_o.append((it, float(loss)))
            if log:
                log("oracle step %d loss %.4f" % hist_o[-1])
    torch.cuda.synchronize()
    t_orc = time.time() - t0
    sdd = {k: v.detach() for k, v in sd.items()}
    # ---- held-out frames, each model's own forward and decode
    e_h, e_o, a_h, a_o, s_h, s_o = [], [], [], [], [], []
    for ex, gt in held:
        with tr._on_stream():
            inf.load_input(ex["rdr"]["rdr_tensor"])
            inf.run_forward()
            inf.run_decode()
        torch.cuda.synchronize()
        kh = inf.keypoints()
        with torch.no_grad():
            preds, _ = O.center_head(sdd, O.hrnet3d(sdd, ex["rdr"]["rdr_tensor"].to(dev), fuse))
            preds = [{k: v.float().cpu() for k, v in preds[0].items()}]
        ko = O.center_head_predict(preds, tcfg)
        for b in range(batch):
            (ph, sh), (po, so) = decoded_points(kh[b]["keypoints"]), decoded_points(ko[b]["keypoints"])
            assert ph.shape == (15, 3) and po.shape == (15, 3)
            e_h.append(pjpe(ph, gt[b])); e_o.append(pjpe(po, gt[b]))
            a_h.append(abs_pjpe(ph, gt[b])); a_o.append(abs_pjpe(po, gt[b]))
            s_h.append(sh); s_o.append(so)
    cm = lambda v: round(float(np.mean(v)) * 100, 4)
    return {"model": name, "dims": list(dims), "seed": seed, "train_steps": steps, "train_batch": batch, "frames": len(e_h),
            "loss_history": {"hip_bf16": [[i, round(l, 4)] for i, l in hist_h], "oracle_fp32": [[i, round(l, 4)] for i, l in hist_o]},
            "mpjpe_cm": {"hip_bf16_trained": cm(e_h), "oracle_fp32_trained": cm(e_o), "delta": round(cm(e_h) - cm(e_o), 4)},
            "abs_mpjpe_cm": {"hip_bf16_trained": cm(a_h), "oracle_fp32_trained": cm(a_o), "delta": round(cm(a_h) - cm(a_o), 4)},
            "mean_peak_score": {"hip_bf16_trained": round(float(np.mean(s_h)), 4), "oracle_fp32_trained": round(float(np.mean(s_o)), 4)},
            "train_seconds": {"hip": round(t_hip, 1), "oracle_on_gpu": round(t_orc, 1)}}


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="hr3d")
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--seeds", type=int, default=2)
    ap.add_argument("--eval-batches", type=int, default=4)
    ap.add_argument("--dims", default="8,32,80")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "trajectory_check.json"))
    a = ap.parse_args()
    dims = tuple(int(v) for v in a.dims.split(","))
    runs = [run(a.model, a.steps, 8, a.eval_batches, sd, dims, log=lambda s: print(s, flush=True)) for sd in range(a.seeds)]
    w = np.array([r["frames"] for r in runs], dtype=np.float64)
    pooled = lambda key, k: round(float((np.array([r[key][k] for r in runs]) * w).sum() / w.sum()), 4)
    res = {"what": "same seeded weights and batches trained by the HIP bf16 step and by the oracle's fp32 step; each model decodes the same "
                   "held-out synthetic frames with its own forward", "model": a.model, "dims": list(dims), "train_steps": a.steps,
           "seeds": a.seeds, "frames": int(w.sum()),
           "mpjpe_cm": {k: pooled("mpjpe_cm", k) for k in ("hip_bf16_trained", "oracle_fp32_trained", "delta")},
           "abs_mpjpe_cm": {k: pooled("abs_mpjpe_cm", k) for k in ("hip_bf16_trained", "oracle_fp32_trained", "delta")},
           "worst_seed_abs_mpjpe_delta_cm": max(abs(r["mpjpe_cm"]["delta"]) for r in runs),
           "budget": "north star: MPJPE within 0.5 cm of the reference", "per_seed": runs}
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "per_seed"}))


if __name__ == "__main__":
    main()
