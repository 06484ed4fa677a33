"""Key-point agreement harness (MPJPE proxy): does the bf16 HIP path decode the same key-points as the reference's fp32
forward ON THE SAME WEIGHTS?  (SURVEY.md 7 last bullet, 8d metric row: "argmax agreement ... decoded key-point shift <= 1
voxel"; reference decode: det3d/models/pose_heads/center_head.py:272-360, metric: eval_util.py:5-10.)

No RT-Pose data exists here, so the harness manufactures a learnable task of the dataset's native shape: one person per
frame, 15 joints on a fixed skeleton (+ jitter) around a random pelvis; the radar tensor is the usual clamped noise plus a
trilinear deposit of a joint-specific amplitude at every joint's continuous position.  The product trainer
(DataParallelTrainer, HIP kernels, bf16) trains ~300 steps on fresh batches so that the heat-maps really peak; then, on
held-out frames and the SAME weights,
    HIP   : inference plan forward + rtp_decode (bf16 activations)
    oracle: oracle/hrradarpose_ref.py fp32 forward + center_head_predict (CPU)
are compared: argmax-voxel agreement rate, metric shift of the decoded key-points, and MPJPE of both against the
synthetic ground truth (so the number the north star cares about -- the MPJPE DIFFERENCE caused by bf16 -- is measured).

This module lives under tests/ because it uses the oracle as the checker.  `python -m tests.keypoint_agreement` writes
gpurun_out/keypoint_agreement.json; copied to profiles/r03_keypoint_agreement.json it is what bench.py quotes as
`keypoint_agreement_artefact` (bench.py does not import this module).
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
from rt_pose_amd import configs, synth  # noqa: E402

# skeleton offsets from the pelvis in METRES (x range, y lateral, z height); joint order = configs.JOINTS
SKELETON = np.array([[0, 0, 0], [0, -0.12, -0.05], [0.02, -0.13, -0.48], [0.0, -0.13, -0.90], [0, 0.12, -0.05],
                     [0.02, 0.13, -0.48], [0.0, 0.13, -0.90], [0, 0, 0.45], [0.02, 0, 0.70], [0, 0.20, 0.42],
                     [0.05, 0.30, 0.15], [0.15, 0.33, -0.08], [0, -0.20, 0.42], [0.05, -0.30, 0.15], [0.15, -0.33, -0.08]])


def make_pose_batch(batch, dims, seed, name="hr3d"):
    """-> (example dict like the reference's collate_fn output, gt metric key-points [B,15,3]).
    name: a 15-heat-map config (hr3d: one heat-map + 3 offsets per joint, pose.py:206-254) or a one-heat-map config
    (hr3d_one_hm*: ONE heat-map at the pelvis voxel, radius 2, and 45 regressed coordinates = every joint's continuous voxel
    position relative to that voxel in (x, y, z) order, pose.py:407-451 / center_head.py:348-355).  Doppler configs (Cin = 32 / 64):
    every joint's deposit goes into a joint-specific channel on top of noise in all channels."""
    Z, Y, X = dims
    spec = configs.spec(name)
    cin, one_hm = spec["cin"], spec["heads"]["hm"] == 1
    rng = np.random.default_rng(seed)
    vs = np.array(configs.VOXEL_SIZE)                       # x, y, z
    org = np.array(configs.test_cfg()["pc_range"])          # x, y, z minimum
    g = torch.Generator().manual_seed(seed)
    rdr = torch.relu(torch.randn(batch, cin, Z, Y, X, generator=g) * 0.5 + 0.1)
    ncls, m = (1, 1) if one_hm else (15, 15)
    hm = torch.zeros(batch, ncls, Z, Y, X)
    ind = torch.zeros(batch, m, dtype=torch.int64)
    anno = torch.zeros(batch, m, 45 if one_hm else 3)
    gt = np.zeros((batch, 15, 3))
    prof = synth.splat_profile(2 if one_hm else 1)
    ext = np.array([X, Y, Z]) * vs
    for b in range(batch):
        pelvis = org + np.array([rng.uniform(0.25, 0.75), rng.uniform(0.2, 0.8), rng.uniform(0.35, 0.55)]) * ext
        pts = pelvis + SKELETON * rng.uniform(0.9, 1.1) + rng.normal(0, 0.03, size=(15, 3))
        gt[b] = pts
        ci0 = None
        for j in range(15):
            c = (pts[j] - org) / vs                          # continuous voxel coordinate (x, y, z)
            ci = np.floor(c).astype(int)
            ci = np.clip(ci, 0, [X - 1, Y - 1, Z - 1])
            x, y, z = int(ci[0]), int(ci[1]), int(ci[2])
            if one_hm:
                if j == 0:                                   # the one heat-map sits at the root joint's voxel
                    ci0 = ci
                    synth.draw_splat(hm[b, 0], z, y, x, 2, prof)
                    ind[b, 0] = (z * Y + y) * X + x
                anno[b, 0, 3 * j:3 * j + 3] = torch.tensor(c - ci0, dtype=torch.float32)
            else:
                synth.draw_splat(hm[b, j], z, y, x, 1, prof)
                ind[b, j] = (z * Y + y) * X + x
                anno[b, j] = torch.tensor(c - ci, dtype=torch.float32)
            # trilinear deposit of a joint-specific amplitude at the voxel CENTRE convention (voxel k covers [k, k+1))
            f = c - 0.5
            f0 = np.floor(f).astype(int)
            w = f - f0
            amp = 2.0 + 0.25 * j
            ch = (2 * j + 1) % cin                           # (Cin = 1: channel 0)
            for dz in (0, 1):
                for dy in (0, 1):
                    for dx in (0, 1):
                        xx, yy, zz = f0[0] + dx, f0[1] + dy, f0[2] + dz
                        if 0 <= xx < X and 0 <= yy < Y and 0 <= zz < Z:
                            rdr[b, ch, zz, yy, xx] += amp * (w[0] if dx else 1 - w[0]) * (w[1] if dy else 1 - w[1]) * (w[2] if dz else 1 - w[2])
    ex = dict(rdr_tensor=rdr, hm=[hm], ind=[ind], mask=[torch.ones(batch, m, dtype=torch.uint8)],
              cat=[torch.zeros(batch, 1, dtype=torch.int64) if one_hm else torch.arange(15).repeat(batch, 1)], anno_pose=[anno])
    return {"rdr": ex, "meta": [{"seq": "synth", "frame": b, "rdr_frame": b} for b in range(batch)]}, gt


def decoded_points(kps):
    """[(id, x, y, z, score) ...] of one frame -> ([15, 3] metric points, [scores])."""
    return np.array([k[1:4] for k in kps]), [k[4] for k in kps]


def run(steps=300, batch=8, eval_batches=2, seed=0, dims=configs.NATIVE_DIMS, name="hr3d", log=None, oracle_device="cuda:0"):
    """oracle_device: where the oracle's fp32 forward of the held-out frames runs (plain torch functional code; "cpu" or the GPU --
    the same fp32 arithmetic through ATen's device kernels, ~20 x faster at the native shape)."""
    from rt_pose_amd.engine import PoseEngine
    from rt_pose_amd.evaluate import abs_pjpe, pjpe
    from rt_pose_amd.trainer import DataParallelTrainer
    spec = configs.spec(name)
    tcfg = configs.test_cfg()
    tr = DataParallelTrainer(name, batch, dims, total_steps=steps, device="cuda:0", use_graph=False, seed=seed)
    t0 = time.time()
    hist = []
    pool = [make_pose_batch(batch, dims, 10_000 + i, name)[0] for i in range(min(steps, 40))]   # 320 distinct training frames, cycled
    for it in range(steps):
        tr.step(pool[it % len(pool)])
        if it % 25 == 0 or it == steps - 1:
            hist.append((it, float(tr.losses()["loss"])))
            if log:
                log("step %d loss %.4f" % hist[-1])
    torch.cuda.synchronize()
    train_s = time.time() - t0
    inf = PoseEngine(tr.be, tr.flat.values, spec["arch"], spec["final_fuse"], spec["heads"], spec["weight"],
                     spec["code_weights"], batch, dims, train=False, test_cfg=tcfg)
    sd = {k: v.detach().float().to(oracle_device) for k, v in tr.flat.values.items()}
    arch, fin, fout, fuse, heads, weight, cw = O.MODEL_CONFIGS[name]
    ncls = heads["hm"]
    vs = np.array(configs.VOXEL_SIZE)
    agree, shifts, e_hip, e_ref, ea_hip, ea_ref, sc_hip, sc_ref, vox_d = [], [], [], [], [], [], [], [], []
    for eb in range(eval_batches):
        ex, gt = make_pose_batch(batch, dims, 900_000 + eb, name)
        with tr._on_stream():
            inf.load_input(ex["rdr"]["rdr_tensor"])
            inf.run_forward()
            inf.run_decode()
        torch.cuda.synchronize()
        kh = inf.keypoints()
        hm_hip = inf.output("hm").float().cpu()
        with torch.no_grad():
            preds, _ = O.center_head(sd, O.hrnet3d(sd, ex["rdr"]["rdr_tensor"].to(oracle_device), fuse))
            preds = [{k: v.float().cpu() for k, v in preds[0].items()}]
        kr = O.center_head_predict(preds, tcfg)
        hm_ref = preds[0]["hm"]
        for b in range(batch):
            a_h = hm_hip[b].reshape(ncls, -1).argmax(1).numpy()
            a_r = hm_ref[b].reshape(ncls, -1).argmax(1).numpy()
            agree.append(a_h == a_r)
            Z, Y, X = dims
            dz, dy, dx = a_h // (Y * X) - a_r // (Y * X), (a_h // X) % Y - (a_r // X) % Y, a_h % X - a_r % X
            vox_d.append(np.maximum(np.maximum(np.abs(dz), np.abs(dy)), np.abs(dx)))
            (ph, sh), (pr, sr) = decoded_points(kh[b]["keypoints"]), decoded_points(kr[b]["keypoints"])
            assert ph.shape == (15, 3) and pr.shape == (15, 3), "score threshold 0 keeps every joint"
            shifts.append(np.linalg.norm(ph - pr, axis=1))
            e_hip.append(pjpe(ph, gt[b])); e_ref.append(pjpe(pr, gt[b]))
            ea_hip.append(abs_pjpe(ph, gt[b])); ea_ref.append(abs_pjpe(pr, gt[b]))
            sc_hip.append(sh); sc_ref.append(sr)
    agree, shifts, vox_d = np.array(agree), np.array(shifts), np.array(vox_d)
    mp_h, mp_r = float(np.mean(e_hip)) * 100, float(np.mean(e_ref)) * 100
    ma_h, ma_r = float(np.mean(ea_hip)) * 100, float(np.mean(ea_ref)) * 100
    return {
        "what": "HIP bf16 inference vs oracle fp32 forward on the same trained weights, decoded key-points on held-out synthetic frames",
        "model": name, "dims": list(dims), "train_steps": steps, "train_batch": batch, "train_seconds": round(train_s, 1),
        "loss_history": [[i, round(l, 4)] for i, l in hist],
        "frames": int(agree.shape[0]), "joints": int(agree.size),
        "argmax_agreement": round(float(agree.mean()), 4),
        "root_argmax_disagreements": int((~agree[:, 0]).sum()),   # frames whose ROOT joint (pelvis) decodes to another voxel: MPJPE is
                                                                  # root-relative, so each of them shifts all 14 other joints of its frame
        "argmax_max_voxel_distance": int(vox_d.max()),
        "argmax_within_1_voxel": round(float((vox_d <= 1).mean()), 4),
        "keypoint_shift_cm": {"mean": round(float(shifts.mean()) * 100, 4), "p95": round(float(np.percentile(shifts, 95)) * 100, 4),
                              "max": round(float(shifts.max()) * 100, 4)},
        "voxel_size_cm": [round(v * 100, 3) for v in vs],
        "mpjpe_cm": {"hip_bf16": round(mp_h, 4), "oracle_fp32": round(mp_r, 4), "delta": round(mp_h - mp_r, 4)},
        "abs_mpjpe_cm": {"hip_bf16": round(ma_h, 4), "oracle_fp32": round(ma_r, 4), "delta": round(ma_h - ma_r, 4)},
        "mean_peak_score": {"hip_bf16": round(float(np.mean(sc_hip)), 4), "oracle_fp32": round(float(np.mean(sc_ref)), 4)},
        "budget": "north star: MPJPE within 0.5 cm of the reference; decoded key-point shift <= 1 voxel",
    }


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--eval-batches", type=int, default=2)
    ap.add_argument("--model", default="hr3d", help="hr3d | hr3d_one_hm | hr3d_one_hm_doppler (the label layout follows the config's head)")
    ap.add_argument("--seeds", type=int, default=1, help="independent trainings (weights seed 0..n-1); the artefact carries every run and their pooled figures")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "keypoint_agreement.json"),
                    help="(gpurun_out/ is the directory that travels back from the GPU box; copy the file to profiles/)")
    a = ap.parse_args()
    runs = [run(steps=a.steps, eval_batches=a.eval_batches, seed=sd, name=a.model, log=lambda s: print(s, flush=True)) for sd in range(a.seeds)]
    res = dict(runs[0])
    if len(runs) > 1:   # pooled over the seeds: frame-weighted means, worst-case maxima
        w = np.array([r["frames"] for r in runs], dtype=np.float64)
        wm = lambda f: float((np.array([f(r) for r in runs]) * w).sum() / w.sum())
        res.update({
            "seeds": len(runs), "frames": int(w.sum()), "joints": int(sum(r["joints"] for r in runs)),
            "root_argmax_disagreements": int(sum(r.get("root_argmax_disagreements", 0) for r in runs)),
            "argmax_agreement": round(wm(lambda r: r["argmax_agreement"]), 4),
            "argmax_within_1_voxel": round(wm(lambda r: r["argmax_within_1_voxel"]), 4),
            "argmax_max_voxel_distance": max(r["argmax_max_voxel_distance"] for r in runs),
            "keypoint_shift_cm": {"mean": round(wm(lambda r: r["keypoint_shift_cm"]["mean"]), 4),
                                  "p95": max(r["keypoint_shift_cm"]["p95"] for r in runs), "max": max(r["keypoint_shift_cm"]["max"] for r in runs)},
            "mpjpe_cm": {k: round(wm(lambda r: r["mpjpe_cm"][k]), 4) for k in ("hip_bf16", "oracle_fp32", "delta")},
            "abs_mpjpe_cm": {k: round(wm(lambda r: r["abs_mpjpe_cm"][k]), 4) for k in ("hip_bf16", "oracle_fp32", "delta")},
            "mean_peak_score": {k: round(wm(lambda r: r["mean_peak_score"][k]), 4) for k in ("hip_bf16", "oracle_fp32")},
            "worst_seed_abs_mpjpe_delta_cm": max(abs(r["mpjpe_cm"]["delta"]) for r in runs),
            "per_seed": [{k: r[k] for k in ("frames", "argmax_agreement", "argmax_within_1_voxel", "keypoint_shift_cm", "mpjpe_cm",
                                            "mean_peak_score", "train_seconds")} for r in runs]})
        res.pop("loss_history", None)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
