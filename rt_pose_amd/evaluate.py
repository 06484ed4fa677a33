"""MPJPE evaluation with the reference's aggregation:
   per-joint error   eval_util.py:5-10  (PJPE is root-relative: both poses are shifted by their joint 0)
   aggregation       det3d/datasets/cruw_pose/cruw_pose.py:277-311 -- per sequence: mean over frames, x1000 (mm), mean over
                     joints; then the mean over sequences.
Host-side numpy on a few dozen floats per frame: not a kernel."""
from collections import defaultdict

import numpy as np


def abs_pjpe(pred, gt):
    return np.linalg.norm(np.asarray(pred, np.float64) - np.asarray(gt, np.float64), axis=-1)


def pjpe(pred, gt):
    pred, gt = np.asarray(pred, np.float64), np.asarray(gt, np.float64)
    return abs_pjpe(pred - pred[:1], gt - gt[:1])


def evaluate(detections, gt, seq_names=None):
    """detections: {'seq/frame/rdr_frame': {'keypoints': [(id,x,y,z,score), ...]}};  gt: {seq: {frame: [{'pose': [[x,y,z]*15]}]}}.
    Returns the reference's `res` dict ({'results': totals, 'seq_results': per sequence + 'ALL'})."""
    rel, ab = defaultdict(list), defaultdict(list)
    for key, val in detections.items():
        seq, frame, _ = key.split("/")
        g = np.array(gt[seq][frame][0]["pose"])
        k = np.array([p[1:4] for p in val["keypoints"]])
        rel[seq].append(pjpe(k, g))
        ab[seq].append(abs_pjpe(k, g))
    seq_res = {}
    for seq in rel:
        name = seq_names[seq] if seq_names else seq
        r, a = np.mean(np.array(rel[seq]), axis=0) * 1000, np.mean(np.array(ab[seq]), axis=0) * 1000
        d = {"MPJPE": float(np.mean(r)), "ABS_MPJPE": float(np.mean(a))}
        for j in range(r.shape[0]):
            d["PJPE_%d" % j], d["ABS_PJPE_%d" % j] = float(r[j]), float(a[j])
        seq_res[name] = d
    total = {k: float(np.mean([v[k] for v in seq_res.values()])) for k in next(iter(seq_res.values()))} if seq_res else {}
    out = dict(seq_res)
    out["ALL"] = total
    return {"results": total, "seq_results": out}
