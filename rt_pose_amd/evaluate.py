"""MPJPE evaluation with the reference's aggregation:
   per-joint error   eval_util.py:5-10  (PJPE is root-relative: both poses are shifted by their joint 0)
   aggregation       det3d/datasets/cruw_pose/cruw_pose.py:277-311 -- per sequence: mean over frames, x1000 (mm), mean over
                     joints; then the mean over sequences.
   prediction file   tools/test.py:41-63 (save_pred): {seq_name: {"<frame>_<rdr_frame>": {"keypoints": [...], ...}}}, sequences
                     sorted by name, frames by int(frame), written to <root>/<checkpoint_name>/<split>_prediction.json
   detections dict   tools/test.py:194-216: key "seq/frame/rdr_frame" from each output's metadata
Host-side numpy / json on a few dozen floats per frame: not a kernel."""
import json
import os
from collections import defaultdict

import numpy as np


def collect_detections(outputs, detections=None):
    """Fold one batch of CenterHead.predict-style outputs ([{'keypoints': [...], 'metadata': {'seq','frame','rdr_frame'}}])
    into the detections dict tools/test.py builds (:203-214)."""
    detections = {} if detections is None else detections
    for out in outputs:
        m = out["metadata"]
        detections["%s/%s/%s" % (m["seq"], m["frame"], m["rdr_frame"])] = {k: v for k, v in out.items() if k != "metadata"}
    return detections


def save_pred(pred, root, checkpoint_name, dataset_split, seq_id_to_name=None):
    """tools/test.py:41-63.  seq_id_to_name: the dataset's file_meta_merge table (id -> sequence name); identity if None."""
    save_dir = os.path.join(root, "%s" % checkpoint_name)
    os.makedirs(save_dir, exist_ok=True)
    result = defaultdict(dict)
    for key, val in pred.items():
        seq, frame, rdr_frame = key.split("/")
        name = seq_id_to_name[seq] if seq_id_to_name is not None else seq
        result[name]["%s_%s" % (frame, rdr_frame)] = val
    result = dict(sorted(result.items(), key=lambda x: x[0]))
    for seq, frames in result.items():
        result[seq] = dict(sorted(frames.items(), key=lambda x: int(x[0].split("_")[0])))
    path = os.path.join(save_dir, "%s_prediction.json" % dataset_split)
    with open(path, "w") as f:
        json.dump(result, f, indent=2)
    return path


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
