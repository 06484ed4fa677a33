"""Golden vectors for the MPJPE evaluation, captured from the REFERENCE's own code.

Runs only in the authoring container (needs /root/reference).  det3d/datasets/cruw_pose/cruw_pose.py is loaded at file level with
the stand-ins of gen_golden_input.py; the real /root/reference/eval_util.py supplies PJPE / ABS_PJPE (cruw_pose.py:15 star-imports
them).  CRUW_POSE_Dataset.evaluation (cruw_pose.py:277-311) is called UNBOUND on a namespace object carrying the two attributes it
reads (label_file, seq_id_to_name) with a seeded synthetic label file and detections dict in the layout tools/test.py:203-214
builds.  The fixture holds those inputs and the reference's `res` dict.

    python tests/golden/gen_golden_eval.py        # rewrites tests/golden/eval_golden.json
"""
import copy
import json
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden_input as GI  # noqa: E402


def main():
    _, ds = GI.import_reference()
    ev = GI._load("ref_eval_util", "eval_util.py")
    ds.PJPE, ds.ABS_PJPE = ev.PJPE, ev.ABS_PJPE          # what `from eval_util import *` binds in the reference's run
    rng = np.random.default_rng(2024)
    seq_names = {"11": "2023_1017_1619", "4": "2023_1012_1420", "27": "2023_1101_0933"}
    gt, dets = {}, {}
    for seq, nfr in (("11", 5), ("4", 3), ("27", 1)):
        gt[seq] = {}
        for i in range(nfr):
            fr = str(int(rng.integers(0, 900)))
            pelvis = np.array([rng.uniform(1.5, 7.0), rng.uniform(-4, 4), rng.uniform(-0.5, 3.5)])
            pose = pelvis + rng.normal(0, [0.25, 0.25, 0.45], size=(15, 3))
            gt[seq][fr] = [{"pose": pose.tolist()}]
            pred = pose + rng.normal(0, 0.04, size=(15, 3)) + rng.normal(0, 0.1, size=(1, 3))
            dets["%s/%s/%s" % (seq, fr, "%06d" % int(fr))] = {"keypoints": [[j, float(pred[j, 0]), float(pred[j, 1]), float(pred[j, 2]),
                                                                             float(rng.uniform(0.2, 0.95))] for j in range(15)]}
    with tempfile.TemporaryDirectory() as tmp:
        lf = os.path.join(tmp, "labels.json")
        with open(lf, "w") as f:
            json.dump(gt, f)
        ns = types.SimpleNamespace(label_file=lf, seq_id_to_name=seq_names)
        res, _ = ds.CRUW_POSE_Dataset.evaluation(ns, copy.deepcopy(dets), output_dir=None, testset=True)
    to_f = lambda d: {k: (to_f(v) if isinstance(v, dict) else float(v)) for k, v in d.items()}  # noqa: E731
    with open(os.path.join(HERE, "eval_golden.json"), "w") as f:
        json.dump({"detections": dets, "gt": gt, "seq_id_to_name": seq_names, "reference_result": to_f(res)}, f)
    print("MPJPE %.4f mm, ABS_MPJPE %.4f mm over %d sequences" % (res["results"]["MPJPE"], res["results"]["ABS_MPJPE"], len(seq_names)))


if __name__ == "__main__":
    main()
