"""Golden vectors for the MPJPE evaluation, captured from the REFERENCE's own code.

Runs only in the authoring container (needs /root/reference).  det3d/datasets/cruw_pose/cruw_pose.py is loaded at file level with
the stand-ins of gen_golden_input.py; the real /root/reference/eval_util.py supplies PJPE / ABS_PJPE (cruw_pose.py:15 star-imports
them).  CRUW_POSE_Dataset.evaluation (cruw_pose.py:277-311) is called UNBOUND on a namespace object carrying the two attributes it
reads (label_file, seq_id_to_name) with a seeded synthetic label file and detections dict in the layout tools/test.py:203-214
builds.  The fixture holds those inputs and the reference's `res` dict.

The prediction file: tools/test.py is loaded at file level (its det3d / apex imports are empty stand-in modules -- none of them is
touched by the function) and its own save_pred (:41-63) writes the same detections; `open` is redirected for the one hard-coded path
it reads (/mnt/ssd3/cruw_pose_label/file_meta_merge.txt -> a temporary "id,name" table).  The fixture holds the file's text.

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


def reference_save_pred(dets, seq_names):
    """tools/test.py:41-63 executed as the reference wrote it; -> the text of <root>/<checkpoint>/<split>_prediction.json."""
    import builtins
    import importlib.util
    for name in ("apex", "yaml", "det3d.torchie.apis", "det3d.torchie.trainer", "det3d.torchie.trainer.utils", "det3d.torchie.utils",
                 "det3d.models", "det3d.datasets"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            sys.modules[name] = m
    for mod, names in (("det3d.datasets", ("build_dataloader", "build_dataset")), ("det3d.models", ("build_detector",)),
                       ("det3d.torchie", ("Config",)), ("det3d.torchie.trainer", ("get_dist_info", "load_checkpoint")),
                       ("det3d.torchie.apis", ("batch_processor", "build_optimizer", "get_root_logger", "init_dist", "set_random_seed", "train_detector")),
                       ("det3d.torchie.trainer.utils", ("all_gather", "synchronize")), ("det3d.torchie.utils", ("count_parameters",))):
        for n in names:
            if not hasattr(sys.modules[mod], n):
                setattr(sys.modules[mod], n, None)
    sys.modules["det3d"].torchie = sys.modules["det3d.torchie"]
    spec = importlib.util.spec_from_file_location("ref_tools_test", os.path.join(GI.REF, "tools", "test.py"))
    rt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rt)
    with tempfile.TemporaryDirectory() as tmp:
        meta = os.path.join(tmp, "file_meta_merge.txt")
        with open(meta, "w") as f:
            f.write("".join("%s,%s\n" % kv for kv in seq_names.items()))
        real_open = builtins.open

        def redirected(path, *a, **k):
            return real_open(meta if str(path) == "/mnt/ssd3/cruw_pose_label/file_meta_merge.txt" else path, *a, **k)
        builtins.open = redirected
        try:
            rt.save_pred(copy.deepcopy(dets), tmp, "epoch_5", "test")
        finally:
            builtins.open = real_open
        with open(os.path.join(tmp, "epoch_5", "test_prediction.json")) as f:
            return f.read()


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
    pred_text = reference_save_pred(dets, seq_names)
    to_f = lambda d: {k: (to_f(v) if isinstance(v, dict) else float(v)) for k, v in d.items()}  # noqa: E731
    with open(os.path.join(HERE, "eval_golden.json"), "w") as f:
        json.dump({"detections": dets, "gt": gt, "seq_id_to_name": seq_names, "reference_result": to_f(res),
                   "reference_prediction_file": {"checkpoint_name": "epoch_5", "dataset_split": "test", "relative_path": "epoch_5/test_prediction.json",
                                                 "text": pred_text}}, f)
    print("MPJPE %.4f mm, ABS_MPJPE %.4f mm over %d sequences" % (res["results"]["MPJPE"], res["results"]["ABS_MPJPE"], len(seq_names)))


if __name__ == "__main__":
    main()
