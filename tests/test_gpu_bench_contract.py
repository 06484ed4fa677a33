"""The bench line's contract at N = 1 (the driver parses it): one JSON line on stdout with the metric fields, `config.workload`,
and the `roofline` object -- bound / achieved / peak / unit / frac / traffic measured live with HIP events around every launch of
the dominant kernel family.  The child is a fresh process; the informational legs (CPU baseline: 60-90 s, PyTorch-on-GPU child,
other configs, DCN, LiDAR) are switched off here and checked for ABSENCE; the default run carries them (profiles/r06_bench_default.json)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_single_rank_bench_line(tmp_path):
    out, err = open(tmp_path / "out", "w+"), open(tmp_path / "err", "w+")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-torch-gpu",
                        "--no-other-models", "--no-dcn", "--no-lidar"], stdout=out, stderr=err, stdin=subprocess.DEVNULL, cwd=ROOT, timeout=500)
    out.seek(0), err.seek(0)
    assert p.returncode == 0, err.read()[-3000:]
    lines = [ln for ln in out.read().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["metric"].startswith("radar frames/sec (train)") and d["unit"] == "frames/s" and d["higher_is_better"] is True
    assert (d["n_gpus"], d["steps"], d["warmup"], d["scaling"], d["vs_baseline"], d["dtype"], d["data"]) == (1, 5, 2, "weak", None, "bf16", "synthetic")
    assert abs(d["value"] - 8 * 1e3 / d["ms_per_step"]) < 1e-2 * d["value"] and 500 < d["value"] < 5000
    assert "hr3d train step" in d["config"]["workload"] and "[1,16,64,160]" in d["config"]["workload"] and d["config"]["global_batch"] == 8
    assert len(d["segments_ms_per_step"]) == 3 and d["collective"] is None and d["allreduce_ms"] is None
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.2 < r["frac"] < 0.6
    assert r["kernel"].startswith("conv_tiled_kernel") and r["traffic"] and 2.4e8 < r["traffic"] < 3.2e8   # bytes per launch (PMC artefact)
    fam = r["families"][r["kernel"]]
    assert fam["launches_per_step"] == 20 and abs(fam["tflops"] - r["achieved"]) < 1e-6
    fw = r["full_width"]
    assert "error" not in fw and 0.25 < fw["frac"] < 0.6 and fw["avg_us_per_launch"] > 40
    assert d["forward_only"]["value"] > d["value"]
    for leg in ("cpu_baseline", "torch_gpu_baseline", "other_models", "dcn_op", "lidar_stream"):
        assert leg not in d, leg
