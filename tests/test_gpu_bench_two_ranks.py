"""bench.py's N > 1 control flow on ONE GPU (reference launch: tools/train.py:93-126, one process per GPU).

The driver launches `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`; on a 1-GPU box RCCL cannot run, so
the two ranks are started here as two fresh child processes (subprocess from a fresh interpreter -- never exec'd from a process that
touched the GPU) with RTP_BENCH_ONE_DEVICE=1: both on cuda:0, gloo as the process group.  What is checked is the file's own
multi-rank logic, the part no other test reaches: rendezvous from RANK / WORLD_SIZE / MASTER_*, barrier-bracketed segments, the
all_gather of every rank's clock and the MAX over ranks, and rank 0 -- only rank 0 -- printing ONE JSON line with the collective
object filled in.  The numbers themselves mean nothing (two processes share one GPU).

Hang-proofing: the children write to FILES (two PIPEs drained one after the other deadlock as soon as the
second child is chatty), both are polled against ONE deadline, on expiry BOTH are killed and the failure carries both ranks'
tails; bench.py itself arms faulthandler.dump_traceback_later when WORLD_SIZE > 1, so a stuck rank names its own line first.
tests/conftest.py runs every multi-process test after the parity tests."""
import json
import os
import sys

import pytest

from tests.util import run_ranks

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
@pytest.mark.parametrize("dims,batch", [("8,16,32", 2), pytest.param("16,64,160", 8, marks=pytest.mark.slow)])
def test_bench_two_ranks_on_one_gpu_prints_one_line_with_the_collective(tmp_path, dims, batch):
    """Default: a small radar tensor, 2 frames per rank (what is checked is control flow; two sets of chip-filling persistent kernels
    from two processes on one GPU are the rig's problem, not the product's).  RTP_SLOW=1 adds the native shape at 8 frames per rank."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-other-models",
           "--no-dcn", "--no-lidar", "--no-torch-gpu", "--no-cpu-baseline", "--dims", dims, "--batch", str(batch)]
    _, outs, _ = run_ranks(cmd, 2, str(tmp_path), {"RTP_BENCH_ONE_DEVICE": "1"})
    lines = [[ln for ln in o.splitlines() if ln.startswith("{")] for o in outs]
    assert len(lines[0]) == 1 and len(lines[1]) == 0, "exactly one JSON line, from rank 0"
    d = json.loads(lines[0][0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1
    assert d["config"]["global_batch"] == 2 * batch and d["config"]["parallelism"] == "dp2"
    assert d["scaling"] == "weak" and d["higher_is_better"] is True and d["vs_baseline"] is None
    c = d["collective"]
    assert c["ranks"] == 2 and c["backend"] == "gloo"
    assert len(c["ms_per_step_by_rank"]) == 2 and all(v > 0 for v in c["ms_per_step_by_rank"])
    assert c["rank_spread_ms"] >= 0 and abs(c["rank_spread_ms"] - (max(c["ms_per_step_by_rank"]) - min(c["ms_per_step_by_rank"]))) < 2e-3
    assert d["allreduce_ms"] is not None and d["allreduce_ms"] > 0
    assert len(d["segments_ms_per_step"]) == 3
    # value = the frames ALL ranks processed / the slowest rank's time of the median segment
    assert abs(d["value"] - 2 * batch * 1e3 / d["ms_per_step"]) < 1e-2 * d["value"]
    assert abs(d["ms_per_step"] - max(c["ms_per_step_by_rank"])) < 2e-3
    assert "roofline" not in d and "cpu_baseline" not in d, "single-rank legs stay out of a multi-rank line"
    assert d["forward_only"]["value"] > 0
