"""Shared helpers for the parity tests."""
import os
import socket
import subprocess
import time

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEADLINE_S = float(os.environ.get("RTP_TWO_RANK_DEADLINE_S", "120"))   # (a two-rank test takes 4-8 s)
_RETRIES_LEFT = [1]   # per pytest session: the driver's window for the whole -m gpu run is finite


def sample_index(numel, nsample=8192):
    # mirrors tests/golden/gen_golden.py:sample_index
    return np.random.RandomState(numel % 2**31).randint(0, numel, size=nsample)


def check_golden(golden, key, t, rtol, atol):
    """Compare tensor `t` with the golden entry `key` (full tensor, or samples+moments for big ones)."""
    a = np.asarray(t.detach().float().cpu().numpy() if torch.is_tensor(t) else t, np.float64)
    if key in golden.files:
        np.testing.assert_allclose(a, golden[key], rtol=rtol, atol=atol, err_msg=key)
        return
    ref = golden[key + "#samples"]
    idx = sample_index(a.size, ref.shape[0])
    np.testing.assert_allclose(a.reshape(-1)[idx], ref, rtol=rtol, atol=atol, err_msg=key)
    mom = golden[key + "#moments"]
    got = np.asarray([a.mean(), a.std(), np.abs(a).max()])
    np.testing.assert_allclose(got, mom, rtol=max(rtol, 1e-4) * 5, atol=atol, err_msg=key + " moments")


def rel_err(a, b):
    a = a.detach().double().cpu() if torch.is_tensor(a) else torch.as_tensor(a, dtype=torch.float64)
    b = b.detach().double().cpu() if torch.is_tensor(b) else torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).norm() / (b.norm() + 1e-30))


def check_golden_like_emulation(golden, key, got, emu, slack=1.5):
    """`got` (the HIP kernels' bf16 result) against the golden entry `key` (the reference's fp32 values), with bounds DERIVED from
    what bf16 storage costs: `emu` is the emulated plan's result (tests/emu_backend.py: the same rounding points, fp32 CPU
    arithmetic in between), so emu - golden is the error distribution the precision choice itself produces, and the kernels may
    differ from it by summation order only.  Norm-wise: rms(got - golden) <= slack * rms(emu - golden).  Per element: no outlier
    beyond the emulated plan's own worst case (slack * max|emu - golden| + 4 rms), and |got - emu| <= 8 rms(emu - golden)."""
    a = np.asarray(got.detach().float().cpu().numpy() if torch.is_tensor(got) else got, np.float64).reshape(-1)
    e = np.asarray(emu.detach().float().cpu().numpy() if torch.is_tensor(emu) else emu, np.float64).reshape(-1)
    if key in golden.files:
        g = np.asarray(golden[key], np.float64).reshape(-1)
    else:
        g = np.asarray(golden[key + "#samples"], np.float64)
        idx = sample_index(a.size, g.shape[0])
        a, e = a[idx], e[idx]
    d_e, d_h = e - g, a - g
    rms_e = float(np.sqrt(np.mean(d_e ** 2))) + 1e-7
    rms_h = float(np.sqrt(np.mean(d_h ** 2)))
    assert rms_h <= slack * rms_e, (key, "norm-wise", rms_h, rms_e)
    assert float(np.abs(d_h).max()) <= slack * float(np.abs(d_e).max()) + 4 * rms_e, (key, "per element", float(np.abs(d_h).max()), float(np.abs(d_e).max()), rms_e)
    assert float(np.abs(a - e).max()) <= 8 * rms_e, (key, "kernels vs emulated plan", float(np.abs(a - e).max()), rms_e)
    return rms_h, rms_e


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_ranks(cmd, world, tmp_path, extra_env=None, deadline_s=DEADLINE_S, attempts=2):
    """Start `world` fresh children of `cmd` (RANK / WORLD_SIZE / MASTER_* in the env, output to files), wait for all of them against
    one deadline; returns (return codes, stdout texts, stderr texts).  On expiry every child still alive is killed.  A rank that
    EXITS non-zero fails the test at once; a deadline expiry (two processes sharing one GPU is a test rig, not the product's
    one-process-per-GPU layout) is retried on a fresh port -- once per pytest session -- with a warning carrying the first attempt's
    tails."""
    import warnings
    for attempt in range(attempts):
        codes, outs, errs, timed_out = _run_ranks_once(cmd, world, os.path.join(tmp_path, "attempt%d" % attempt), extra_env, deadline_s)
        if not timed_out and all(c == 0 for c in codes):
            return codes, outs, errs
        tails = "\n".join("---- rank %d: rc %s%s\n[stdout]\n%s\n[stderr]\n%s" % (r, codes[r], " (killed at the deadline)" if r in timed_out else "",
                                                                                   outs[r][-1500:], errs[r][-4000:]) for r in range(world))
        msg = "attempt %d: ranks %s still running after %.0f s / exit codes %s\n%s" % (attempt, timed_out, deadline_s, codes, tails)
        crashed = any(c != 0 for r, c in enumerate(codes) if r not in timed_out)
        if crashed or attempt == attempts - 1 or _RETRIES_LEFT[0] <= 0:
            pytest.fail(msg, pytrace=False)
        _RETRIES_LEFT[0] -= 1
        warnings.warn("two-rank rig: " + msg)


def _run_ranks_once(cmd, world, out_dir, extra_env, deadline_s):
    os.makedirs(out_dir, exist_ok=True)
    port = _free_port()
    procs, files = [], []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", RTP_HANG_DUMP_S=str(int(deadline_s * 0.75)))
        env.update(extra_env or {})
        fo, fe = open(os.path.join(out_dir, "rank%d.out" % r), "w+"), open(os.path.join(out_dir, "rank%d.err" % r), "w+")
        files.append((fo, fe))
        procs.append(subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=fo, stderr=fe, stdin=subprocess.DEVNULL, text=True))
    end = time.monotonic() + deadline_s
    try:
        while time.monotonic() < end and any(p.poll() is None for p in procs):
            if any(p.poll() not in (None, 0) for p in procs):   # one rank died: its peer would wait for the collective's own timeout
                time.sleep(2.0)
                break
            time.sleep(0.2)
    finally:
        timed_out = [r for r, p in enumerate(procs) if p.poll() is None]
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()

    def text(f):
        f.flush()
        f.seek(0)
        t = f.read()
        f.close()
        return t
    outs, errs = [text(fo) for fo, _ in files], [text(fe) for _, fe in files]
    return [p.returncode for p in procs], outs, errs, timed_out
