"""Shared helpers for the parity tests."""
import numpy as np
import torch


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
