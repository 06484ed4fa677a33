"""tests/module_cases.py on the emulated kernels: checks the harness (tests/subgraph.py) and the plan logic of each sub-module on CPU;
the HIP run of the same cases is tests/test_gpu_modules_vs_oracle.py."""
import pytest
import torch

from tests import module_cases as MC
from tests.emu_backend import EmuBackend


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("name", ["hr_module_stage3", "final_concat_conv", "layer1_stem", "layer1_doppler"])
def test_module_against_the_oracle_exact_plan(name):
    """fp32 storage (EmuBackend(exact=True)): the sub-module's plan -- folds, un-folds, fan-ins, adjoints -- is the oracle's module
    up to fp32 round-off."""
    pairs = MC.run_case(EmuBackend(exact=True), name)
    for k, (got, want) in pairs.items():
        assert tuple(got.shape) == tuple(want.shape), k
        assert rel(got, want) < 2e-4, (k, rel(got, want))


def test_module_against_the_oracle_bf16_plan():
    pairs = MC.run_case(EmuBackend(), "hr_module_stage2")
    assert not MC.check_bf16(pairs), MC.check_bf16(pairs)
