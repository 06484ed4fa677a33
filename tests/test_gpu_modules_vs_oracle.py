"""Module-level parity of the HIP kernels against the ORACLE: each sub-module of the path -- built by the product's
own graph constructors on graph-input activations (tests/subgraph.py) -- forward AND backward against oracle/hrradarpose_ref.py and
torch.autograd on the same bf16-representable inputs:

    HighResolutionModule, 2 / 3 / 4 branches      O.hr_module       hr_util/hr3d.py:66-229   
    HRNet3D 'conat_conv' fuse                      cat + conv1x1     hrnet3d.py:37-42         (A6)
    layer1 = ResNetBlock(Cin -> 32), Cin = 1 / 32  O.resnet_block    hr_util/common.py:98-148 (A2; the stem kernel / the 1x1x1 conv)

The same cases run on the emulated kernels with fp32 storage in tests/test_modules_vs_oracle_cpu.py (2e-4: the plan IS the module).
Tolerances here: tests/module_cases.py::check_bf16."""
import pytest
import torch

from tests import module_cases as MC

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from rt_pose_amd.backend import HipBackend
    return HipBackend("cuda:0")


@pytest.mark.parametrize("name", list(MC.CASES))
def test_module_against_the_oracle(hip, name):
    pairs = MC.run_case(hip, name, sync=torch.cuda.synchronize)
    for k, (got, want) in pairs.items():
        assert tuple(got.shape) == tuple(want.shape), k
    bad = MC.check_bf16(pairs)
    assert not bad, bad
