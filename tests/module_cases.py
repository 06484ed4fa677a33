"""Module-level cases for tests/subgraph.py: each returns (build, inputs, params, oracle) where oracle(sd, xs) -> list of outputs
computed by oracle/hrradarpose_ref.py on fp32 tensors.  Shared by the CPU (emulated kernels) and GPU (HIP) tests."""
from collections import OrderedDict

import torch
import torch.nn.functional as F

from oracle import hrradarpose_ref as O
from rt_pose_amd import net


def _rnd(shape, seed, scale=1.0, relu=False):
    g = torch.Generator().manual_seed(seed)
    t = torch.randn(*shape, generator=g) * scale
    return torch.relu(t + 0.1) if relu else t


def _gn(sd, p, c, seed):
    sd[p + ".weight"] = _rnd((c,), seed) * 0.2 + 1.0
    sd[p + ".bias"] = _rnd((c,), seed + 1) * 0.2


def _block(sd, p, c, seed):
    for k, name in enumerate(("conv2", "conv3")):
        _gn(sd, "%s.%s.groupnorm" % (p, name), c, seed + 10 * k)
        sd["%s.%s.conv.weight" % (p, name)] = _rnd((c, c, 3, 3, 3), seed + 10 * k + 2, 0.05)


def hr_module_case(stage=3, n=2, top=(8, 16, 32), ch=(32, 32, 64, 64), seed=100):
    """HighResolutionModule of `stage` branches (hr_util/hr3d.py:66-229): one ResNetBlock per branch, every fuse row -- 1x1x1
    GroupNorm convs + trilinear upsample upwards, chains of stride-2 GroupNorm convs downwards, ReLU."""
    nb, ch = stage, list(ch[:stage])
    dims = [tuple(max(1, v >> i) for v in top) for i in range(nb)]
    sd = OrderedDict()
    for i in range(nb):
        _block(sd, "m.branches.%d.0" % i, ch[i], seed + 100 * i)
    for i in range(nb):
        for j in range(nb):
            if j > i:
                p = "m.fuse_layers.%d.%d" % (i, j)
                _gn(sd, p + ".0", ch[j], seed + 7 * i + j)
                sd[p + ".1.weight"] = _rnd((ch[i], ch[j], 1, 1, 1), seed + 7 * i + j + 3, 0.2)
            elif j < i:
                for k in range(i - j):
                    p = "m.fuse_layers.%d.%d.%d" % (i, j, k)
                    co = ch[i] if k == i - j - 1 else ch[j]
                    _gn(sd, p + ".0", ch[j], seed + 13 * i + j + k)
                    sd[p + ".1.weight"] = _rnd((co, ch[j], 3, 3, 3), seed + 13 * i + j + k + 5, 0.05)
    inputs = [("x%d" % i, _rnd((n, ch[i], *dims[i]), seed + 900 + i, relu=True)) for i in range(nb)]

    def build(g, acts):
        return net.hr_module(g, "m", stage, acts[:-1], lambda: acts[-1])

    def oracle(sdr, xs):
        return O.hr_module(sdr, "m", xs)
    return build, inputs, sd, oracle


def final_concat_conv_case(n=2, top=(8, 16, 32), ch=(32, 32, 64, 64), cout=128, seed=300):
    """HRNet3D.forward, final_fuse='conat_conv' (hrnet3d.py:37-42): cat(x0, up(x1), up(x2), up(x3)) -> Conv3d(192, 128, 1) with bias."""
    dims = [tuple(max(1, v >> i) for v in top) for i in range(len(ch))]
    sd = OrderedDict([("backbone.final_conv.weight", _rnd((cout, sum(ch), 1, 1, 1), seed, 0.1)), ("backbone.final_conv.bias", _rnd((cout,), seed + 1, 0.3))])
    inputs = [("y%d" % i, _rnd((n, ch[i], *dims[i]), seed + 10 + i, relu=True)) for i in range(len(ch))]

    def build(g, acts):
        return [net.final_concat_conv(g, acts, "backbone")]

    def oracle(sdr, xs):
        ups = [F.interpolate(t, size=xs[0].shape[2:], mode="trilinear", align_corners=True) for t in xs[1:]]
        return [F.conv3d(torch.cat([xs[0]] + ups, 1), sdr["backbone.final_conv.weight"], sdr["backbone.final_conv.bias"])]
    return build, inputs, sd, oracle


def layer1_case(cin=1, n=2, dims=(8, 16, 32), seed=500):
    """layer1 = ResNetBlock(Cin -> 32) (hr_util/common.py:98-148): conv1 1x1x1 WITH bias because Cin != Cout (:111-118) -- the Cin = 1
    stem kernel for the zyx configs, a 1x1x1 conv for the Doppler ones -- then GroupNorm convs, residual = conv1's output, ReLU."""
    sd = OrderedDict([("l.conv1.weight", _rnd((32, cin, 1, 1, 1), seed, 0.5)), ("l.conv1.bias", _rnd((32,), seed + 1, 0.2))])
    _block(sd, "l", 32, seed + 10)
    inputs = [("f32:rdr", _rnd((n, cin, *dims), seed + 50, 0.5, relu=True))]

    def build(g, acts):
        x = acts[0]
        if cin == 1:
            t0 = g.stem("l1.c1", x, dims, "l.conv1.weight", "l.conv1.bias")
        else:
            t0 = g.conv("l1.c1", g.pack("l1.in", x, cin, dims), "l.conv1.weight", bname="l.conv1.bias", ks=1)
        return [net._resblock(g, "l", t0, "l1")]

    def oracle(sdr, xs):
        return [O.resnet_block(sdr, "l", xs[0])]
    return build, inputs, sd, oracle


CASES = {"hr_module_stage2": lambda: hr_module_case(2), "hr_module_stage3": lambda: hr_module_case(3),
         "hr_module_stage4": lambda: hr_module_case(4, top=(8, 16, 32)), "final_concat_conv": final_concat_conv_case,
         "layer1_stem": lambda: layer1_case(1), "layer1_doppler": lambda: layer1_case(32)}


def run_case(be, name, sync=None, seed=7):
    """-> {key: (got, want)} over outputs, parameter gradients and input gradients of the module against the oracle's autograd."""
    from tests.subgraph import run_subgraph
    build, inputs, sd, oracle = CASES[name]()
    # forward once to learn the output shapes, then seeded output gradients
    xs0 = [t.to(torch.bfloat16).float() if not nm.startswith("f32:") else t for nm, t in inputs]
    with torch.no_grad():
        outs0 = oracle(sd, xs0)
    gys = [_rnd(tuple(o.shape), seed + i) for i, o in enumerate(outs0)]
    gys = [gy.to(torch.bfloat16).float() for gy in gys]
    res = run_subgraph(be, build, inputs, sd, gys, sync=sync)
    sdr = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in sd.items())
    xr = [t.clone().requires_grad_(True) for t in res["inputs"]]
    outs = oracle(sdr, xr)
    torch.autograd.backward(outs, gys)
    pairs = OrderedDict()
    for i, (a, b) in enumerate(zip(res["outputs"], outs)):
        pairs["out%d" % i] = (a, b.detach())
    for k, v in sdr.items():
        assert v.grad is not None and k in res["param_grads"], k
        pairs["grad." + k] = (res["param_grads"][k], v.grad)
    for i, (a, x) in enumerate(zip(res["input_grads"], xr)):
        if a is not None:
            pairs["grad.input%d" % i] = (a, x.grad)
    return pairs


def check_bf16(pairs):
    """Tolerances for a bf16-storage plan of a WHOLE sub-module (4-10 stored layers deep, small test volumes) against the oracle's fp32
    (stated; the per-layer tolerances are those of tests/test_gpu_kernels_vs_oracle.py): module outputs 1 % norm-wise; conv-weight
    and input gradients 10 % with cosine >= 0.995 (measured on the emulated plan at these sizes: 4-6 %, ReLU masks of single voxels
    flip); GroupNorm affine gradients -- 32-64 numbers, each a difference of large cancelling sums of bf16-rounded products -- 20 %
    with cosine >= 0.985.  The fp32-storage run of the same plan (tests/test_modules_vs_oracle_cpu.py) agrees to 2e-4."""
    bad = []
    for k, (got, want) in pairs.items():
        a, b = got.double().reshape(-1), want.double().reshape(-1)
        r = float((a - b).norm() / (b.norm() + 1e-30))
        cos = float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))
        if k.startswith("out"):
            ok = r < 1e-2
        elif "groupnorm" in k or k.endswith((".0.weight", ".0.bias")):
            ok = r < 0.20 and cos > 0.985
        else:
            ok = r < 0.10 and cos > 0.995
        if not ok:
            bad.append((k, round(r, 4), round(cos, 5)))
    return bad
