"""A sub-module of HRRadarPose as a plan of its own, for module-level parity against the ORACLE (the per-kernel
suite mostly compares with the builder's emulation).  The real graph constructors (rt_pose_amd.net / graph) build the sub-module on
graph-input activations; forward and backward launch lists run on one stream; outputs, parameter gradients and input gradients
come back as fp32 NCDHW CPU tensors -- what O.<module> and torch.autograd give on the same (bf16-representable) inputs.

    res = run_subgraph(be, build, inputs, params, out_grads)
      be        HipBackend (GPU tests) or EmuBackend (CPU tests: the harness itself)
      build     fn(g, [Act, ...]) -> [Act, ...]   (outputs)
      inputs    [(name, fp32 NCDHW tensor), ...]   rounded to bf16 on the way in (returned, rounded, as res["inputs"]);
                a name starting with "f32:" is handed over as the fp32 NCDHW network input (Graph.input_f32), not as an activation
      params    {reference state_dict name: fp32 tensor}
      out_grads [fp32 NCDHW tensor | None per output] -> seeds of the backward sweep (None: forward only)
"""
from collections import OrderedDict

import torch

from rt_pose_amd.graph import Graph, View, pad_to


def _to_cl(t, c_pad, dtype, device):
    n, c, d, h, w = t.shape
    cl = torch.zeros(n, d, h, w, c_pad, dtype=dtype)
    cl[..., :c] = t.permute(0, 2, 3, 4, 1).to(dtype)
    return cl.to(device)


def _from_cl(buf, c):
    return buf[..., :c].float().cpu().permute(0, 4, 1, 2, 3).contiguous()


def run_subgraph(be, build, inputs, params, out_grads=None, sync=None):
    dev = getattr(be, "device", "cpu")
    n = inputs[0][1].shape[0]
    dparams = OrderedDict((k, v.detach().float().to(dev).contiguous()) for k, v in params.items())
    train = out_grads is not None
    g = Graph(be, n, dparams, train=train)
    acts, rounded = [], []
    for name, t in inputs:
        c, dims = t.shape[1], tuple(t.shape[2:])
        if name.startswith("f32:"):
            buf = g.input_f32(name[4:], c, dims)
            buf.copy_(t.reshape(buf.shape))
            acts.append(buf)
            rounded.append(t.clone())
            continue
        a = g.act(name, c, dims, c=pad_to(c, 32), needs_grad=train)
        a.buf.copy_(_to_cl(t, a.c, a.buf.dtype, a.buf.device))
        acts.append(a)
        rounded.append(_from_cl(a.buf, c))
    outs = build(g, acts)
    if train:
        for y, gy in zip(outs, out_grads):
            if gy is None:
                continue
            gc = pad_to(pad_to(y.c_real, 16), 32) if y.dtype == "f32" else y.c
            buf = _to_cl(gy, gc, torch.bfloat16 if not getattr(be, "exact", False) else torch.float32, y.buf.device)
            g.seed_grad(y, View(buf, n, y.d, y.h, y.w, gc, 0, gc))
        g.grad_leaves = [a for a in acts if hasattr(a, "contribs")]
        g.build_backward()
    s = be.stream()
    for L in g.forward_list():
        L.fn(s)
    for L in (g.bwd if train else ()):
        L.fn(s)
    if sync is not None:
        sync()
    res = {"inputs": rounded, "outputs": [_from_cl(y.buf, y.c_real) for y in outs], "graph": g}
    if train:
        res["param_grads"] = OrderedDict((k, v.detach().float().cpu()) for k, v in g.pgrad.items() if v is not None and k in g.used_params)
        res["input_grads"] = [None if getattr(a, "grad", None) is None or not hasattr(a, "contribs")
                              else _from_cl(a.grad.buf[..., a.grad.co:], a.c_real) for a in acts]
    return res
