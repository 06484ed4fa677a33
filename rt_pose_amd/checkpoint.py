"""Reference-format checkpoints: weights trained with det3d load here and vice versa.

Format written by the reference (det3d/torchie/trainer/checkpoint.py:235-260, trainer.py:354-368):
    torch.save({"meta": {"epoch", "iter", ...}, "state_dict": {name: cpu tensor}, "optimizer": <torch Adam state_dict>,
                ["scaler": GradScaler state]}, "epoch_N.pth")   + a relative symlink latest.pth
Loading (checkpoint.py:67-137, 166-217): a leading "module." (DDP) is stripped from every key, tensors are copied by name,
unexpected / missing / shape-mismatched keys are reported and skipped unless strict.  Resume (trainer.py:494-509) restores
epoch / iter and the optimizer state.

Optimizer state: the reference wraps torch.optim.Adam in fastai's OptimWrapper whose state_dict is the inner Adam's
(det3d/solver/fastai_optim.py:121-175; two param groups from split_bn_bias :17-28 -- the model has no BatchNorm, so every
parameter sits in group 0 in module-tree order = state_dict order, group 1 is empty).  Parameters that never received a
gradient (stage-4 fuse rows under final_fuse='top') have no state entry, exactly as torch's Adam skips them.
This is host-side plumbing only: tensors move between the flat device buffers and a dict; nothing is computed.
"""
import os
from collections import OrderedDict

import torch

ADAM_GROUP_DEFAULTS = dict(lr=0.0, betas=(0.9, 0.99), eps=1e-8, weight_decay=0, amsgrad=False, maximize=False, foreach=None,
                           capturable=False, differentiable=False, fused=None)


def model_state_dict(flat):
    """FlatParams -> OrderedDict(name -> cpu fp32 tensor), reference names and [Cout,Cin,kd,kh,kw] shapes."""
    return OrderedDict((k, v.detach().cpu().clone()) for k, v in flat.values.items())


def optimizer_state_dict(trainer):
    """torch.optim.Adam-style state_dict of the fused flat optimiser (index = position in the state_dict order)."""
    flat, opt = trainer.flat, trainer.opt
    live = trainer.engine.live_params
    state = {}
    for i, k in enumerate(flat.shapes):
        if k not in live or opt.t == 0:
            continue
        state[i] = {"step": torch.tensor(float(opt.t)), "exp_avg": flat._view(flat.m, k).detach().cpu().clone(),
                    "exp_avg_sq": flat._view(flat.v, k).detach().cpu().clone()}
    g0 = dict(ADAM_GROUP_DEFAULTS, betas=(0.9, opt.beta2), eps=opt.eps, params=list(range(len(flat.shapes))))
    g1 = dict(ADAM_GROUP_DEFAULTS, betas=(0.9, opt.beta2), eps=opt.eps, params=[])
    return {"state": state, "param_groups": [g0, g1]}


def save_checkpoint(trainer, filename, meta=None, save_optimizer=True):
    """checkpoint.py:235-260.  meta defaults to the trainer's position (epoch counts from 1 like trainer.py:358)."""
    if meta is None:
        meta = {}
    elif not isinstance(meta, dict):
        raise TypeError("meta must be a dict or None, but got {}".format(type(meta)))
    meta = dict(meta)
    meta.setdefault("iter", trainer.step_idx)
    meta.setdefault("epoch", getattr(trainer, "epoch", 0) + 1)
    d = os.path.dirname(filename)
    if d:
        os.makedirs(d, exist_ok=True)
    ckpt = {"meta": meta, "state_dict": model_state_dict(trainer.flat)}
    if save_optimizer:
        ckpt["optimizer"] = optimizer_state_dict(trainer)
    torch.save(ckpt, filename)
    return ckpt


def save_epoch(trainer, out_dir, filename_tmpl="epoch_{}.pth", save_optimizer=True, meta=None):
    """Trainer.save_checkpoint (trainer.py:354-368): epoch_N.pth plus a relative symlink latest.pth."""
    epoch = getattr(trainer, "epoch", 0) + 1
    name = filename_tmpl.format(epoch)
    path = os.path.join(out_dir, name)
    save_checkpoint(trainer, path, meta=dict(meta or {}, epoch=epoch, iter=trainer.step_idx), save_optimizer=save_optimizer)
    link = os.path.join(out_dir, "latest.pth")
    if os.path.lexists(link):
        os.remove(link)
    os.symlink(name, link)
    return path


def load_state_dict(flat, state_dict, strict=False, logger=None):
    """checkpoint.py:67-137: copy by name, skip (and report) unexpected / mismatched keys; -> (unexpected, missing, mismatched)."""
    unexpected, mismatched = [], []
    for name, param in state_dict.items():
        if name not in flat.values:
            unexpected.append(name)
            continue
        if isinstance(param, torch.nn.Parameter):
            param = param.data
        own = flat.values[name]
        if tuple(param.shape) != tuple(own.shape):
            mismatched.append([name, tuple(own.shape), tuple(param.shape)])
            continue
        own.copy_(param.to(own.dtype))
    missing = [k for k in flat.values if k not in state_dict and "num_batches_tracked" not in k]
    msgs = []
    if unexpected:
        msgs.append("unexpected key in source state_dict: {}\n".format(", ".join(unexpected)))
    if missing:
        msgs.append("missing keys in source state_dict: {}\n".format(", ".join(missing)))
    if mismatched:
        msgs.append("these keys have mismatched shape:\n" + "\n".join("%s: expected %r, loaded %r" % tuple(m) for m in mismatched))
    if msgs:
        msg = "The model and loaded state dict do not match exactly\n\n" + "\n".join(msgs)
        if strict:
            raise RuntimeError(msg)
        (logger.warning if logger is not None else print)(msg)
    return unexpected, missing, mismatched


def load_checkpoint(target, filename, map_location="cpu", strict=False, logger=None):
    """checkpoint.py:166-217 for local files.  target: a trainer or a FlatParams.  Returns the loaded checkpoint dict."""
    if not os.path.isfile(filename):
        raise IOError("{} is not a checkpoint file".format(filename))
    ckpt = torch.load(filename, map_location=map_location, weights_only=False)
    if isinstance(ckpt, OrderedDict):
        sd = ckpt
    elif isinstance(ckpt, dict) and "state_dict" in ckpt:
        sd = ckpt["state_dict"]
    else:
        raise RuntimeError("No state_dict found in checkpoint file {}".format(filename))
    if list(sd.keys())[0].startswith("module."):
        sd = {k[7:]: v for k, v in sd.items()}
    flat = getattr(target, "flat", target)
    # pre-version-2 DeformConvPack keys (deform_conv.py:298-321): '<name>_offset.*' -- not the current '<name>.conv_offset.*'
    if any(k.endswith(("_offset.weight", "_offset.bias")) and not k.endswith((".conv_offset.weight", ".conv_offset.bias"))
           for k in sd):
        from .dcn import migrate_pre_v2_keys
        sd = migrate_pre_v2_keys(dict(sd), target_keys=set(flat.values))
    load_state_dict(flat, sd, strict, logger)
    return ckpt


def resume(trainer, filename, resume_optimizer=True, map_location="cpu"):
    """Trainer.resume (trainer.py:494-509): weights, epoch / iter, and the Adam moments."""
    ckpt = load_checkpoint(trainer, filename, map_location=map_location)
    trainer.epoch = ckpt["meta"]["epoch"]
    trainer.step_idx = ckpt["meta"]["iter"]
    if "optimizer" in ckpt and resume_optimizer:
        flat, opt = trainer.flat, trainer.opt
        names = list(flat.shapes)
        steps = set()
        flat.m.zero_()
        flat.v.zero_()
        for i, st in ckpt["optimizer"]["state"].items():
            k = names[int(i)]
            flat._view(flat.m, k).copy_(st["exp_avg"].to(flat.m.dtype))
            flat._view(flat.v, k).copy_(st["exp_avg_sq"].to(flat.v.dtype))
            steps.add(int(st["step"]))
        # the fused optimiser keeps one step count (bias correction): torch's Adam has one per parameter, all equal for
        # parameters that received a gradient at every step
        opt.t = max(steps) if steps else 0
    return ckpt
