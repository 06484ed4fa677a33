"""Host-side mirror of the reference's module interface, registered under the same registry names:

    RadarFeatureNet   det3d/models/readers/radar_encoder.py:7-17
    HRNet3D           det3d/models/backbones/hrnet3d.py:8-56           (backbone_cfg, final_conv_in/out, final_fuse, ds_factor)
    CenterHead        det3d/models/pose_heads/center_head.py:166-360   (tasks, in_channels, share_conv_channel, weight, ...)
    RadarPoseNet      det3d/models/detectors/radar_pose_net.py:9-46    (reader, backbone, neck, pose_head, ...)

Each class is an nn.Module whose parameters carry the reference's state_dict names and shapes (so reference
checkpoints load with load_state_dict), but whose arithmetic is the static HIP launch plan of rt_pose_amd.engine:
RadarPoseNet.forward(example, return_loss=True) returns the reference's loss dict with `loss` attached to autograd
(loss.backward() replays the backward launch list and fills p.grad), return_loss=False returns the reference's
key-point list.  There is no eager-PyTorch or CPU path: without the built library / a GPU the first forward raises.
"""
from collections import OrderedDict, defaultdict

import torch
import torch.nn as nn

from . import net
from .engine import FlatParams, PoseEngine
from .registry import BACKBONES, DETECTORS, HEADS, READERS, build_backbone, build_head, build_neck, build_reader


def _default_backend(device):
    from .backend import HipBackend
    return HipBackend(device)


_backend_factory = _default_backend


def set_backend_factory(fn):
    """Test hook: CPU tests inject the emulated kernels (tests/emu_backend.py) to check the module plumbing without a
    GPU.  The product never calls this; the default factory is HipBackend, which raises without a GPU."""
    global _backend_factory
    _backend_factory = fn if fn is not None else _default_backend


class ParamTree(nn.Module):
    """Parameters addressed by dotted reference names ('stage2.0.branches.1.0.conv2.conv.weight')."""

    def add(self, dotted, tensor):
        parts = dotted.split(".")
        mod = self
        for p in parts[:-1]:
            if p not in mod._modules:
                mod.add_module(p, ParamTree())
            mod = mod._modules[p]
        mod.register_parameter(parts[-1], nn.Parameter(tensor))


def _build_params(shapes, seed=0):
    from .trainer import init_state_dict
    return init_state_dict(shapes, seed)


def _backbone_shapes(arch, final_in, final_out):
    from . import configs
    return configs.backbone_shapes(arch, final_in, final_out)


@READERS.register_module
class RadarFeatureNet(nn.Module):
    def __init__(self, name="RadarFeatureNet"):
        super().__init__()
        self.name = name

    def forward(self, rdr_cube):
        return rdr_cube


from .lidar import DynamicVoxelEncoder  # noqa: E402  (det3d/models/readers/dynamic_voxel_encoder.py:69-101)
READERS.register_module(DynamicVoxelEncoder)


@BACKBONES.register_module
class HRNet3D(nn.Module):
    def __init__(self, backbone_cfg="hr_tiny_feat32_zyx_l4", feat_transform=None, **kwargs):
        super().__init__()
        if feat_transform is not None:
            raise NotImplementedError("feat_transform is unreachable from the shipped configs (polar_to_cart.py is unregistered)")
        if backbone_cfg not in net.ARCH_TABLES:
            raise KeyError("backbone_cfg %r: only the tables used by configs/cruw_pose/*.py are built (%s)"
                           % (backbone_cfg, ", ".join(net.ARCH_TABLES)))
        self.backbone_cfg = backbone_cfg
        self.final_fuse = kwargs["final_fuse"]
        self.final_conv_in, self.final_conv_out = kwargs["final_conv_in"], kwargs["final_conv_out"]
        shapes = _backbone_shapes(backbone_cfg, self.final_conv_in, self.final_conv_out)
        init = _build_params(shapes)
        self.backbone = ParamTree()
        self.final_conv = ParamTree() if self.final_conv_in != self.final_conv_out else nn.Identity()
        for k, t in init.items():
            if k.startswith("backbone.backbone."):
                self.backbone.add(k[len("backbone.backbone."):], t)
            else:
                self.final_conv.add(k[len("backbone.final_conv."):], t)
        self._engines = {}

    def forward(self, x_):
        """Inference-only when used stand-alone (training goes through RadarPoseNet's fused plan)."""
        eng = _standalone_engine(self, x_, "backbone")
        eng.load_input(x_.float())
        eng.run_forward()
        return eng.features()


@HEADS.register_module
class CenterHead(nn.Module):
    def __init__(self, in_channels=128, tasks=[], dataset="cruw_pose", common_heads=dict(), logger=None, init_bias=-2.19,
                 share_conv_channel=64, num_hm_conv=2, weight=0.1, code_weights=[], dcn_head=False, lidar_channels=0):
        super().__init__()
        # lidar_channels > 0: two-stream fusion (BASELINE config 5; no reference counterpart, configs.LIDAR_VARIANTS): the towers'
        # first conv reads the radar feature concatenated with the dense LiDAR voxel grid (example["rdr"]["lidar_grid"])
        self.lidar_channels = int(lidar_channels)
        # dcn_head=True: the reference's DCNSepHead (center_head.py:111-163) is 2-D and cannot run on the 5-D feature (its
        # constructor also raises, :152); here the two FeatureAdaption modules run per (frame, z) slice in front of the
        # SepHead towers (parity unpinned by construction)
        self.dcn_head = bool(dcn_head)
        if len(tasks) != 1 or num_hm_conv != 2:
            raise NotImplementedError("one task with two convs per head, as in every shipped config")
        self.class_names = [t["class_names"] for t in tasks]
        self.num_classes = [len(t["class_names"]) for t in tasks]
        self.weight, self.code_weights, self.dataset = weight, list(code_weights), dataset
        self.in_channels = in_channels
        self.heads = OrderedDict((k, v[0]) for k, v in dict(common_heads).items())
        self.heads["hm"] = self.num_classes[0]
        shapes = OrderedDict()
        # shared_conv (center_head.py:203-211): GroupNorm(8, in) -> Conv3d(in, share, 3x3x3, no bias) -> ReLU when the channel counts
        # differ, Identity otherwise (every shipped config)
        self.has_shared_conv = in_channels != share_conv_channel
        if self.has_shared_conv:
            if self.dcn_head or self.lidar_channels:
                raise NotImplementedError("shared_conv together with dcn_head / lidar_channels")
            if in_channels % 8:
                raise ValueError("GroupNorm(8, %d): in_channels must be a multiple of 8 (center_head.py:205)" % in_channels)
            shapes["pose_head.shared_conv.0.weight"] = (in_channels,)
            shapes["pose_head.shared_conv.0.bias"] = (in_channels,)
            shapes["pose_head.shared_conv.1.weight"] = (share_conv_channel, in_channels, 3, 3, 3)
        if self.dcn_head:
            for which in ("cls", "reg"):
                p = "pose_head.tasks.0.feature_adapt_%s" % which
                shapes[p + ".conv_offset.weight"] = (72, share_conv_channel, 1, 1)
                shapes[p + ".conv_offset.bias"] = (72,)
                shapes[p + ".conv_adaption.weight"] = (share_conv_channel, share_conv_channel, 3, 3)
        for hname, ncls in self.heads.items():
            p = "pose_head.tasks.0.%s" % hname
            shapes[p + ".0.weight"] = (32, share_conv_channel + self.lidar_channels, 3, 3, 3)
            shapes[p + ".0.bias"] = (32,)
            shapes[p + ".2.weight"] = (ncls, 32, 3, 3, 3)
            shapes[p + ".2.bias"] = (ncls,)
        init = _build_params(shapes)
        if init_bias != -2.19:
            init["pose_head.tasks.0.hm.2.bias"].fill_(init_bias)
        self.shared_conv = ParamTree() if self.has_shared_conv else nn.Identity()
        self.tasks = ParamTree()
        for k, t in init.items():
            if k.startswith("pose_head.shared_conv."):
                self.shared_conv.add(k[len("pose_head.shared_conv."):], t)
            else:
                self.tasks.add(k[len("pose_head.tasks."):], t)
        self._engines = {}
        self._last = None
        self._last_x = None

    def forward(self, x, *kwargs):
        """-> ([{head name: [B, classes, Z, Y, X]}], x) like center_head.py:232-238 (x: the feature the towers read, i.e. behind
        shared_conv when there is one).  In training mode the plan keeps what CenterHead.loss needs for the backward sweep."""
        eng = _standalone_engine(self, x, "head_train" if (self.training and torch.is_grad_enabled()) else "head")
        eng.load_features(x.detach())
        eng.run_forward()
        self._last, self._last_x = eng, x
        return [OrderedDict((k, eng.output(k)) for k in self.heads)], (eng.tower_input() if self.has_shared_conv else x)

    def loss(self, example, preds_dicts, test_cfg, **kwargs):
        """center_head.py:244-270 on the stand-alone head: the loss kernels run on the logits of the LAST forward() (preds_dicts must be
        what that call returned: the plan owns those buffers), and `loss` is attached to autograd -- backward() replays the head's
        backward launch list, fills the head parameters' .grad and hands the feature's gradient to whatever produced the feature."""
        eng = self._last
        if eng is None or not eng.train:
            raise RuntimeError("CenterHead.loss: call forward() in training mode (head.train(), gradients enabled) first")
        got = preds_dicts[0]
        if any(got[k].data_ptr() != eng.output(k).data_ptr() for k in self.heads):
            raise ValueError("CenterHead.loss: preds_dicts must be the output of the last forward() call of this head")
        eng.load_targets(example)
        eng.run_losses_only()
        named = list(self.named_parameters())
        loss = _HeadLoss.apply(self, eng, self._last_x, *[p for _, p in named])
        return _loss_dict(eng, loss)

    @torch.no_grad()
    def predict(self, example, preds_dicts, test_cfg, **kwargs):
        eng = self._last
        if eng is None:
            raise RuntimeError("CenterHead.predict: call forward() first")
        eng.set_test_cfg(_plain(test_cfg))
        eng.run_decode()
        return eng.keypoints(example.get("meta") if isinstance(example, dict) else None)


def _plain(cfg):
    return {k: cfg[k] for k in ("out_size_factor", "voxel_size", "pc_range", "score_threshold") if k in cfg}


def _loss_dict(eng, loss_tensor=None):
    l = eng.losses()
    rets = defaultdict(list)
    rets["loss"].append(loss_tensor if loss_tensor is not None else l["loss"])
    rets["hm_loss"].append(l["hm_loss"].detach().cpu())
    rets["loc_loss"].append(l["loc_loss"])
    rets["loc_loss_elem"].append(l["loc_loss_elem"].detach().cpu())
    rets["num_positive"].append(l["num_positive"])
    return rets


def _standalone_engine(mod, x, kind):
    """Plans for a backbone (inference) or a head (inference / training) used on its own."""
    key = (tuple(x.shape), str(x.device), kind)
    eng = mod._engines.get(key)
    if eng is None:
        off = [k for k, p in mod.named_parameters() if p.device != x.device]
        if off:   # (the kernels would read host pointers: fail here, not in a GPU memory fault)
            raise RuntimeError("%s: parameters (%s, ...) are on %s but the input is on %s -- move the module first (module.to(%r))"
                               % (type(mod).__name__, off[0], next(mod.parameters()).device, x.device, str(x.device)))
        eng = _StandaloneEngine(mod, x, kind) if kind == "backbone" else _HeadEngine(mod, x, kind == "head_train")
        mod._engines[key] = eng
    return eng


class _StandaloneEngine:
    """Inference plan of a backbone used on its own."""

    def __init__(self, mod, x, kind):
        from .graph import Graph
        be = _backend_factory(x.device)
        self.be, self.kind, self.train = be, kind, False
        b, c, d, h, w = x.shape
        self.n, self.dims = b, (d, h, w)
        params = OrderedDict(("backbone." + k, p.data) for k, p in mod.named_parameters())
        g = self.graph = Graph(be, b, params, train=False)
        self.x_in = g.input_f32("rdr", c, self.dims)
        self.feats = net.build_hrnet3d(g, self.x_in, mod.backbone_cfg, self.dims, mod.final_fuse)
        self.fwd = list(g.forward_list())

    def load_input(self, x):
        self.x_in.copy_(x.reshape(self.x_in.shape))

    def run_forward(self):
        s = self.be.stream()
        for f in self.fwd:
            f(s)

    def features(self):
        a = self.feats
        return a.buf[..., :a.c_real].permute(0, 4, 1, 2, 3)


class _HeadEngine(PoseEngine):
    """CenterHead on its own (engine.PoseEngine's head-only mode): inference, or training with the losses and the backward list."""

    def __init__(self, mod, x, train):
        be = _backend_factory(x.device)
        b, c, d, h, w = x.shape
        params = OrderedDict(("pose_head." + k, p.data) for k, p in mod.named_parameters())
        self.param_grads = OrderedDict((k, be.alloc(tuple(p.shape), "f32")) for k, p in params.items()) if train else {}
        cw = list(mod.code_weights) if len(mod.code_weights) == mod.heads["reg"] else [1.0] * mod.heads["reg"]
        super().__init__(be, params, None, None, mod.heads, mod.weight, cw, b, (d, h, w), train=train,
                         pgrads=self.param_grads if train else None, feature_channels=c)

    def tower_input(self):
        """The feature behind shared_conv (what center_head.py:234-238 returns beside the predictions)."""
        a = next(op.y for op in self.graph.ops if getattr(op.y, "name", "") == "shared")
        return a.buf[..., :a.c_real].permute(0, 4, 1, 2, 3)


class _PlanLoss(torch.autograd.Function):
    """Attaches the plan's scalar loss to autograd: backward replays the backward launch list, which writes the
    parameter gradients straight into the flat gradient buffer the parameters' .grad tensors view."""

    @staticmethod
    def forward(ctx, holder, *params):
        ctx.holder = holder
        return holder.engine.losses()["loss"].detach().clone()

    @staticmethod
    def backward(ctx, go):
        h = ctx.holder
        h.engine.run_backward_only()
        grads = []
        unit = float(go) == 1.0
        for name, p in h.named:
            if name in h.engine.live_params:
                g = h.flat.grads[name]
                grads.append(g.clone() if unit else g * go)
            else:
                grads.append(None)
        return (None, *grads)


class _HeadLoss(torch.autograd.Function):
    """The stand-alone head's scalar loss in autograd: backward replays the head plan's backward list (parameter gradients land in the
    plan's gradient buffers, the feature's gradient in eng.feat_grad)."""

    @staticmethod
    def forward(ctx, head, eng, x, *params):
        ctx.head, ctx.eng = head, eng
        ctx.x_needs = bool(torch.is_tensor(x) and x.requires_grad)
        return eng.losses()["loss"].detach().clone()

    @staticmethod
    def backward(ctx, go):
        eng = ctx.eng
        eng.run_backward_only()
        gx = (eng.feat_grad * go).to(ctx.head._last_x.dtype) if (ctx.x_needs and eng.feat_grad is not None) else None
        grads = []
        for name, _ in ctx.head.named_parameters():
            g = eng.param_grads.get("pose_head." + name)
            grads.append(None if (g is None or ("pose_head." + name) not in eng.live_params) else g * go)
        return (None, None, gx, *grads)


@DETECTORS.register_module
class RadarPoseNet(nn.Module):
    def __init__(self, reader, backbone, neck, pose_head, sensor_type="rdr", train_cfg=None, test_cfg=None, pretrained=None):
        super().__init__()
        self.reader = build_reader(reader)
        self.backbone = build_backbone(backbone)
        if neck is not None:
            raise NotImplementedError("every shipped config has neck=None (configs/cruw_pose/hr3d.py:81)")
        self.pose_head = build_head(pose_head)
        self.train_cfg, self.test_cfg, self.sensor_type = train_cfg, test_cfg, sensor_type
        self._plans = {}
        self.flat = None
        if pretrained is not None:
            self.load_state_dict(torch.load(pretrained, map_location="cpu").get("state_dict", {}), strict=False)

    @property
    def with_neck(self):
        return False

    # ------------------------------------------------------------------ plan management
    def _flatten(self, device, be):
        """Re-point every parameter at a view of one flat fp32 buffer (values preserved) so the optimiser step and the
        gradient all-reduce are single launches; p.grad views the flat gradient buffer."""
        named = list(self.named_parameters())
        shapes = OrderedDict((n, tuple(p.shape)) for n, p in named)
        flat = FlatParams(shapes, be.alloc)
        for n, p in named:
            flat.values[n].copy_(p.data)
            p.data = flat.values[n]
        self.flat, self._named = flat, named

    def _plan(self, x, train):
        key = (tuple(x.shape), str(x.device), bool(train))
        plan = self._plans.get(key)
        if plan is None:
            be = _backend_factory(x.device)
            if self.flat is None or self.flat.p.device != x.device:
                self._flatten(x.device, be)
            bb, hd = self.backbone, self.pose_head
            eng = PoseEngine(be, self.flat.values, bb.backbone_cfg, bb.final_fuse, hd.heads, hd.weight, hd.code_weights,
                             x.shape[0], tuple(x.shape[2:]), train=train, pgrads=self.flat.grads if train else None,
                             test_cfg=_plain(self.test_cfg) if self.test_cfg else None,
                             lidar_channels=getattr(hd, "lidar_channels", 0))
            plan = type("Plan", (), {})()
            plan.engine, plan.flat, plan.named = eng, self.flat, self._named
            self._plans[key] = plan
        return plan

    # ------------------------------------------------------------------ reference call convention
    def extract_feat(self, data):
        plan = self._plan(data["rdr_tensor"], False)
        plan.engine.load_input(data["rdr_tensor"].float())
        plan.engine.run_forward()
        return plan.engine.features()

    def forward(self, example, return_loss=True, **kwargs):
        ex = dict(example[self.sensor_type])
        ex["meta"] = example.get("meta")
        x = ex["rdr_tensor"]
        plan = self._plan(x, bool(return_loss))
        eng = plan.engine
        eng.load_input(x.float())
        if eng.lidar_in is not None:
            eng.load_lidar(ex["lidar_grid"].float())
        if return_loss:
            eng.load_targets(ex)
            eng.run_forward()
            eng.run_losses_only()
            params = [p for _, p in plan.named]
            loss = _PlanLoss.apply(plan, *params)
            return _loss_dict(eng, loss)
        eng.run_forward()
        eng.run_decode()
        return eng.keypoints(ex["meta"])
