"""Plain PyTorch-CPU fp32 restatement of the reference's HRRadarPose hot path.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Written functionally: every
function takes a flat ``state_dict`` whose keys/shapes are the reference module
tree's (so reference checkpoints drive it unchanged) and calls ``torch.nn.functional``
ops in the same order the reference modules do.  Each function cites the
reference file:line it restates (paths relative to /root/reference).
"""
from collections import OrderedDict, defaultdict

import numpy as np
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------
# arch tables -- det3d/models/backbones/hrnet3D_config.py:85-187 (only the
# tables the four shipped configs select; NUM_MODULES = NUM_BLOCKS = 1 everywhere)
# ----------------------------------------------------------------------------
ARCHS = {
    "hr_tiny_feat32_zyx_l4": dict(inplanes=1, channels=[32, 32, 64, 64]),
    "hr_tiny_feat32_zyx_l4_in32": dict(inplanes=32, channels=[32, 32, 64, 64]),
    "hr_tiny_feat64_zyx_l4_in64": dict(inplanes=64, channels=[64, 64, 128, 128]),
    "hr_tiny_feat16_zyx_l4": dict(inplanes=1, channels=[16, 32, 64, 64]),
}


def _gn(sd, key, x, groups=8):
    w = sd[key + ".weight"]
    if w.numel() < groups:  # hr_util/common.py:53-54
        groups = 1
    return F.group_norm(x, groups, w, sd[key + ".bias"], eps=1e-5)


def _conv(sd, key, x, stride=1, padding=0):
    return F.conv3d(x, sd[key + ".weight"], sd.get(key + ".bias"), stride=stride, padding=padding)


def single_conv(sd, p, x, relu):
    """hr_util/common.py:73-96 with order 'gcr' / 'gc' (create_conv :25-71)."""
    x = _gn(sd, p + ".groupnorm", x)
    x = _conv(sd, p + ".conv", x, 1, 1)
    return F.relu(x) if relu else x


def resnet_block(sd, p, x):
    """hr_util/common.py:98-148: conv1 (1x1x1 iff Cin!=Cout) -> gcr -> gc -> +res -> ReLU."""
    res = _conv(sd, p + ".conv1", x) if (p + ".conv1.weight") in sd else x
    out = single_conv(sd, p + ".conv2", res, relu=True)
    out = single_conv(sd, p + ".conv3", out, relu=False)
    return F.relu(out + res)


def _gn_conv_seq(sd, p, x, stride, padding, relu):
    """nn.Sequential(GroupNorm(8,C), Conv3d(bias=False)[, ReLU]) -- hr3d.py:147-155,168-197,297-305,323-324."""
    x = _gn(sd, p + ".0", x)
    x = _conv(sd, p + ".1", x, stride, padding)
    return F.relu(x) if relu else x


def hr_module(sd, p, xs, n_out=None):
    """HighResolutionModule.forward -- hr_util/hr3d.py:205-229 (fuse layers :135-200)."""
    nb = len(xs)
    xs = [resnet_block(sd, f"{p}.branches.{i}.0", xs[i]) for i in range(nb)]
    if nb == 1:
        return xs
    outs = []
    for i in range(nb if n_out is None else n_out):
        y = None
        for j in range(nb):
            if j == i:
                t = xs[j]
            elif j > i:
                t = _gn_conv_seq(sd, f"{p}.fuse_layers.{i}.{j}", xs[j], 1, 0, relu=False)
                t = F.interpolate(t, size=xs[i].shape[2:], mode="trilinear", align_corners=True)
            else:
                t = xs[j]
                for k in range(i - j):
                    t = _gn_conv_seq(sd, f"{p}.fuse_layers.{i}.{j}.{k}", t, 2, 1, relu=(k != i - j - 1))
            y = t if y is None else y + t
        outs.append(F.relu(y))
    return outs


def hr3d_backbone(sd, x, p="backbone.backbone"):
    """HighResolution3DNet.forward -- hr_util/hr3d.py:373-399; transitions :286-331."""
    x = resnet_block(sd, p + ".layer1", x)
    ys = [x]
    for stage in (2, 3, 4):
        if f"{p}.stage{stage}.0.branches.0.0.conv2.conv.weight" not in sd:
            break
        # new lowest branch from the LAST branch of the previous stage (:386,:394)
        new = _gn_conv_seq(sd, f"{p}.transition{stage - 1}.{stage - 1}.0", ys[-1], 2, 1, relu=True)
        ys = hr_module(sd, f"{p}.stage{stage}.0", ys + [new])
    return ys


def hrnet3d(sd, x, final_fuse, p="backbone"):
    """HRNet3D.forward -- det3d/models/backbones/hrnet3d.py:29-43."""
    ys = hr3d_backbone(sd, x, p + ".backbone")
    if final_fuse == "top":
        feats = ys[0]
        if (p + ".final_conv.weight") in sd:
            feats = _conv(sd, p + ".final_conv", feats)
        return feats
    size = ys[0].shape[2:]
    ups = [F.interpolate(t, size=size, mode="trilinear", align_corners=True) for t in ys[1:]]
    feats = torch.cat([ys[0]] + ups, 1)
    if final_fuse == "conat_conv":  # sic, hrnet3d.py:41
        feats = _conv(sd, p + ".final_conv", feats)
    return feats


def feature_adaption(sd, p, x, deformable_groups=4):
    """FeatureAdaption.forward -- center_head.py:24-62 -- applied per (frame, z) slice of a 5-D feature (Z folded into the
    batch; the reference's module is 2-D and its DCNSepHead cannot run on the 5-D feature, appendix 4):
    offset = Conv2d 1x1 (zero-initialised weight) -> DCNv1 3x3, pad 1, deformable_groups 4 -> ReLU.
    The deformable convolution is oracle/dcn_ref.py -- PARITY UNPINNED by the reference (see that file's header)."""
    from . import dcn_ref
    b, c, d, h, w = x.shape
    x2 = x.permute(0, 2, 1, 3, 4).reshape(b * d, c, h, w)
    off = F.conv2d(x2, sd[p + ".conv_offset.weight"], sd[p + ".conv_offset.bias"])
    y2 = F.relu(dcn_ref.deform_conv2d(x2, off, sd[p + ".conv_adaption.weight"], 1, 1, 1, 1, deformable_groups))
    return y2.reshape(b, d, c, h, w).permute(0, 2, 1, 3, 4)


def sep_head(sd, p, x, heads):
    """SepHead.forward -- pose_heads/center_head.py:66-109 (final_kernel=3, 2 convs per head).  With the DCN head's
    parameters present (dcn_head=True, DCNSepHead.forward :156-163) the heat-map tower reads feature_adapt_cls(x) and the
    other towers feature_adapt_reg(x)."""
    out = {}
    dcn = (p + ".feature_adapt_cls.conv_adaption.weight") in sd
    adapted = {}
    for name in heads:
        src = x
        if dcn:
            which = "cls" if name == "hm" else "reg"
            if which not in adapted:
                adapted[which] = feature_adaption(sd, f"{p}.feature_adapt_{which}", x)
            src = adapted[which]
        t = F.relu(_conv(sd, f"{p}.{name}.0", src, 1, 1))
        out[name] = _conv(sd, f"{p}.{name}.2", t, 1, 1)
    return out


def center_head(sd, x, heads=("reg", "hm"), p="pose_head", lidar=None):
    """CenterHead.forward -- center_head.py:232-238; shared_conv :203-211.
    lidar (this repo's two-stream fusion, BASELINE config 5 / SURVEY 8f N3; no reference counterpart): the dense LiDAR voxel
    grid [B, C_l, Z, Y, X] is concatenated with the radar feature along the channels in front of the towers."""
    if (p + ".shared_conv.1.weight") in sd:
        x = F.relu(_conv(sd, p + ".shared_conv.1", _gn(sd, p + ".shared_conv.0", x), 1, 1))
    if lidar is not None:
        x = torch.cat([x, lidar.to(x.dtype)], dim=1)
    return [sep_head(sd, p + ".tasks.0", x, heads)], x


def transpose_and_gather(feat, ind):
    """core/utils/center_utils.py:103-117."""
    b, c = feat.shape[:2]
    feat = feat.permute(0, 2, 3, 4, 1).reshape(b, -1, c)
    return feat.gather(1, ind.unsqueeze(2).expand(b, ind.shape[1], c))


def fast_focal_loss(out, target, ind, mask, cat):
    """losses/centernet_loss.py:34-54."""
    mask = mask.float()
    neg = (torch.log(1 - out) * out.pow(2) * (1 - target).pow(4)).sum()
    pos_pred = transpose_and_gather(out, ind).gather(2, cat.unsqueeze(2))
    num_pos = mask.sum()
    pos = (torch.log(pos_pred) * (1 - pos_pred).pow(2) * mask.unsqueeze(2)).sum()
    if num_pos == 0:
        return -neg
    return -(pos + neg) / num_pos


def reg_loss(output, mask, ind, target):
    """losses/centernet_loss.py:17-24."""
    pred = transpose_and_gather(output, ind)
    mask = mask.float().unsqueeze(2)
    loss = F.l1_loss(pred * mask, target * mask, reduction="none") / (mask.sum() + 1e-4)
    return loss.transpose(2, 0).sum(dim=2).sum(dim=1)


def center_head_loss(preds, example, weight, code_weights):
    """CenterHead.loss -- center_head.py:240-270 (one task).  `preds` hm is raw logits."""
    rets = defaultdict(list)
    for t, pd in enumerate(preds):
        hm = torch.clamp(torch.sigmoid(pd["hm"]), min=1e-4, max=1 - 1e-4)
        hm_loss = fast_focal_loss(hm, example["hm"][t], example["ind"][t], example["mask"][t], example["cat"][t])
        rl = reg_loss(pd["reg"], example["mask"][t], example["ind"][t], example["anno_pose"][t])
        loc = (rl * rl.new_tensor(code_weights)).sum()
        for k, v in dict(loss=hm_loss + weight * loc, hm_loss=hm_loss.detach(), loc_loss=loc,
                         loc_loss_elem=rl.detach(), num_positive=example["mask"][t].float().sum()).items():
            rets[k].append(v)
    return rets


def center_head_predict(preds, test_cfg, metas=None):
    """CenterHead.predict + post_processing -- center_head.py:272-360.

    test_cfg: dict with out_size_factor (z,y,x), voxel_size (x,y,z), pc_range (x,y,z), score_threshold.
    """
    pd = preds[0]
    hm = torch.sigmoid(pd["hm"].permute(0, 2, 3, 4, 1))
    reg = pd["reg"].permute(0, 2, 3, 4, 1)
    b, H, W, L, ncls = hm.shape
    reg = reg.reshape(b, H * W * L, -1)
    hm = hm.reshape(b, H * W * L, ncls)
    nk = reg.shape[-1] // 3
    zs, ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), torch.arange(L), indexing="ij")
    zs, ys, xs = [t.reshape(1, -1, 1).to(hm) for t in (zs, ys, xs)]
    osf, vs, pr = test_cfg["out_size_factor"], test_cfg["voxel_size"], test_cfg["pc_range"]
    pts = []
    for i in range(nk):
        pts += [(xs + reg[:, :, 3 * i:3 * i + 1]) * osf[2] * vs[0] + pr[0],
                (ys + reg[:, :, 3 * i + 1:3 * i + 2]) * osf[1] * vs[1] + pr[1],
                (zs + reg[:, :, 3 * i + 2:3 * i + 3]) * osf[0] * vs[2] + pr[2]]
    pts = torch.cat(pts, dim=2)
    out = []
    for n in range(b):
        kps = []
        if nk == 1:  # one argmax per heat-map channel (:341-347)
            for c in range(ncls):
                ind = torch.argmax(hm[n, :, c])
                score = hm[n, ind, c]
                if score > test_cfg["score_threshold"]:
                    kps.append((c, *pts[n, ind].tolist(), score.item()))
        else:  # one argmax, 45 regressed coords (:348-355)
            ind = torch.argmax(hm[n, :, 0])
            score = hm[n, ind, 0].item()
            pose = pts[n, ind].tolist()
            if score > test_cfg["score_threshold"]:
                kps.append((0, *pose[:3], score))
            for i in range(1, 15):
                kps.append((i, *pose[3 * i:3 * i + 3], score))
        out.append({"keypoints": kps, "metadata": None if metas is None else metas[n]})
    return out


def radar_pose_net(sd, example, final_fuse, weight, code_weights, return_loss=True, test_cfg=None):
    """RadarPoseNet.forward -- detectors/radar_pose_net.py:26-46 (reader is identity, radar_encoder.py:15-17)."""
    ex = dict(example["rdr"])
    feats = hrnet3d(sd, ex["rdr_tensor"], final_fuse)
    preds, _ = center_head(sd, feats, lidar=ex.get("lidar_grid"))
    if return_loss:
        return center_head_loss(preds, ex, weight, code_weights)
    return center_head_predict(preds, test_cfg, example.get("meta"))


# ----------------------------------------------------------------------------
# label synthesis -- core/utils/center_utils.py:67-91, datasets/pipelines/pose.py:206-254
# ----------------------------------------------------------------------------
def gaussian3d(shape, sigma=1.0):
    """center_utils.py:67-72 -- note the (2 sigma^2)^(3/2) exponent denominator (sic)."""
    m, n, p = [(s - 1.0) / 2.0 for s in shape]
    z, y, x = np.ogrid[-m:m + 1, -n:n + 1, -p:p + 1]
    h = np.exp(-(x * x + y * y + z * z) / (2 * sigma * sigma) ** (3 / 2))
    h[h < np.finfo(h.dtype).eps * h.max()] = 0
    return h


def draw_gaussian3d(hm, center, radius, k=1):
    """center_utils.py:74-91.  hm [Z,Y,X]; center (x,y,z) ints."""
    d = 2 * radius + 1
    g = gaussian3d((d, d, d), sigma=d / 6)
    x, y, z = int(center[0]), int(center[1]), int(center[2])
    Z, Y, X = hm.shape
    x0, x1 = min(x, radius), min(X - x, radius + 1)
    y0, y1 = min(y, radius), min(Y - y, radius + 1)
    z0, z1 = min(z, radius), min(Z - z, radius + 1)
    mh = hm[z - z0:z + z1, y - y0:y + y1, x - x0:x + x1]
    mg = g[radius - z0:radius + z1, radius - y0:radius + y1, radius - x0:radius + x1]
    if min(mg.shape) > 0 and min(mh.shape) > 0:
        np.maximum(mh, mg * k, out=mh)
    return hm


def synth_example(batch, cin, dims, seed, one_hm=False, rank=0):
    """Synthetic batch with the layout of CRUW_POSE_Dataset.collate_fn."""
    Z, Y, X = dims
    g = torch.Generator().manual_seed(seed + rank)
    rdr = torch.relu(torch.randn(batch, cin, Z, Y, X, generator=g) * 0.5 + 0.1)
    ncls, nreg, radius = (1, 45, 2) if one_hm else (15, 3, 1)
    hm = np.zeros((batch, ncls, Z, Y, X), np.float32)
    ind = np.zeros((batch, ncls), np.int64)
    for b in range(batch):
        for c in range(ncls):
            cz = int(torch.randint(0, Z, (1,), generator=g))
            cy = int(torch.randint(0, Y, (1,), generator=g))
            cx = int(torch.randint(0, X, (1,), generator=g))
            draw_gaussian3d(hm[b, c], (cx, cy, cz), radius)
            ind[b, c] = cz * Y * X + cy * X + cx
    if one_hm:
        anno = torch.rand(batch, 1, 45, generator=g) * 16 - 8
    else:
        anno = torch.rand(batch, 15, 3, generator=g)
    ex = dict(rdr_tensor=rdr, hm=[torch.from_numpy(hm)], ind=[torch.from_numpy(ind)],
              mask=[torch.ones(batch, ncls, dtype=torch.uint8)],
              cat=[torch.arange(ncls).repeat(batch, 1)], anno_pose=[anno])
    return {"rdr": ex, "meta": [{"seq": "synth", "frame": b} for b in range(batch)]}


# ----------------------------------------------------------------------------
# evaluation -- eval_util.py:5-10, datasets/cruw_pose.py:277-311
# ----------------------------------------------------------------------------
def pjpe(pred, gt):
    pred = pred - pred[:1]
    gt = gt - gt[:1]
    return np.linalg.norm(pred - gt, axis=-1)


def abs_pjpe(pred, gt):
    return np.linalg.norm(pred - gt, axis=-1)


# ----------------------------------------------------------------------------
# train-step rule -- restated from the formulas; pinned (round 6) against steps captured from the reference's own OptimWrapper + OneCycle
# (tests/golden/gen_golden_optim.py, tests/test_oracle_golden.py::test_train_step_rule_against_the_reference_optimiser)
#   torchie/apis/train.py:157-174, solver/fastai_optim.py:121-175,
#   solver/learning_schedules_fastai.py:53-95, trainer/hooks/optimizer.py:14-24
# ----------------------------------------------------------------------------
def _cos(start, end, pct):
    return end + (start - end) / 2 * (np.cos(np.pi * pct) + 1)


def one_cycle(step, total_step, lr_max, moms=(0.95, 0.85), div_factor=10.0, pct_start=0.4):
    """OneCycle(LRSchedulerStep).step -- learning_schedules_fastai.py:53-95.  Returns (lr, beta1)."""
    a1 = int(total_step * pct_start)
    low = lr_max / div_factor
    if step >= a1:
        pct = (step - a1) / (total_step - a1)
        return float(_cos(lr_max, low / 1e4, pct)), float(_cos(moms[1], moms[0], pct))
    pct = step / a1
    return float(_cos(low, lr_max, pct)), float(_cos(moms[0], moms[1], pct))


class AdamTrueWD:
    """OptimWrapper(true_wd=True, bn_wd=True) around torch.optim.Adam(betas=(mom,0.99)) -- fastai_optim.py:154-172."""

    def __init__(self, params, wd=0.01, beta2=0.99, eps=1e-8):
        self.params = list(params)
        self.wd, self.beta2, self.eps = wd, beta2, eps
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.t = [0 for _ in self.params]

    @torch.no_grad()
    def step(self, lr, beta1, max_norm=35.0):
        grads = [p.grad for p in self.params if p.grad is not None]
        total = torch.sqrt(sum((g.float() ** 2).sum() for g in grads))
        coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)  # clip_grad_norm_
        for i, p in enumerate(self.params):
            p.mul_(1 - self.wd * lr)  # decoupled decay hits every param, with or without a grad
            if p.grad is None:
                continue
            g = p.grad * coef
            self.t[i] += 1
            self.m[i].mul_(beta1).add_(g, alpha=1 - beta1)
            self.v[i].mul_(self.beta2).addcmul_(g, g, value=1 - self.beta2)
            bc1 = 1 - beta1 ** self.t[i]
            bc2 = 1 - self.beta2 ** self.t[i]
            denom = (self.v[i].sqrt() / np.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(self.m[i], denom, value=-lr / bc1)
        return float(total)


# ----------------------------------------------------------------------------
# parameter schema + seeded init recipe (used by golden generation and tests)
# ----------------------------------------------------------------------------
def param_shapes(arch, final_in, final_out, head_in, heads, dcn_head=False):
    """Names/shapes of the reference module tree (checked against the reference import in gen_golden.py).
    dcn_head: plus the two FeatureAdaption modules of DCNSepHead (center_head.py:44-57, 125-135; 2-D shapes)."""
    a = ARCHS[arch]
    ch = a["channels"]
    sd = OrderedDict()
    bb = "backbone.backbone"

    def gn(p, c):
        sd[p + ".weight"] = (c,)
        sd[p + ".bias"] = (c,)

    def block(p, cin, cout):
        if cin != cout:
            sd[p + ".conv1.weight"] = (cout, cin, 1, 1, 1)
            sd[p + ".conv1.bias"] = (cout,)
        for c in ("conv2", "conv3"):
            gn(f"{p}.{c}.groupnorm", cout)
            sd[f"{p}.{c}.conv.weight"] = (cout, cout, 3, 3, 3)

    def seq(p, cin, cout, k):
        gn(p + ".0", cin)
        sd[p + ".1.weight"] = (cout, cin, k, k, k)

    block(bb + ".layer1", a["inplanes"], ch[0])
    for stage in (2, 3, 4):
        nb = stage
        seq(f"{bb}.transition{stage - 1}.{stage - 1}.0", ch[nb - 2], ch[nb - 1], 3)
        p = f"{bb}.stage{stage}.0"
        for i in range(nb):
            block(f"{p}.branches.{i}.0", ch[i], ch[i])
        for i in range(nb):
            for j in range(nb):
                if j > i:
                    seq(f"{p}.fuse_layers.{i}.{j}", ch[j], ch[i], 1)
                elif j < i:
                    for k in range(i - j):
                        seq(f"{p}.fuse_layers.{i}.{j}.{k}", ch[j], ch[i] if k == i - j - 1 else ch[j], 3)
    if final_in != final_out:
        sd["backbone.final_conv.weight"] = (final_out, final_in, 1, 1, 1)
        sd["backbone.final_conv.bias"] = (final_out,)
    if dcn_head:
        for which in ("cls", "reg"):
            p = f"pose_head.tasks.0.feature_adapt_{which}"
            sd[p + ".conv_offset.weight"] = (72, head_in, 1, 1)
            sd[p + ".conv_offset.bias"] = (72,)
            sd[p + ".conv_adaption.weight"] = (head_in, head_in, 3, 3)
    for name, ncls in heads.items():
        p = f"pose_head.tasks.0.{name}"
        sd[p + ".0.weight"] = (32, head_in, 3, 3, 3)
        sd[p + ".0.bias"] = (32,)
        sd[p + ".2.weight"] = (ncls, 32, 3, 3, 3)
        sd[p + ".2.bias"] = (ncls,)
    return sd


MODEL_CONFIGS = {
    # name: (arch, final_conv_in, final_conv_out, final_fuse, heads, weight, code_weights)
    "hr3d": ("hr_tiny_feat32_zyx_l4", 32, 32, "top", OrderedDict(reg=3, hm=15), 0.2, [1.0, 1.5, 2.0]),
    "hr3d_one_hm": ("hr_tiny_feat32_zyx_l4", 192, 128, "conat_conv", OrderedDict(reg=45, hm=1), 0.5, [1.0] * 45),
    "hr3d_one_hm_doppler": ("hr_tiny_feat32_zyx_l4_in32", 192, 128, "conat_conv", OrderedDict(reg=45, hm=1), 0.5, [1.0] * 45),
    "hr3d_one_hm_doppler_phase": ("hr_tiny_feat64_zyx_l4_in64", 384, 256, "conat_conv", OrderedDict(reg=45, hm=1), 0.5, [1.0] * 45),
}


def seeded_state_dict(shapes, seed=0, dtype=torch.float32):
    """Deterministic weights by NAME (independent of module construction order).

    conv weights ~ U(+-1/sqrt(fan_in)); biases ~ U(+-0.1); GN weight ~ 1+U(+-0.2), GN bias ~ U(+-0.2);
    heat-map final bias = -2.19 (center_head.py:94-95).
    """
    sd = OrderedDict()
    for i, (name, shape) in enumerate(shapes.items()):
        g = torch.Generator().manual_seed(seed * 100003 + i)
        u = torch.rand(*shape, generator=g, dtype=torch.float64) * 2 - 1
        is_gn = len(shapes[name.rsplit(".", 1)[0] + ".weight"]) == 1
        if len(shape) == 5:
            fan_in = shape[1] * shape[2] * shape[3] * shape[4]
            t = u / np.sqrt(fan_in) * np.sqrt(3.0)
        elif name.endswith("hm.2.bias"):
            t = torch.full(shape, -2.19, dtype=torch.float64)
        elif is_gn and name.endswith(".weight"):
            t = 1 + u * 0.2
        elif is_gn:
            t = u * 0.2
        else:
            t = u * 0.1
        sd[name] = t.to(dtype)
    return sd
