"""HRRadarPose as a static graph: mirrors the reference module tree op for op.

  HighResolution3DNet.forward   det3d/models/backbones/hr_util/hr3d.py:373-399 (transitions :286-331)
  HighResolutionModule.forward  hr3d.py:205-229 (fuse layers :135-200)
  ResNetBlock.forward           hr_util/common.py:138-148
  HRNet3D.forward               det3d/models/backbones/hrnet3d.py:29-43
  CenterHead / SepHead          det3d/models/pose_heads/center_head.py:66-109, 232-238

Parameter names are the reference state_dict keys (so reference checkpoints load unchanged).
"""
from .graph import Graph

# det3d/models/backbones/hrnet3D_config.py:85-187 -- tables selected by configs/cruw_pose/*.py
ARCH_TABLES = {
    "hr_tiny_feat32_zyx_l4": dict(inplanes=1, channels=(32, 32, 64, 64)),
    "hr_tiny_feat32_zyx_l4_in32": dict(inplanes=32, channels=(32, 32, 64, 64)),
    "hr_tiny_feat64_zyx_l4_in64": dict(inplanes=64, channels=(64, 64, 128, 128)),
}


def down_dims(dims):
    return tuple((s + 2 - 3) // 2 + 1 for s in dims)


def _resblock(g: Graph, p, x, tag):
    """GN->conv3->ReLU, GN->conv3, +x, ReLU   (conv1 is Identity here: Cin == Cout)."""
    t = g.conv(tag + ".c2", x, p + ".conv2.conv.weight", gn=(p + ".conv2.groupnorm.weight", p + ".conv2.groupnorm.bias"),
               relu=True)
    return g.conv(tag + ".c3", t, p + ".conv3.conv.weight", gn=(p + ".conv3.groupnorm.weight", p + ".conv3.groupnorm.bias"),
                  relu=True, residual=x)


def _seq(g: Graph, p, x, tag, ks, stride, relu, want_stats=True):
    """nn.Sequential(GroupNorm, Conv3d(bias=False)[, ReLU])."""
    return g.conv(tag, x, p + ".1.weight", gn=(p + ".0.weight", p + ".0.bias"), ks=ks, stride=stride, relu=relu,
                  want_stats=want_stats)


def build_backbone(g: Graph, x_f32, arch, dims, prefix="backbone.backbone", live_rows_last=None):
    """Returns the list of stage-4 outputs.  live_rows_last: how many fuse rows of the LAST stage to compute
    (final_fuse='top' only consumes row 0; rows 1..3 of stage4 are dead code in the reference)."""
    a = ARCH_TABLES[arch]
    ch = a["channels"]
    p = prefix + ".layer1"
    if a["inplanes"] == 1:
        t0 = g.stem("l1.c1", x_f32, dims, p + ".conv1.weight", p + ".conv1.bias")
    else:
        xin = g.pack("l1.in", x_f32, a["inplanes"], dims)
        if a["inplanes"] != ch[0] or (p + ".conv1.weight") in g.params:
            t0 = g.conv("l1.c1", xin, p + ".conv1.weight", bname=p + ".conv1.bias", ks=1)
        else:
            t0 = xin
    ys = [_resblock(g, p, t0, "l1")]
    for stage in (2, 3, 4):
        nb = stage
        rows = nb if (stage < 4 or live_rows_last is None) else live_rows_last
        tp = "%s.transition%d.%d.0" % (prefix, stage - 1, stage - 1)
        last = ys[-1]
        ys = hr_module(g, "%s.stage%d.0" % (prefix, stage), stage, ys, lambda: _seq(g, tp, last, "t%d" % (stage - 1), 3, 2, True), rows)
    return ys


def hr_module(g: Graph, sp, stage, ys, new_branch, rows=None):
    """One HighResolutionModule (hr3d.py:205-229) of `stage` branches: ys = the stage's inputs for the existing branches,
    new_branch() -> the input of the new lowest branch (the transition conv of HighResolution3DNet.forward, hr3d.py:386,394; called
    after the existing branches' blocks are created).  -> the `rows` first fuse rows (default: all)."""
    nb = stage
    ch = [y.c_real for y in ys]
    # Creation order = launch order, and the backward sweep runs it in reverse: the existing branches' blocks are created
    # BEFORE the transition conv, so that a branch block's first conv -- whose data gradient runs on the LDS-tiled kernel --
    # is the last gradient contribution to the previous stage's output and can absorb the others (graph.ConvOp._fusable)
    xs = [_resblock(g, "%s.branches.%d.0" % (sp, i), ys[i], "s%d.b%d" % (stage, i)) for i in range(nb - 1)]
    new = new_branch()
    xs.append(_resblock(g, "%s.branches.%d.0" % (sp, nb - 1), new, "s%d.b%d" % (stage, nb - 1)))
    rows = nb if rows is None else rows
    # rows are CREATED last-to-first: the stride-2 conv that starts a lower row's chain from branch 0 then precedes row 0's
    # fuse node, i.e. it is the first-created consumer of the branch-0 output and its (tiled) data gradient can absorb the
    # other gradient contributions to that full-resolution tensor (graph.ConvOp._fusable)
    out = [None] * rows
    for i in reversed(range(rows)):
        g.group = ("fuse%d" % stage, i)   # (graph.Graph.build_backward may sweep a fuse block's rows in another order)
        terms = []
        for j in range(nb):
            if j == i:
                terms.append(xs[j])
            elif j > i:
                terms.append(_seq(g, "%s.fuse_layers.%d.%d" % (sp, i, j), xs[j], "s%d.f%d%d" % (stage, i, j), 1, 1, False,
                                  want_stats=False))   # feeds only the fuse sum
            else:
                t = xs[j]
                for k in range(i - j):
                    t = _seq(g, "%s.fuse_layers.%d.%d.%d" % (sp, i, j, k), t, "s%d.f%d%d.%d" % (stage, i, j, k), 3, 2,
                             relu=(k != i - j - 1), want_stats=(k != i - j - 1))
                terms.append(t)
        out[i] = g.fuse("s%d.row%d" % (stage, i), terms, relu=True, want_stats=stage < 4)   # (stage-4 rows feed no GroupNorm)
    g.group = None
    return out


def build_hrnet3d(g: Graph, x_f32, arch, dims, final_fuse, prefix="backbone"):
    ch = ARCH_TABLES[arch]["channels"]
    has_final = (prefix + ".final_conv.weight") in g.params
    if final_fuse == "top":
        ys = build_backbone(g, x_f32, arch, dims, prefix + ".backbone", live_rows_last=1)
        f = ys[0]
        if has_final:
            f = g.conv("final", f, prefix + ".final_conv.weight", bname=prefix + ".final_conv.bias", ks=1,
                       want_stats=(prefix.replace("backbone", "pose_head") + ".shared_conv.1.weight") in g.params)
        return f
    ys = build_backbone(g, x_f32, arch, dims, prefix + ".backbone")
    if final_fuse != "conat_conv" or not has_final:
        # hrnet3d.py:37-43: any other value returns the plain concatenation cat(x0, up(x1), up(x2), up(x3)) -- final_conv, if the
        # constructor made one, is never applied (appendix quirk 5: 'conat_conv' is the only spelling that reaches it) -- and so does
        # 'conat_conv' when final_conv_in == final_conv_out made final_conv an Identity (hrnet3d.py:13-14)
        if (prefix.replace("backbone", "pose_head") + ".shared_conv.1.weight") in g.params and sum(ch) not in (32, 64, 128, 256):
            # GroupNorm(8, 192) -> Conv3d(192, share): the fold kernels' weight images come 32 / 64 / 128 / 256 input channels wide
            # (rtp_wgrad_fold reads whole 1-KB rows).  The concatenation itself and towers that read it directly are built.
            raise NotImplementedError("shared_conv over the %d-channel plain concatenation (final_fuse=%r): not built; use "
                                      "in_channels == share_conv_channel, or final_fuse='conat_conv'" % (sum(ch), final_fuse))
        return g.concat("final.cat", ys)
    return final_concat_conv(g, ys, prefix)


def final_concat_conv(g: Graph, ys, prefix="backbone"):
    """HRNet3D.forward's 'conat_conv' fuse (hrnet3d.py:37-42): cat(x0, up(x1), up(x2), up(x3)) -> final_conv 1x1x1 (with bias)
    ==  sum_j up(conv1x1_j(x_j))  -- both ops are linear and the upsample acts per channel -- so the 192-channel concatenation is
    never materialised: one 1x1x1 conv per branch over its column block of the weight, then one fuse row."""
    ch = [y.c_real for y in ys]
    total = sum(ch)
    off, terms = 0, []
    for j, y in enumerate(ys):
        terms.append(g.conv("final.%d" % j, y, prefix + ".final_conv.weight",
                            bname=(prefix + ".final_conv.bias") if j == 0 else None, ks=1,
                            w_ci_total=total, w_ci_off=off, ci_real=ch[j], want_stats=False))
        off += ch[j]
    return g.fuse("final.sum", terms, relu=False,
                  want_stats=(prefix.replace("backbone", "pose_head") + ".shared_conv.1.weight") in g.params)


def build_head(g: Graph, feats, heads, prefix="pose_head", lidar=None):
    """SepHead with head_conv=32, final_kernel=3 (center_head.py:223): Conv3d(C,32,3)+ReLU -> Conv3d(32,classes,3).
    lidar = (activation, real channels): the dense LiDAR grid of the two-stream fusion variant (configs.LIDAR_VARIANTS); the
    towers' first conv then reads concat(feats, lidar) -- as two input-channel slices, the concatenation is never built."""
    if (prefix + ".shared_conv.1.weight") in g.params:
        feats = g.conv("shared", feats, prefix + ".shared_conv.1.weight",
                       gn=(prefix + ".shared_conv.0.weight", prefix + ".shared_conv.0.bias"), relu=True)
    out = {}
    # dcn_head (BASELINE config 4): separate deformable feature adaptions for the heat-map and the regression towers
    # (DCNSepHead.forward, center_head.py:156-163), each on [B*Z, C, Y, X]
    adapt = {}
    if (prefix + ".tasks.0.feature_adapt_cls.conv_adaption.weight") in g.params:
        adapt = {"hm": g.dcn_adapt("adapt.cls", feats, prefix + ".tasks.0.feature_adapt_cls"),
                 "reg": g.dcn_adapt("adapt.reg", feats, prefix + ".tasks.0.feature_adapt_reg")}
    src = feats
    # the two towers' first convs over a 128- / 256-channel feature: one 64-wide launch per 64-channel slice (Graph.conv_pair)
    paired = {}
    hl = list(heads)
    if lidar is None and not adapt and len(hl) == 2:
        ps = ["%s.tasks.0.%s" % (prefix, name) for name in hl]
        ts = g.conv_pair(["head.%s.0" % name for name in hl], src, [p + ".0.weight" for p in ps], [p + ".0.bias" for p in ps], relu=True)
        if ts is not None:
            paired = dict(zip(hl, ts))
    for name in heads:
        p = "%s.tasks.0.%s" % (prefix, name)
        feats = adapt.get(name, src)
        if name in paired:
            out[name] = g.conv("head.%s.2" % name, paired[name], p + ".2.weight", bname=p + ".2.bias", out_fp32=True)
            continue
        if lidar is not None:
            t = g.conv_cat("head.%s.0" % name, [feats, lidar[0]], [feats.c_real, lidar[1]], p + ".0.weight", bname=p + ".0.bias",
                           relu=True)
        else:
            t = g.conv("head.%s.0" % name, feats, p + ".0.weight", bname=p + ".0.bias", relu=True, want_stats=False)
        out[name] = g.conv("head.%s.2" % name, t, p + ".2.weight", bname=p + ".2.bias", out_fp32=True)
    return out
