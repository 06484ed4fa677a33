"""The four shipped model configurations (configs/cruw_pose/{hr3d,hr3d_one_hm,hr3d_one_hm_doppler,
hr3d_one_hm_doppler_phase}.py :54-82 of each) as plain data, for callers that have no det3d config file at hand
(bench.py, tests on the GPU box).  `model_dict(name)` returns the same `model=dict(type="RadarPoseNet", ...)` the
reference config files define, so it can be fed to build_detector unchanged."""
from collections import OrderedDict

JOINTS = ["Pelvis", "Right_Hip", "Right_Knee", "Right_Ankle", "Left_Hip", "Left_Knee", "Left_Ankle", "Thomx", "Head",
          "Left_Shoulder", "Left_Elbow", "Left_Wrist", "Right_Shoulder", "Right_Elbow", "Right_Wrist"]

NATIVE_DIMS = (16, 64, 160)            # Z, Y, X after the ROI crop (cruw_pose.py:140-146, hr3d.py:38)
VOXEL_SIZE = [0.0453125, 0.15703125, 0.3625]   # x, y, z  (hr3d.py:39)
ROI1 = {"z": [-1.0875000000000021, 4.7125], "y": [-5.0250000000000234, 5.024999999999931], "x": [0.7703125, 8.0203125]}

_TABLE = {
    # name: backbone_cfg, cin, final_conv_in, final_conv_out, final_fuse, n_hm, n_reg, weight, lr_max, batch
    "hr3d": ("hr_tiny_feat32_zyx_l4", 1, 32, 32, "top", 15, 3, 0.2, 1e-3, 16),
    "hr3d_one_hm": ("hr_tiny_feat32_zyx_l4", 1, 192, 128, "conat_conv", 1, 45, 0.5, 2e-3, 8),
    "hr3d_one_hm_doppler": ("hr_tiny_feat32_zyx_l4_in32", 32, 192, 128, "conat_conv", 1, 45, 0.5, 2e-3, 8),
    "hr3d_one_hm_doppler_phase": ("hr_tiny_feat64_zyx_l4_in64", 64, 384, 256, "conat_conv", 1, 45, 0.5, 2e-3, 1),
}
NAMES = list(_TABLE)
# BASELINE config 4: a shipped configuration + `dcn_head=True` (the DCN head of center_head.py:111-163 with Z folded into the
# batch; the reference's own DCNSepHead cannot run on the 5-D feature, SURVEY appendix 4 -- parity unpinned by construction)
DCN_VARIANTS = {"hr3d_dcn": "hr3d"}
# BASELINE config 5 (two-stream fusion): a shipped configuration whose head towers read the radar feature
# CONCATENATED with the dense-ified LiDAR voxel grid (mean point features per voxel: x, y, z, intensity -- DynamicVoxelEncoder,
# readers/dynamic_voxel_encoder.py:69-101 -> rt_pose_amd.lidar.DynamicVoxelEncoder.to_dense).  The reference ships no fusion
# detector (voxelnet.py:47-49 calls a missing backbone), so the composition is this repo's and parity-unpinned by construction.
LIDAR_VARIANTS = {"hr3d_lidar": ("hr3d", 4)}


def spec(name):
    dcn = name in DCN_VARIANTS
    base, lidar_c = LIDAR_VARIANTS.get(name, (name, 0))
    arch, cin, fin, fout, fuse, nhm, nreg, weight, lr_max, batch = _TABLE[DCN_VARIANTS.get(base, base)]
    cw = [1.0, 1.5, 2.0] if nreg == 3 else [1.0] * nreg
    return dict(arch=arch, cin=cin, final_conv_in=fin, final_conv_out=fout, final_fuse=fuse,
                heads=OrderedDict(reg=nreg, hm=nhm), weight=weight, code_weights=cw, lr_max=lr_max, batch=batch, dcn_head=dcn,
                lidar_channels=lidar_c)


def model_dict(name):
    s = spec(name)
    nhm = s["heads"]["hm"]
    tasks = [dict(num_class=nhm, class_names=JOINTS[:nhm])]
    return dict(
        type="RadarPoseNet", pretrained=None, reader=dict(type="RadarFeatureNet"),
        backbone=dict(type="HRNet3D", backbone_cfg=s["arch"], final_conv_in=s["final_conv_in"],
                      final_conv_out=s["final_conv_out"], final_fuse=s["final_fuse"], ds_factor=1),
        pose_head=dict(type="CenterHead", tasks=tasks, in_channels=s["final_conv_out"],
                       share_conv_channel=s["final_conv_out"], dataset="cruw_pose", weight=s["weight"],
                       code_weights=s["code_weights"], common_heads={"reg": (s["heads"]["reg"], 2)}, dcn_head=s["dcn_head"],
                       lidar_channels=s["lidar_channels"]),
        neck=None)


def test_cfg():
    return dict(post_center_limit_range=[ROI1["x"][0], ROI1["y"][0], ROI1["z"][0], ROI1["x"][1], ROI1["y"][1], ROI1["z"][1]],
                score_threshold=0.0, pc_range=[ROI1["x"][0], ROI1["y"][0], ROI1["z"][0]], out_size_factor=[1, 1, 1],
                voxel_size=VOXEL_SIZE, input_type="rdr_cube")


def backbone_shapes(arch, final_conv_in, final_conv_out):
    """HRNet3D(backbone_cfg=arch, final_conv_in, final_conv_out): reference state_dict names -> shapes (hrnet3d.py:11-20 around
    hr_util/hr3d.py's HighResolution3DNet)."""
    from .net import ARCH_TABLES
    ch, cin = ARCH_TABLES[arch]["channels"], ARCH_TABLES[arch]["inplanes"]
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
            gn("%s.%s.groupnorm" % (p, c), cout)
            sd["%s.%s.conv.weight" % (p, c)] = (cout, cout, 3, 3, 3)

    def seq(p, cin, cout, k):
        gn(p + ".0", cin)
        sd[p + ".1.weight"] = (cout, cin, k, k, k)

    block(bb + ".layer1", cin, ch[0])
    for stage in (2, 3, 4):
        nb = stage
        seq("%s.transition%d.%d.0" % (bb, stage - 1, stage - 1), ch[nb - 2], ch[nb - 1], 3)
        p = "%s.stage%d.0" % (bb, stage)
        for i in range(nb):
            block("%s.branches.%d.0" % (p, i), ch[i], ch[i])
        for i in range(nb):
            for j in range(nb):
                if j > i:
                    seq("%s.fuse_layers.%d.%d" % (p, i, j), ch[j], ch[i], 1)
                elif j < i:
                    for k in range(i - j):
                        seq("%s.fuse_layers.%d.%d.%d" % (p, i, j, k), ch[j], ch[i] if k == i - j - 1 else ch[j], 3)
    if final_conv_in != final_conv_out:
        sd["backbone.final_conv.weight"] = (final_conv_out, final_conv_in, 1, 1, 1)
        sd["backbone.final_conv.bias"] = (final_conv_out,)
    return sd


def param_shapes(name):
    """Reference state_dict names -> shapes (det3d module tree; checked against tests/golden/param_schema.json)."""
    s = spec(name)
    sd = backbone_shapes(s["arch"], s["final_conv_in"], s["final_conv_out"])
    if s["dcn_head"]:   # FeatureAdaption x 2 (center_head.py:44-57, 125-135); before the towers: keeps every view 16-B aligned
        c = s["final_conv_out"]
        for which in ("cls", "reg"):
            p = "pose_head.tasks.0.feature_adapt_%s" % which
            sd[p + ".conv_offset.weight"] = (4 * 18, c, 1, 1)
            sd[p + ".conv_offset.bias"] = (4 * 18,)
            sd[p + ".conv_adaption.weight"] = (c, c, 3, 3)
    for hname, ncls in s["heads"].items():
        p = "pose_head.tasks.0.%s" % hname
        sd[p + ".0.weight"] = (32, s["final_conv_out"] + s["lidar_channels"], 3, 3, 3)
        sd[p + ".0.bias"] = (32,)
        sd[p + ".2.weight"] = (ncls, 32, 3, 3, 3)
        sd[p + ".2.bias"] = (ncls,)
    return sd
