"""Whole-model execution engine: forward, losses, backward and the optimiser step as one static launch plan.

Mirrors the reference's train step semantics:
  Trainer.train -> batch_processor_inline -> RadarPoseNet.forward(return_loss=True)
      det3d/torchie/trainer/trainer.py:399-434, det3d/models/detectors/radar_pose_net.py:36-46
  CenterHead.loss                      det3d/models/pose_heads/center_head.py:244-270
  OptimizerHook.after_train_iter       det3d/torchie/trainer/hooks/optimizer.py:14-24  (clip 35)
  OptimWrapper.step (true_wd) + Adam   det3d/solver/fastai_optim.py:154-172
  OneCycle                             det3d/solver/learning_schedules_fastai.py:53-95
Data parallel: one process per GPU, batch sharded by rank, ONE all-reduce of the flat fp32 gradient buffer
(replaces DDP buckets + the reference's redundant second all-reduce, core/utils/dist_utils.py:45-57).
"""
import math
from collections import OrderedDict

import os

import torch

from . import net
from .graph import Graph, View, pad_to
from .lanes import LanePlan, LANE_MAP, LANE_MAP_4


def one_cycle(step, total_step, lr_max, moms=(0.95, 0.85), div_factor=10.0, pct_start=0.4):
    """(lr, beta1) at `step` -- learning_schedules_fastai.py:53-95 (cosine up over pct_start, then down to lr_max/div/1e4)."""
    def cos(start, end, pct):
        return end + (start - end) / 2 * (math.cos(math.pi * pct) + 1)

    a1 = int(total_step * pct_start)
    low = lr_max / div_factor
    if step >= a1:
        pct = (step - a1) / max(1, total_step - a1)
        return cos(lr_max, low / 1e4, pct), cos(moms[1], moms[0], pct)
    pct = step / max(1, a1)
    return cos(low, lr_max, pct), cos(moms[0], moms[1], pct)


class FlatParams:
    """All parameters (reference state_dict names/shapes, fp32) as views of ONE flat buffer, plus flat grad/m/v."""

    def __init__(self, shapes: OrderedDict, alloc):
        self.shapes = OrderedDict((k, tuple(v)) for k, v in shapes.items())
        self.offsets, off = OrderedDict(), 0
        for k, s in self.shapes.items():
            self.offsets[k] = off
            off += int(torch.Size(s).numel())
        self.numel = off
        self.p = alloc((off,), "f32")
        self.g = alloc((off,), "f32")
        self.m = alloc((off,), "f32")
        self.v = alloc((off,), "f32")
        self.values = OrderedDict((k, self._view(self.p, k)) for k in self.shapes)
        self.grads = OrderedDict((k, self._view(self.g, k)) for k in self.shapes)

    def _view(self, flat, k):
        o = self.offsets[k]
        return flat[o:o + int(torch.Size(self.shapes[k]).numel())].view(self.shapes[k])

    def load_state_dict(self, sd):
        for k, v in self.values.items():
            v.copy_(sd[k].to(v.dtype))

    def state_dict(self):
        return OrderedDict((k, v.detach().clone()) for k, v in self.values.items())

    def runs(self, live_names):
        """Maximal contiguous [start, end, live] element ranges in buffer order."""
        out = []
        for k in self.shapes:
            o, n = self.offsets[k], int(torch.Size(self.shapes[k]).numel())
            live = k in live_names
            if out and out[-1][2] == live and out[-1][1] == o:
                out[-1][1] = o + n
            else:
                out.append([o, o + n, live])
        return [tuple(r) for r in out]


# Main-lane launches that give up a quarter of the CUs (192 workgroups instead of 256) while the side lanes' dependent chains are
# the critical path.  Measured (hr3d, B = 8, round 4): every LDS-tiled launch asks for all 256 CUs with all their LDS, so the lanes'
# kernels serialise as whole kernels and a side lane's chain of 8-20 small dependent launches advances ONE launch per main-kernel
# boundary -- the main lane then waits 150-440 us in front of each fuse row / fan-in for chains whose kernels add up to a fraction of
# that.  The tiled kernels are power- and bandwidth-limited rather than CU-limited (a full-resolution conv on 128 workgroups takes
# 91 us against 78 on 256), so leaving 64 CUs free costs them 5 % and lets the chains run BESIDE them: 5.77-5.88 -> 5.62-5.66 ms per
# step (-2.5 ... -3.5 %, five same-box pairs).  Forward convs and weight gradients of the full-resolution branch in stages 2-4, the
# conv3 data gradients of stages 3-4; data gradients elsewhere, 208 / 176 / 160 workgroups, the head and layer1: no better or worse.
# Results do not change (include/rtp.h: RtpConvGeom::wgs).  PlanOptions.width_hints = "" switches them off, any other value replaces them.
# The width is an explicit field of each launch's geometry, fixed when the plan is built (graph.Graph.with_width).
DEFAULT_WIDTH_HINTS = ";".join(["conv:s%d.b0=192" % s for s in (2, 3, 4)] + ["wgrad:s%d.b0=192" % s for s in (2, 3, 4)]
                               + ["dgrad:s4.b0.c3=192", "dgrad:s3.b0.c3=192"])


def parse_width_hints(spec):
    """"tag-prefix=workgroups;..." (e.g. "conv:s3.b0=192;wgrad:s4.b0=208") -> [(prefix, workgroups)] in order (first match wins)."""
    items = (spec or "").replace(",", ";").split(";")   # (',' inside RTP_PLAN, whose own separator is ';')
    return [(k.strip(), int(v)) for k, v in (item.rsplit("=", 1) for item in items if "=" in item) if int(v) > 0]


class PoseEngine:
    """HRRadarPose for a fixed (batch, Cin, dims): buffers + launch lists built once, replayed every step."""

    def __init__(self, backend, params, arch, final_fuse, heads, loss_weight, code_weights, batch, dims, train=True,
                 pgrads=None, test_cfg=None, max_objs=None, lidar_channels=0, early_flush=False, feature_channels=None, options=None):
        """feature_channels=C: a plan of the HEAD alone (CenterHead used without RadarPoseNet, center_head.py:232-270): the input is a
        feature [B, C, Z, Y, X] fp32 (load_features); in training mode the backward list ends with the feature's gradient, unpacked
        into feat_grad [B, C, Z, Y, X] fp32.  `arch` / `final_fuse` are not used then."""
        self.be, self.n, self.dims, self.train = backend, batch, tuple(dims), train
        self.heads = OrderedDict(heads)
        self.nreg, self.ncls = self.heads["reg"], self.heads["hm"]
        self.loss_weight = float(loss_weight)
        cin = net.ARCH_TABLES[arch]["inplanes"] if feature_channels is None else None
        be = backend
        # width hints: a launch's wgs / batch workgroups per sample, whatever the batch.  Measured (round 5, same box, hr3d, four-stream
        # map): B = 8 5.44 -> 5.26 ms per step, B = 16 9.44 -> 9.28 (1 694 -> 1 723 frames/s), B = 4 3.69 -> 3.56 (1 083 -> 1 123)
        from .options import PlanOptions
        opt = self.options = options if options is not None else PlanOptions.from_env()
        rules = parse_width_hints(DEFAULT_WIDTH_HINTS if opt.width_hints is None else opt.width_hints)
        g = self.graph = Graph(be, batch, params, train=train, pgrads=pgrads, width_rules=rules, options=opt)
        g.early_flush = bool(early_flush)   # two gradient buckets (trainer): one early flush of the deferred tail
        self.feat_in = self.feat_grad = None
        if feature_channels is None:
            self.x_in = g.input_f32("rdr", cin, dims)
            self.feats = net.build_hrnet3d(g, self.x_in, arch, dims, final_fuse)
        else:
            from .graph import pad_to as _pad
            self.x_in = None
            self.feat_in = g.input_f32("feats_in", feature_channels, dims)
            self.feats = g.act("feats", feature_channels, dims, c=_pad(feature_channels, 32), needs_grad=bool(train))
            g.emit_fwd(be.pack_ncdhw(self.feat_in, self.feats, feature_channels), g.lane_of(self.feats), [self.feat_in], [self.feats],
                       "pack:feats")
        # two-stream fusion (BASELINE config 5): the dense LiDAR voxel grid [B, C_l, Z, Y, X] fp32 enters beside the radar feature
        self.lidar_in = None
        lidar = None
        if lidar_channels:
            self.lidar_in = g.input_f32("lidar", lidar_channels, dims)
            lidar = (g.pack("lidar", self.lidar_in, lidar_channels, dims), lidar_channels)
        self.outs = net.build_head(g, self.feats, list(self.heads), lidar=lidar)
        self.fwd = list(g.forward_list())
        # the main lane's fuse-row feeders (the 1x1x1 convs of row 0) ahead of the other rows' chains on their FIFO lanes: neutral on hr3d
        # (5.608 -> 5.595 ms, the wait moves from fuse:s3.row0 to fuse:s4.row0), -0.6 % on the configs whose stage 4 keeps every row
        # (hr3d_one_hm_doppler 10.43 -> 10.36); options.fwd_row0_first = 0: creation order
        if opt.fwd_row0_first:
            from .lanes import main_row_first
            self.fwd = main_row_first(self.fwd)
        self.merged = []   # (tag_a, tag_b) whose algorithmic cost was re-accounted to a's kernel family (none in the default plan)
        # The two head towers (hm, reg) are independent chains of full-resolution launches on the main lane: they run pairwise in ONE
        # launch, each problem on half of every XCD's workgroups (lanes.merge_launches -> HipBackend.multi -> rtp_multi_*).  Alone a
        # full-resolution tiled kernel is limited by fixed costs, power and HBM rather than by CUs (91 us on 128 workgroups against 78
        # on 256), so two of them side by side finish well before two in a row: hr3d 5.56 -> 5.48 ms per step (-1.4 %, two same-box
        # pairs) with five pairs merged -- conv .0 / .2, dgrad .2, wgrad .0 / .2; dgrad:head.reg.0 adds dgrad:head.hm.0's result in
        # its epilogue and stays behind it.  (Merging a side lane's level-1 convs INTO the main lane's full-resolution launches of the same
        # position -- round 4's RTP_MERGE -- measured 0.7-1 % slower in lane mode: work moved onto the critical path; removed.)
        # Any batch whose samples tile the chip's 256 workgroups (rtp_multi_end: n divides 256).  options.merge_head = 0: off.
        self._merge_head = bool(opt.merge_head) and hasattr(be, "multi") and 256 % batch == 0
        self.merged_head = []
        if self._merge_head:
            from .lanes import merge_launches
            self.fwd, done = merge_launches(self.fwd, be, [("conv:head.reg.0", "conv:head.hm.0"), ("conv:head.reg.2", "conv:head.hm.2")])
            self.merged_head = list(done)   # (both launches of a pair belong to the same kernel family: nothing to re-account)
        # lanes -> streams (lanes.py): four streams where the full-resolution weight-gradient lane is (nearly) empty, i.e. no channel-
        # sliced head whose weight gradients stay there; one stream per lane otherwise
        from .graph import SplitConvOp, CoSplitConvOp
        sliced_head = any(isinstance(op, (SplitConvOp, CoSplitConvOp)) for op in g.ops)
        # (any batch: B = 4 / 16 measured like B = 8 -- 3.83 -> 3.69 and 9.51 -> 9.44 ms per step for the map alone, round 5)
        self.lane_map = opt.int_list("lanes") if opt.lanes is not None else (LANE_MAP if sliced_head else LANE_MAP_4)
        self.fwd_plan = LanePlan(be, self.fwd, self.lane_map)
        self.bwd_plan = None
        self.use_lanes = True      # False: replay everything on the caller's stream in list order
        d, h, w = dims
        self.m = max_objs or self.ncls  # max_poses(1) * 15 key-points for hr3d, 1 for the one-heat-map variant
        # ---- decode (inference)
        self.dec_out = be.alloc((batch, self.ncls, 2 + self.nreg), "f32")
        self.dec = None
        if test_cfg is not None:
            self.set_test_cfg(test_cfg)
        # ---- losses + backward (training)
        self.loss_launches, self.bwd = [], []
        if train:
            self.tgt_hm = be.alloc((batch, self.ncls, d, h, w), "f32")
            self.tgt_ind = be.alloc((batch, self.m), "i64")
            self.tgt_mask = be.alloc((batch, self.m), "u8")
            self.tgt_cat = be.alloc((batch, self.m), "i64")
            self.tgt_pose = be.alloc((batch, self.m, self.nreg), "f32")
            self.code_w = be.alloc((self.nreg,), "f32")
            self.code_w.copy_(torch.tensor(list(code_weights), dtype=torch.float32))
            self.loss_hm = be.alloc((1,), "f32")
            self.loss_reg = be.alloc((self.nreg + 1,), "f32")
            hm, reg = self.outs["hm"], self.outs["reg"]
            ghm_c, greg_c = pad_to(pad_to(self.ncls, 16), 32), pad_to(pad_to(self.nreg, 16), 32)
            self.ghm = View(be.alloc((batch, d, h, w, ghm_c), "bf16"), batch, d, h, w, ghm_c, 0, ghm_c)
            self.greg = View(be.alloc((batch, d, h, w, greg_c), "bf16"), batch, d, h, w, greg_c, 0, greg_c)
            scratch = be.focal_scratch(batch)
            # ghm is zeroed at allocation and written by this kernel only: its padding channels need no store
            fl = be.focal_loss(hm, self.tgt_hm, self.tgt_ind, self.tgt_mask, self.tgt_cat, self.ncls, 1.0, scratch, self.loss_hm,
                               self.ghm, False)
            self.loss_launches.append(fl)
            # the regression gradient is non-zero at <= m voxels per frame: instead of zero-filling its 84 MB every step the
            # loss kernel clears the voxels it wrote last time (state: reg_prev; the buffer starts zeroed, nothing else writes it)
            self.reg_prev = be.alloc((batch, self.m), "i64")
            self.reg_prev.fill_(-1)
            rl = be.reg_loss(reg, self.tgt_pose, self.tgt_ind, self.tgt_mask, self.code_w, self.nreg, self.loss_weight,
                             self.loss_reg, self.greg, self.reg_prev)
            self.loss_launches.append(rl)
            g.seed_grad(hm, self.ghm)
            g.seed_grad(reg, self.greg)
            if feature_channels is not None:
                g.grad_leaves = [self.feats]
            g.build_backward()
            if feature_channels is not None:   # the head alone: hand the feature's gradient back in the caller's layout
                gf = self.feats.grad
                if gf is not None:
                    self.feat_grad = be.alloc((batch, feature_channels, d, h, w), "f32")
                    g.emit_bwd(be.unpack_ncdhw(gf, self.feat_grad, feature_channels), g.lane_of(self.feats), [gf], [self.feat_grad],
                               "unpack:feats.grad")
            self.bwd = list(g.bwd)
            if opt.bwd_f10_first:
                # stage 3's row-1 stride-2 data gradient (level-1 lane) is issued ahead of the row-2 chain it does not depend on:
                # the main lane's fan-in `combine:s3.b0.c3` waits for both, and the level-1 lane used to start this one only after
                # the level-2 lane's chain had arrived (-1 % on the step; options.bwd_f10_first = 0: creation order)
                from .lanes import hoist_tagged
                self.bwd = hoist_tagged(self.bwd, r":s3\.f10\.0$", r":s3\.(row2|f2)")
            if opt.bwd_sink_wg and self.lane_map is LANE_MAP_4:
                # the lower levels' weight gradients behind the other launches of their stage (they share a stream with the level-3
                # chain under the four-stream map: lanes.sink_lane_in_segments); options.bwd_sink_wg = 0: creation order
                from .lanes import sink_lane_in_segments, L_WG_LOW
                self.bwd = sink_lane_in_segments(self.bwd, L_WG_LOW, r"^combine:s\d\.b0\.c3$")
            if self._merge_head:
                from .lanes import merge_launches
                self.bwd, done = merge_launches(self.bwd, be, [("dgrad:head.hm.2", "dgrad:head.reg.2"), ("dgrad:head.hm.0", "dgrad:head.reg.0"),
                                                               ("wgrad:head.hm.2", "wgrad:head.reg.2"), ("wgrad:head.hm.0", "wgrad:head.reg.0")])
                self.merged_head += list(done)
            self.bwd_plan = LanePlan(be, self.bwd, self.lane_map)
        self.live_params = set(g.used_params)
        self.width_hints = list(g.widths_applied)   # [(tag, workgroups)] the rules reached

    def _account_merged(self, done, *families):
        """The merged-in launches' algorithmic FLOPs / bytes join the family their shared launch is timed under (bench.py roofline)."""
        g = self.graph
        for ta, tb in done:
            self.merged.append((ta, tb))
            fl, nb = g.cost.get(tb, (0, 0))
            for fam in families:
                if fam:
                    g.flops[fam] += fl
                    g.alg_bytes[fam] += nb

    # ------------------------------------------------------------------ data in (plumbing copies)
    def load_input(self, rdr_tensor):
        self.x_in.copy_(rdr_tensor.reshape(self.x_in.shape), non_blocking=True)

    def load_features(self, feats):
        if self.feat_in is None:
            raise ValueError("this plan starts at the radar tensor (built without feature_channels)")
        self.feat_in.copy_(feats.reshape(self.feat_in.shape), non_blocking=True)

    def load_lidar(self, grid):
        """grid: dense LiDAR voxel grid [B, C_l, Z, Y, X] (fp32; rt_pose_amd.lidar.DynamicVoxelEncoder.to_dense per frame)."""
        if self.lidar_in is None:
            raise ValueError("this plan was built without a LiDAR input (lidar_channels=0)")
        self.lidar_in.copy_(grid.reshape(self.lidar_in.shape), non_blocking=True)

    def load_targets(self, ex):
        """ex: the reference's example['rdr'] dict with per-task lists (cruw_pose.py:225-275)."""
        self.tgt_hm.copy_(ex["hm"][0], non_blocking=True)
        self.tgt_ind.copy_(ex["ind"][0], non_blocking=True)
        self.tgt_mask.copy_(ex["mask"][0], non_blocking=True)
        self.tgt_cat.copy_(ex["cat"][0], non_blocking=True)
        self.tgt_pose.copy_(ex["anno_pose"][0].reshape(self.tgt_pose.shape), non_blocking=True)

    # ------------------------------------------------------------------ launch lists
    def run_forward(self, stream=None):
        self.fwd_plan.run(stream if stream is not None else self.be.stream(), self.use_lanes)

    def _run_bwd(self, s):
        """Replay the backward list, one HIP stream per lane (lanes.LanePlan) unless use_lanes is off."""
        self.bwd_plan.run(s, self.use_lanes)

    def run_loss_backward(self, stream=None):
        s = stream if stream is not None else self.be.stream()
        for f in self.loss_launches:
            f(s)
        self._run_bwd(s)

    def run_losses_only(self, stream=None):
        """Loss values (and the loss kernels' logit gradients) without the backward sweep."""
        s = stream if stream is not None else self.be.stream()
        for f in self.loss_launches:
            f(s)

    def run_backward_only(self, stream=None):
        self._run_bwd(stream if stream is not None else self.be.stream())

    def losses(self):
        """The reference's loss dict (center_head.py:260): device scalars, no host sync here."""
        hm_loss = self.loss_hm[0]
        loc = self.loss_reg[self.nreg]
        return OrderedDict(loss=hm_loss + self.loss_weight * loc, hm_loss=hm_loss, loc_loss=loc,
                           loc_loss_elem=self.loss_reg[:self.nreg], num_positive=self.tgt_mask.float().sum())

    # ------------------------------------------------------------------ inference
    def set_test_cfg(self, test_cfg):
        osf, vs, pr = test_cfg["out_size_factor"], test_cfg["voxel_size"], test_cfg["pc_range"]
        scale = (osf[2] * vs[0], osf[1] * vs[1], osf[0] * vs[2])
        self.score_threshold = float(test_cfg.get("score_threshold", 0.0))
        scratch = self.be.decode_scratch(self.n, self.ncls)
        self.dec = self.be.decode(self.outs["hm"], self.outs["reg"], self.ncls, self.nreg, scale, tuple(pr[:3]),
                                  scratch, self.dec_out)

    def run_decode(self, stream=None):
        self.dec(stream if stream is not None else self.be.stream())

    def keypoints(self, metas=None):
        """Format like CenterHead.post_processing (center_head.py:332-360): list of dicts per frame."""
        out = self.dec_out.detach().cpu()
        rets = []
        for b in range(self.n):
            kps = []
            if self.nreg == 3:
                for c in range(self.ncls):
                    score = float(out[b, c, 1])
                    if score > self.score_threshold:
                        kps.append((c, *[float(v) for v in out[b, c, 2:5]], score))
            else:
                score = float(out[b, 0, 1])
                pose = [float(v) for v in out[b, 0, 2:2 + self.nreg]]
                if score > self.score_threshold:
                    kps.append((0, *pose[:3], score))
                for i in range(1, self.nreg // 3):
                    kps.append((i, *pose[3 * i:3 * i + 3], score))
            rets.append({"keypoints": kps, "metadata": None if metas is None else metas[b]})
        return rets

    # ------------------------------------------------------------------ exported tensors (logical NCDHW views)
    def output(self, name):
        a = self.outs[name]
        return a.buf[..., :self.heads[name]].permute(0, 4, 1, 2, 3)

    def features(self):
        a = self.feats
        return a.buf[..., :a.c_real].permute(0, 4, 1, 2, 3)


class FlatAdam:
    """Clip + decoupled weight decay + Adam over FlatParams in (at most a few) fused launches."""

    def __init__(self, backend, flat: FlatParams, live_names, wd=0.01, beta2=0.99, eps=1e-8, max_norm=35.0):
        self.be, self.flat = backend, flat
        self.wd, self.beta2, self.eps, self.max_norm = wd, beta2, eps, max_norm
        self.hyper = backend.alloc((10,), "f32")
        # ring of pinned host slots: the async upload of step k must not see step k+1's values when the host runs ahead
        self.hyper_host = torch.zeros(256, 10, dtype=torch.float32)
        if self.hyper.is_cuda:
            self.hyper_host = self.hyper_host.pin_memory()
        self.partial = backend.alloc((max(256, backend.sqnorm_blocks()),), "f32")
        self.norm = backend.alloc((1,), "f32")
        self.t = 0
        self.launches = [backend.sqnorm(flat.g, flat.numel, self.hyper, self.partial)]
        for (a, b, live) in flat.runs(live_names):
            self.launches.append(backend.adam_step(flat.p[a:b], flat.g[a:b] if live else None,
                                                   flat.m[a:b] if live else None, flat.v[a:b] if live else None,
                                                   b - a, self.hyper, self.partial, 0 if live else 1,
                                                   self.norm if live else None))

    def set_hyper(self, lr, beta1, grad_scale=1.0):
        """Host-side scalars for the NEXT step (uploaded asynchronously; graph-capture safe)."""
        self.t += 1
        h = self.hyper_host[self.t % 256]
        h[0], h[1], h[2], h[3], h[4], h[5] = lr, beta1, self.beta2, self.eps, self.wd, self.max_norm
        h[6], h[7], h[8] = 1.0 - beta1 ** self.t, 1.0 - self.beta2 ** self.t, grad_scale
        self.hyper.copy_(h, non_blocking=True)

    def run(self, stream=None):
        s = stream if stream is not None else self.be.stream()
        for f in self.launches:
            f(s)
