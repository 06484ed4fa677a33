"""Static execution plan for the HRRadarPose hot path.

The network is a fixed DAG, so instead of a tracing compiler the host builds, once per input shape, two flat
lists of kernel launches (forward, backward) over statically allocated HBM buffers; a step replays the lists
(and can be captured into a HIP graph).  Nodes are the fused macro-ops the HIP kernels implement:

  stem / pack            network input -> bf16 channels-last
  conv                   [GroupNorm folded] Conv3d k in {1,3}, stride in {1,2} [+bias] [+residual] [ReLU]
  fuse                   sum of same-resolution terms and trilinear-upsampled low-resolution terms [ReLU]

Backward is emitted mechanically in reverse creation order.  Every activation collects "contributions" from
its consumers (plain addends, or GroupNorm-backward terms A*dxhat + B*x + C); one grad_combine launch per
activation sums them and applies the activation's own ReLU mask, so no gradient tensor is written twice.

Every list entry is a lanes.Launch: the closure plus its LANE (HIP stream: one per resolution group, plus lanes for
the weight-gradient chains, which only feed the optimiser) and the buffers it reads / writes, from which
lanes.LanePlan derives the cross-stream event waits.  List order is always a valid single-stream order.

The `backend` supplies the kernels (rt_pose_amd.backend.HipBackend in the product; tests inject a torch-CPU
emulation of each kernel to check this file's plan logic without a GPU).  Every backend method returns a
closure f(stream) so argument marshalling happens once, at build time.
"""
import os
from dataclasses import dataclass, field, replace
from typing import List, Optional

from .lanes import Launch, L_FULL, L_MID, L_LOW, L_LOW3, L_WG, L_WG_LOW


GROUPS = 8      # nn.GroupNorm(num_groups=8, ...) everywhere on the path (hr_util/common.py:57, hr3d.py:147)
GN_EPS = 1e-5


def pad_to(v, m):
    return (v + m - 1) // m * m


@dataclass
class Geom:
    n: int
    di: int
    hi: int
    wi: int
    do: int
    ho: int
    wo: int
    ci: int          # padded to 32
    co: int          # padded to 16
    ks: int
    stride: int
    pad: int
    w_ci_total: int = 0
    w_ci_off: int = 0
    wgs: int = 0     # launch width of the LDS-tiled kernels (include/rtp.h RtpConvGeom::wgs): 0 = one workgroup per CU


class View:
    """A channels-last tensor view handed to kernels: buf [n,d,h,w,cs], channels [co, co+c)."""

    def __init__(self, buf, n, d, h, w, cs, co, c):
        self.buf, self.n, self.d, self.h, self.w, self.cs, self.co, self.c = buf, n, d, h, w, cs, co, c

    @property
    def vox(self):
        return self.d * self.h * self.w

    @property
    def dims(self):
        return (self.d, self.h, self.w)


class LazyCoeff:
    """GroupNorm-backward coefficients of one conv's input that no launch has computed yet: the fan-in pass that consumes them
    (rtp_grad_combine_cls_lazy) does it in its own prologue, from the statistics partials of the data gradient.  `materialise`
    emits the rtp_gn_bwd_coeffs launch instead, for consumers that need the finished table (fused data-gradient epilogues, the
    plain combine)."""

    def __init__(self, g, name, lane, pq, nsplit, mr, gamma, n, c, groups, vox, tensor):
        self.g, self.name, self.lane = g, name, lane
        self.pq, self.nsplit, self.mr, self.gamma, self.n, self.c, self.groups, self.vox = pq, nsplit, mr, gamma, n, c, groups, vox
        self.tensor = tensor          # [n*c*5] fp32: coefficients [n][c][3] + dgamma/dbeta partials [n][c][2]
        self.done = False

    def materialise(self):
        if not self.done:
            g = self.g
            g.emit_bwd(g.be.gn_bwd_coeffs(self.pq, self.nsplit, self.mr, self.gamma, self.n, self.c, self.groups, self.vox,
                                          self.tensor, None, None, 0),
                       self.lane, [self.pq, self.mr], [self.tensor], "gncoef:" + self.name)
            self.done = True
        return self.tensor


class Act(View):
    """An activation node: its buffer, whether it went through a ReLU, and its gradient bookkeeping."""

    def __init__(self, g, name, c_real, dims, c=None, dtype="bf16", relu=False, needs_grad=True, buf=None):
        c = c or c_real
        d, h, w = dims
        if buf is None:
            buf = g.be.alloc((g.n, d, h, w, c), dtype)
        super().__init__(buf, g.n, d, h, w, c, 0, c)
        self.g, self.name, self.c_real, self.dtype, self.relu = g, name, c_real, dtype, relu
        self.needs_grad = needs_grad
        self.stats = None          # chan_stats partials [n, S, c, 2]
        self.stats_split = 0
        self.contribs = []         # list of (View, coeff tensor | None)
        self.grad: Optional[View] = None
        self.grad_cls = None       # (nsplit, partial class sums of .grad) when grad_combine emitted them
        self.final = False         # .grad was written complete by a fused data gradient (no fan-in pass needed)
        self.grad_tot = None       # (nsplit, per-channel total partials of .grad) when that fused data gradient emitted them
        self.producer = None


def stats_split(vox):
    return max(1, min(64, (vox + 1023) // 1024))


def cls_split(d, h):
    return max(1, min(64, (d * h + 3) // 4))


def wgrad_split(vox):
    """Slabs per sample of the generic weight-gradient kernel: two 128-voxel chunks per block keep the low-resolution
    layers spread over the chip without multiplying the slab bytes the tail has to fold."""
    return max(1, min(64, (vox + 255) // 256))


class Graph:
    def __init__(self, backend, n, params, train=True, pgrads=None, width_rules=(), options=None):
        """params: dict name -> fp32 tensor (reference state_dict names/shapes).
        pgrads: optional dict name -> fp32 tensor the backward plan writes parameter gradients into
        (views of a flat buffer); allocated per parameter when absent.
        width_rules: [(launch-tag prefix, workgroups)]: the LDS-tiled launches whose tag starts with a prefix run on that many
        workgroups (first match wins) -- an explicit field of the launch's geometry (Geom.wgs -> RtpConvGeom::wgs)."""
        self.be, self.n, self.params, self.train = backend, n, params, train
        self.width_rules = [(str(k), int(v)) for k, v in width_rules]
        self.widths_applied = []   # (tag, workgroups) of the launches a rule reached
        self._pgrads_ext = pgrads
        self.fwd: List = []
        self.bwd: List = []
        self.ops: List = []
        self.acts: List[Act] = []
        self.pgrad = {}            # name -> fp32 grad tensor written by the backward plan
        self.used_params = []      # creation order
        self.bytes = 0
        self.tail_a, self.tail_b = [], []   # deferred optimiser-only items (emit_tail)
        from .options import PlanOptions
        self.opt = options if options is not None else PlanOptions.from_env()
        # (folds are flushed once, at the end of the sweep: earlier, smaller flushes on the weight-gradient lane measured the same,
        # 1 158-1 170 frames/s for 12 / 20 / all in round 1)
        # Launches of these lanes (default: the full-resolution weight-gradient lane) are queued onto the MAIN lane in front of its
        # fan-ins; all of them in front of the SECOND fan-in, where the main lane waits longest (round 5: 5.22 -> 5.17 ms)
        self._defer_wg = self.opt.int_list("defer_wg")
        self._defer_keep = 0
        # at most this many launches move (the two 32-channel head towers of hr3d: 8; the wide heads of the one-heat-map configs
        # queue 20+ launches there, more than the waits absorb: measured 11.20 ms/step without the move, 11.45 with all of it)
        self._defer_max, self._ndeferred = 8, 0
        # (the stem's weight gradient is the sweep's last launch: nothing left to wait for, it stays beside the tail on its lane)
        self._defer_skip = {"stem_bwd"}
        self._deferred = []
        self.head, self._head_emitted = [], False   # activation-independent weight packing (forward_list)
        self._lazy_by_coeff = {}   # coefficient tensor -> its LazyCoeff (an early tail flush has to materialise pending ones)
        # Data-parallel training with TWO gradient buckets (trainer.DataParallelTrainer(ar_buckets=2)): the deferred tail is
        # flushed once early -- when the reversed sweep leaves stage 3 -- so that the gradients of transition2 .. pose_head (a
        # contiguous suffix of the flat buffer, state_dict order) are final while stage 2 / layer 1 are still being swept, and
        # their all-reduce runs beside the rest of the backward pass.  early_tail_index: position of that flush in self.bwd.
        self.early_flush = False
        self.early_tail_index = None
        self.early_tail_launch = None
        self.group = None          # (block, row) tag the ops created from now on carry (net.build_backbone: fuse rows)
        self.full_vox = None       # voxels of the first (full-resolution) activation: lanes are assigned by resolution
        # algorithmic FLOPs (2*MACs of real channels) per kernel family, per replay of the lists
        self.flops = {"conv_fwd": 0, "conv_dgrad": 0, "wgrad": 0, "conv_tiled": 0, "conv_generic": 0, "conv64": 0,
                      "wgrad_tiled": 0, "wgrad_generic": 0, "conv_tiled_full": 0, "conv_tiled_full_bwd": 0}
        # algorithmic HBM bytes (fused minimum, SURVEY 8d: every operand tensor of a launch read or written once) per family
        self.cost = {}   # launch tag -> (algorithmic FLOPs, bytes) of the tiled conv / data-gradient launches (engine: shared launches)
        self.alg_bytes = {"conv_tiled": 0, "conv_generic": 0, "conv64": 0, "wgrad_tiled": 0, "wgrad_generic": 0, "conv_tiled_full": 0,
                          "conv_tiled_full_bwd": 0}

    # ------------------------------------------------------------------ helpers
    def with_width(self, geom, tag):
        """`geom` for the launch tagged `tag`: a copy carrying the launch width of the first matching rule, or geom itself."""
        cache = self.__dict__.setdefault("_width_cache", {})
        if tag in cache:   # (one geometry object per launch tag: the slot-count queries and the launch itself must see the same width)
            return cache[tag]
        wgs = next((w for pre, w in self.width_rules if tag.startswith(pre)), 0)
        out = geom
        if wgs:
            self.widths_applied.append((tag, wgs))
            out = replace(geom, wgs=wgs)
        cache[tag] = out
        return out

    def _add_op(self, op):
        op.group = self.group
        self.ops.append(op)

    def sweep_order(self):
        """Order in which build_backward processes the ops: reverse creation order, except that inside one fuse block
        (HighResolutionModule.forward's rows, hr3d.py:205-229) the ROW GROUPS are taken lowest resolution first (row 2, row 1, row 0)
        instead of row 0 first.  Every lane is a FIFO stream that executes its launches in list order: with row 0 first, the
        level-1 lane queues row 0's up-sampling adjoint -- which waits for the main lane's newest gradient -- in FRONT of the
        stride-2 data gradients of rows 1 and 2, whose inputs are long finished and whose outputs the main lane needs next
        (head-of-line blocking at every stage boundary).  Rows of a block only meet at the branch outputs they all contribute
        to, so any row order is a valid reverse-topological order."""
        rev = list(reversed(self.ops))
        # (options.fused_s2, the A/B route where a row's stride-2 data gradient absorbs the fan-in of the branch-0 output, needs
        # that conv to be the last contributor: plain reverse creation order)
        if self.opt.fused_s2:
            return rev
        out, i = [], 0
        while i < len(rev):
            gi = getattr(rev[i], "group", None)
            if gi is None:
                out.append(rev[i])
                i += 1
                continue
            j, chunks = i, []
            while j < len(rev) and getattr(rev[j], "group", None) is not None and rev[j].group[0] == gi[0]:
                k = j
                while k < len(rev) and getattr(rev[k], "group", None) == rev[j].group:
                    k += 1
                chunks.append(rev[j:k])
                j = k
            chunks = list(reversed(chunks))                         # lowest resolution first: row 2, row 1, row 0
            for ch in chunks:
                out.extend(ch)
            i = j
        return out

    def act(self, *a, **k):
        t = Act(self, *a, **k)
        self.acts.append(t)
        if self.full_vox is None:
            self.full_vox = t.vox
        return t

    def lane_of(self, v):
        """Resolution group of a tensor: one lane per HRNet level (the low-resolution branches are chains of small
        latency-bound launches, so each level gets its own stream)."""
        r = self.full_vox // max(1, v.vox)
        return L_FULL if r < 4 else L_MID if r < 32 else L_LOW if r < 256 else L_LOW3

    def wg_lane_of(self, v):
        return L_WG if self.lane_of(v) == L_FULL else L_WG_LOW

    def emit_fwd(self, fn, lane, reads=(), writes=(), tag=""):
        self.fwd.append(Launch(fn, lane, reads, writes, tag))

    def emit_bwd(self, fn, lane, reads=(), writes=(), tag=""):
        # The launches of the full-resolution weight-gradient lane (head towers: weight gradients and their class sums; they feed
        # only the deferred tail) do not run beside the main lane's kernels on a stream of their own -- two persistent kernels
        # only time-share the CUs and stretch the critical chain -- but are queued and issued ON the main lane where it is about
        # to wait for the side lanes anyway: in front of the second fan-in of a full-resolution block output (RTP_DEFER_WG="3"
        # default, "" = off; only lanes whose results nothing but the tail reads may be listed: measured 6.30 -> 6.22 ms/step)
        if self._defer_wg and lane in self._defer_wg and tag not in self._defer_skip and self._ndeferred < self._defer_max:
            self._ndeferred += 1
            self._deferred.append(Launch(fn, L_FULL, reads, writes, tag))
            return
        L = Launch(fn, lane, reads, writes, tag)
        # List order must stay a valid serial order: a launch that reads (or overwrites) what a still-queued launch writes, or
        # overwrites what one reads, goes behind it -- the queue is issued first (the fused backward puts producers
        # of the data gradient's prologue inputs on the deferred lane in the A/B modes)
        if self._deferred:
            dw = set(k for D in self._deferred for k in D.writes)
            dr = set(k for D in self._deferred for k in D.reads)
            if dw.intersection(L.reads) or dw.intersection(L.writes) or dr.intersection(L.writes):
                self.flush_deferred()
        self.bwd.append(L)

    def flush_deferred(self, keep=0):
        """Issue the queued weight-gradient launches on the main lane (all but the newest `keep`)."""
        n = len(self._deferred) - keep
        if n > 0:
            self.bwd.extend(self._deferred[:n])
            del self._deferred[:n]

    def param(self, name):
        p = self.params[name]
        if name not in self.pgrad:
            self.used_params.append(name)
            if not self.train:
                self.pgrad[name] = None
            elif self._pgrads_ext is not None:
                self.pgrad[name] = self._pgrads_ext[name]
            else:
                self.pgrad[name] = self.be.alloc(tuple(p.shape), "f32")
        return p

    def slice_ws(self, lane, floats):
        """fp32 scratch of the channel-sliced convs (backend.conv(ws=...)), one buffer per lane: launches of one lane are
        stream-ordered anyway, and a shared buffer would tie the lanes together through false dependencies."""
        d = self.__dict__.setdefault("_slice_ws", {})
        if (lane, floats) not in d:
            d[(lane, floats)] = self.be.alloc((floats,), "f32")
        return d[(lane, floats)]

    def ensure_stats(self, x: Act):
        if x.stats is None:
            x.stats_split = stats_split(x.vox)
            x.stats = self.be.alloc((self.n, x.stats_split, x.c, 2), "f32")
            self.emit_fwd(self.be.chan_stats(x, None, x.stats_split, x.stats), self.lane_of(x), [x], [x.stats], "stats:" + x.name)
        return x.stats

    # ------------------------------------------------------------------ forward node constructors
    def input_f32(self, name, c, dims):
        d, h, w = dims
        buf = self.be.alloc((self.n, c, d, h, w), "f32")
        return buf

    def stem(self, name, x_f32, dims, wname, bname):
        """layer1.conv1 with Cin == 1 (hr_util/common.py:111-113)."""
        w, b = self.param(wname), self.param(bname)
        y = self.act(name, w.shape[0], dims)
        op = StemOp(self, x_f32, y, wname, bname)
        y.producer = op
        self._add_op(op)
        S = self.be.stem_stats_nsplit(self.n, y.c, y.vox) if hasattr(self.be, "stem_fwd_stats") else 0
        if S > 0:   # the statistics the block's first GroupNorm needs, as the stem's epilogue: no rtp_chan_stats pass over its output
            y.stats_split = S
            y.stats = self.be.alloc((self.n, S, y.c, 2), "f32")
            self.emit_fwd(self.be.stem_fwd_stats(x_f32, w, b, y, y.stats, S), self.lane_of(y), [x_f32], [y, y.stats], "stem:" + name)
        else:
            self.emit_fwd(self.be.stem_fwd(x_f32, w, b, y), self.lane_of(y), [x_f32], [y], "stem:" + name)
        return y

    def pack(self, name, x_f32, c, dims):
        y = self.act(name, c, dims, c=pad_to(c, 32), needs_grad=False)
        self.emit_fwd(self.be.pack_ncdhw(x_f32, y, c), self.lane_of(y), [x_f32], [y], "pack:" + name)
        return y

    def conv(self, name, x: Act, wname, bname=None, gn=None, ks=3, stride=1, relu=False, residual=None,
             out_fp32=False, w_ci_total=0, w_ci_off=0, ci_real=None, want_stats=True):
        """want_stats: a GroupNorm will consume the output, so the conv's epilogue should emit its statistics."""
        w = self.param(wname)
        co_real = w.shape[0]
        ci_real = ci_real if ci_real is not None else w.shape[1]
        pad = ks // 2
        do, ho, wo = [(s + 2 * pad - ks) // stride + 1 for s in x.dims]
        co_pad = pad_to(co_real, 16)
        # A wide 3x3x3 conv without GroupNorm (the head towers of the 128/256-channel configs, center_head.py:86-93) runs as
        # 32-channel input slices on the LDS-tiled kernel when its geometry allows (SplitConvOp)
        # ... and a 32 -> 45 tower output (the 45-offset regression head of the one-heat-map configs) as output-channel
        # slices 32 + 16 (CoSplitConvOp)
        if (gn is None and ks == 3 and stride == 1 and residual is None and out_fp32 and ci_real == 32 and x.c == 32
                and x.co == 0 and co_pad == 48 and bname and not w_ci_total and hasattr(self.be, "conv_tiled_ok")):
            gs = Geom(self.n, x.d, x.h, x.w, do, ho, wo, 32, 32, ks, stride, pad)
            if self.be.conv_tiled_ok(x, gs, False):
                y = self.act(name, co_real, (do, ho, wo), c=co_pad, dtype="f32", relu=relu)
                op = CoSplitConvOp(self, name, x, y, wname, bname, relu, co_real)
                y.producer = op
                self._add_op(op)
                op.emit_forward()
                return y
        if (gn is None and ks == 3 and stride == 1 and residual is None and not out_fp32 and ci_real > 32 and ci_real % 32 == 0
                and x.c == ci_real and x.co == 0 and not w_ci_total and co_pad in (16, 32) and co_real % 16 == 0
                and hasattr(self.be, "conv_tiled_ok")):
            gs = Geom(self.n, x.d, x.h, x.w, do, ho, wo, 32, co_pad, ks, stride, pad, ci_real, 0)
            if self.be.conv_tiled_ok(View(x.buf, x.n, x.d, x.h, x.w, x.cs, 0, 32), gs, False):
                y = self.act(name, co_real, (do, ho, wo), c=co_real, relu=relu)
                op = SplitConvOp(self, name, x, y, gs, wname, bname, relu, ci_real, co_real)
                y.producer = op
                self._add_op(op)
                op.emit_forward()
                return y
        geom = Geom(self.n, x.d, x.h, x.w, do, ho, wo, pad_to(ci_real, 32), co_pad, ks, stride, pad,
                    w_ci_total, w_ci_off)
        assert x.c >= geom.ci, (name, x.c, geom.ci)
        y = self.act(name, co_real, (do, ho, wo), c=co_pad if out_fp32 else pad_to(co_real, 32) if co_real % 16 else co_real,
                     dtype="f32" if out_fp32 else "bf16", relu=relu)
        op = ConvOp(self, name, x, y, geom, wname, bname, gn, relu, residual, out_fp32, ci_real, co_real)
        op.want_stats = want_stats
        y.producer = op
        self._add_op(op)
        op.emit_forward()
        return y

    def conv_pair(self, names, x: Act, wnames, bnames, relu=True):
        """Two Conv3d(C, 32, 3x3x3, bias) [+ ReLU] over the SAME wide feature x -- SepHead's towers for the 128- / 256-channel features
        of the one-heat-map configs (center_head.py:86-93) -- sharing their launches: forward and data gradient run 64 wide on
        csrc/conv64_tiled.hip (one launch per 64-channel slice of x instead of two per 32-channel slice), and x receives ONE gradient
        tensor, the sum over both towers.  -> [Act, Act], or None when the pair cannot share (the caller then builds them one by one)."""
        be = self.be
        if not hasattr(be, "conv64_blocks") or not self.opt.pair_heads:
            return None
        ws = [self.param(n) for n in wnames]
        ci_real = ws[0].shape[1]
        if (len(names) != 2 or any(tuple(w.shape) != (32, ci_real, 3, 3, 3) for w in ws) or ci_real % 64 or x.c != ci_real or x.co != 0
                or x.cs != ci_real or not all(bnames)):
            return None
        gs = Geom(self.n, x.d, x.h, x.w, x.d, x.h, x.w, 32, 32, 3, 1, 1, ci_real, 0)
        if not be.conv_tiled_ok(View(x.buf, x.n, x.d, x.h, x.w, x.cs, 0, 32), gs, False):
            return None
        ops = []
        for name, wn, bn in zip(names, wnames, bnames):
            y = self.act(name, 32, x.dims, c=32, relu=relu)
            op = SplitConvOp(self, name, x, y, gs, wn, bn, relu, ci_real, 32)
            y.producer = op
            self._add_op(op)
            op.emit_forward(launches=False)
            ops.append(op)
        a, b = ops
        a.partner, b.partner, a._pair_first, b._pair_first = b, a, True, False
        a.emit_pair_forward()
        return [a.y, b.y]

    def conv_cat(self, name, xs: List[Act], reals, wname, bname=None, relu=False):
        """Conv3d(sum(reals), Cout in {16, 32}, 3x3x3, bias) [+ ReLU] over the channel concatenation of `xs` without building
        it: every source (<= 32 real channels, padded to 32) is one input-channel slice of a SplitConvOp -- the two-stream
        fusion head (radar feature ++ dense LiDAR grid)."""
        w = self.param(wname)
        co_real, ci_total = w.shape[0], w.shape[1]
        assert ci_total == sum(reals) and all(x.c == 32 and x.cs == 32 and x.co == 0 and r <= 32 for x, r in zip(xs, reals)), name
        x0 = xs[0]
        gs = Geom(self.n, x0.d, x0.h, x0.w, x0.d, x0.h, x0.w, 32, pad_to(co_real, 16), 3, 1, 1, ci_total, 0)
        if not (hasattr(self.be, "conv_tiled_ok") and self.be.conv_tiled_ok(View(x0.buf, x0.n, x0.d, x0.h, x0.w, 32, 0, 32), gs, False)
                and co_real % 16 == 0):
            raise NotImplementedError("conv_cat needs the LDS-tiled conv geometry (D % 2, H % 4, W % 16, Cout 16 | 32)")
        y = self.act(name, co_real, x0.dims, c=co_real, relu=relu)
        offs = [sum(reals[:k]) for k in range(len(xs))]
        op = SplitConvOp(self, name, x0, y, gs, wname, bname, relu, ci_total, co_real, sources=list(zip(xs, reals, offs)))
        y.producer = op
        self._add_op(op)
        op.emit_forward()
        return y

    def fuse(self, name, terms: List[Act], relu=True, want_stats=True):
        """HighResolutionModule fuse row (hr3d.py:213-228): terms at other resolutions are upsampled."""
        hi = max(terms, key=lambda t: t.vox)
        y = self.act(name, hi.c_real, hi.dims, c=hi.c, relu=relu)
        op = FuseOp(self, terms, y)
        y.producer = op
        self._add_op(op)
        # a fuse row feeds GroupNorm convs of the next stage: it emits the statistics of what it stores (no chan_stats pass)
        S = self.be.fuse_stats_nsplit(y) if (want_stats and hasattr(self.be, "fuse_stats_nsplit") and self.opt.fuse_stats) else 0
        st = None
        if S > 0:
            y.stats_split = S
            y.stats = self.be.alloc((self.n, S, y.c, 2), "f32")
            st = (S, y.stats)
        self.emit_fwd(self.be.fuse_sum(terms, None, y, relu, st) if st else self.be.fuse_sum(terms, None, y, relu),
                      self.lane_of(y), terms, [y, y.stats if st else None], "fuse:" + name)
        return y

    def concat(self, name, terms: List[Act]):
        """cat(terms, channel axis) with the lower-resolution terms upsampled (trilinear, align_corners=True) to the first term's
        size -- HRNet3D.forward's plain-concat fuse (hrnet3d.py:37-40; any final_fuse other than 'top' / 'conat_conv' leaves it
        un-convolved, appendix quirk 5).  Every term is one single-term launch of the fuse-row kernel writing its channel slice of
        the wide tensor; the backward hands each term the matching slice of the gradient (through the upsample adjoint)."""
        hi = terms[0]
        assert all(t.c == t.c_real and t.co == 0 and t.c % 8 == 0 for t in terms), name
        total = sum(t.c_real for t in terms)
        y = self.act(name, total, hi.dims, c=total, relu=False)
        op = ConcatOp(self, terms, y)
        y.producer = op
        self._add_op(op)
        off = 0
        for k, t in enumerate(terms):
            yk = View(y.buf, y.n, y.d, y.h, y.w, y.cs, off, t.c)
            self.emit_fwd(self.be.fuse_sum([t], None, yk, False), self.lane_of(y), [t], [y], "concat:%s.%d" % (name, k))
            off += t.c
        return y

    def dcn_adapt(self, name, x: Act, prefix):
        """FeatureAdaption of the reference's DCN head (center_head.py:24-62) with Z folded into the batch:
        relu(DeformConv3x3_dg4(x, Conv1x1(x))) per (frame, z) slice.  The 1x1 offset conv is an ordinary conv node of the plan
        (MFMA kernel, fp32 output, its own weight / bias / data gradients); the deformable conv is the native operator.
        Parameters: prefix + .conv_offset.{weight,bias}, .conv_adaption.weight (2-D shapes, reference names)."""
        w_ad = self.param(prefix + ".conv_adaption.weight")
        assert x.c == x.c_real == w_ad.shape[1] and x.cs == x.c and x.co == 0, (name, x.c, x.c_real)
        off = self.conv(name + ".off", x, prefix + ".conv_offset.weight", bname=prefix + ".conv_offset.bias", ks=1, out_fp32=True,
                        want_stats=False)
        y = self.act(name, x.c_real, x.dims, relu=True)
        op = DcnAdaptOp(self, name, x, off, y, prefix)
        y.producer = op
        self._add_op(op)
        op.fwd_fn, op.make_bwd = self.be.dcn_adapt(x, off, off.c_real, w_ad, y)
        # The two feature adaptions (heat-map / regression) are independent and their operator's kernels are ordinary grids, not
        # chip-filling persistent ones: the regression one runs on the level-1 lane beside the heat-map one's
        op.lane = L_MID if name.endswith(".reg") else self.lane_of(y)
        self.emit_fwd(op.fwd_fn, op.lane, [x, off, w_ad], [y], "dcn:" + name)
        if self.train and hasattr(op.make_bwd, "prep"):
            # the forward runs on the plan's layout (rtp_dcn_cl_forward); the fp32 planes the BACKWARD operator reads are unpacked by a
            # launch of their own on a lane that idles while the head runs, instead of in front of the backward operator
            # (the planes are plan-tracked buffers: written here, read by the backward operator's launch -- the lane planner and
            # the reorder passes see the dependency instead of relying on the forward plan's final join)
            self.emit_fwd(op.make_bwd.prep(), L_WG_LOW, [x, off], list(getattr(op.make_bwd, "planes", ())), "dcnprep:" + name)
        return y

    def forward_list(self):
        """The forward launches, opened by the one launch that packs every activation-independent weight image."""
        if self.head and not self._head_emitted:
            outs = [t for it in self.head for t in (it[13:17] if it[0] == "fold_fwd" else it[-1:]) if t is not None]
            self.fwd.insert(0, Launch(self.be.tail(self.head), L_FULL, [it[1] for it in self.head], outs, "pack_weights"))
            self._head_emitted = True
        return self.fwd

    # ------------------------------------------------------------------ backward
    def seed_grad(self, y: Act, gview: View):
        """Declare the gradient of a graph output (already scaled; written by a loss kernel or by autograd)."""
        y.contribs.append((gview, None))

    def finalize_grad(self, t: Act, want_cls=False):
        """want_cls: the consumer needs per-boundary-class sums of this gradient (ConvOp with bias / GroupNorm);
        when a grad_combine launch produces the tensor anyway, it emits them in the same pass."""
        if t.final:
            return t.grad
        if not t.contribs:
            return None
        if len(t.contribs) == 1 and t.contribs[0][1] is None and not t.relu:
            t.grad = t.contribs[0][0]
            return t.grad
        c = t.contribs[0][0].c
        gbuf = self.be.alloc((self.n, t.d, t.h, t.w, c), "bf16")
        t.grad = View(gbuf, self.n, t.d, t.h, t.w, c, 0, c)
        need_x = any(cf is not None for _, cf in t.contribs)
        # at most RTP_MAX_TERMS terms per launch; chain if a node ever has more
        terms = list(t.contribs)
        first = True
        while terms:
            chunk, terms = terms[:5 if not first else 6], terms[5 if not first else 6:]
            if not first:
                chunk = [(t.grad, None)] + chunk
            last = not terms
            cls = None
            can_lazy = hasattr(self.be, "grad_combine_lazy_ok") and bool(self.opt.lazy_coef)
            pending = can_lazy and any(isinstance(cf, LazyCoeff) and not cf.done for _, cf in chunk)
            # the class-sum variant of the combine also when nobody wants the sums but a term's coefficients are still to be
            # computed: its prologue does that (per-sample blocks), which is cheaper than a coefficient launch in the chain
            if last and (want_cls or pending) and self.be.grad_combine_cls_ok(c):
                nsplit = cls_split(t.d, t.h)
                cls = (nsplit, self.be.alloc((self.n, nsplit, 64, c), "f32"))
                if want_cls:
                    t.grad_cls = cls
            # coefficients nobody has computed yet: the class-sum combine does it in its prologue, anything else gets the launch
            lazy_ok = cls is not None and can_lazy
            reads, writes = [], []
            for i, (v, cf) in enumerate(chunk):
                if isinstance(cf, LazyCoeff):
                    if cf.done or not lazy_ok:
                        chunk[i] = (v, cf.materialise())
                        reads.append(cf.tensor)
                    else:
                        cf.done = True          # computed (and written out) by this launch
                        reads += [cf.pq, cf.mr]
                        writes.append(cf.tensor)
                else:
                    reads.append(cf)
            if self.lane_of(t) == L_FULL and self._deferred:
                self._ncomb0 = getattr(self, "_ncomb0", 0) + 1
                if self._ncomb0 == 2:
                    self.flush_deferred(self._defer_keep)
                elif self._ncomb0 > 2:
                    self.flush_deferred()
            self.emit_bwd(self.be.grad_combine(chunk, t if need_x else None, t if (t.relu and last) else None, t.grad, cls),
                          self.lane_of(t), [v for v, _ in chunk] + reads + [t],
                          [t.grad, cls[1] if cls else None] + writes, "combine:" + t.name)
            first = False
        return t.grad

    def class_sums_for(self, y: Act, gy: View, lane, name, early=False):
        """Per-boundary-class channel sums [n][64][c] of gy, the gradient of activation y (bias gradient, GroupNorm un-fold of
        the weight gradient, and P of the fused GroupNorm backward).  The pass that produced gy may already have emitted
        them (grad_combine) or the per-channel totals (fused data gradient: only the boundary voxels are scanned then).
        early: the sums are needed during the sweep, not just by the deferred tail."""
        be = self.be
        csum = be.alloc((self.n, 64, gy.c), "f32")
        if y.grad is gy and y.grad_cls is not None:
            split, scratch = y.grad_cls
            if early:
                self.emit_bwd(be.class_sums_reduce(scratch, split, self.n, gy.c, csum), lane, [scratch], [csum], "clsred:" + name)
            else:
                self.tail_a.append(("class_reduce", scratch, split, self.n, gy.c, csum))
        elif y.grad is gy and y.grad_tot is not None and hasattr(be, "class_sums_boundary"):
            tsplit, tot = y.grad_tot
            split = min(32, cls_split(gy.d, gy.h))
            scratch = be.alloc((self.n, split, 64, gy.c), "f32")
            self.emit_bwd(be.class_sums_boundary(gy, split, scratch, tot, tsplit, csum), lane, [gy, tot], [scratch, csum],
                          "clsb:" + name)
        else:
            split = cls_split(gy.d, gy.h)
            scratch = be.alloc((self.n, split, 64, gy.c), "f32")
            if early:
                self.emit_bwd(be.class_sums(gy, split, scratch, csum), lane, [gy], [scratch, csum], "cls:" + name)
            else:
                self.emit_bwd(be.class_sums(gy, split, scratch, None), lane, [gy], [scratch], "cls:" + name)
                self.tail_a.append(("class_reduce", scratch, split, self.n, gy.c, csum))
        return csum

    def build_backward(self):
        assert self.train
        import os
        # The FIRST-created consumer of an activation is the LAST to add its gradient contribution in the reversed sweep:
        # if that consumer is a conv whose data gradient runs on the LDS-tiled kernel, its epilogue absorbs the fan-in
        # (all other contributions + GroupNorm backward + ReLU mask) and writes the finished gradient (ConvOp.emit_backward).
        self.first_consumer = {}
        self.fused_dgrad = hasattr(self.be, "conv_dgrad_fused")
        order = self.sweep_order()
        for op in order:   # ("first consumer" = the consumer the sweep reaches LAST: its contribution completes the tensor's gradient)
            for t in op.inputs():
                self.first_consumer[id(t)] = op
        # (the move of the weight-gradient lane onto the main lane pays for the two 32-channel towers of hr3d; the wide heads of
        # the one-heat-map configs -- slice ops -- queue more work there than the waits absorb: 11.3 vs 11.5 ms/step, so not by default)
        if self.opt.defer_wg == "3" and any(isinstance(op, (SplitConvOp, CoSplitConvOp)) for op in self.ops):   # (the default only)
            self._defer_wg = []
        for op in order:
            if self.early_flush and self.early_tail_index is None and op.y.name.startswith(("l1.", "t1", "s2.")):
                # (coefficients a later fan-in pass would have computed in its prologue are needed by the GroupNorm parameter
                # sums of this flush: their launches are emitted now)
                for it in self.tail_a:
                    lz = self._lazy_by_coeff.get(it[1].data_ptr()) if it[0] == "gn_param" else None
                    if lz is not None:
                        lz.materialise()
                self.emit_tail(L_WG_LOW)
                self.early_tail_index = len(self.bwd) - 1
                self.early_tail_launch = self.bwd[-1]   # the Launch object itself: list positions do not survive re-orderings
            gy = self.finalize_grad(op.y, isinstance(op, (ConvOp, SplitConvOp, CoSplitConvOp)) and bool(op.gn or op.bname))
            if gy is None:
                continue
            op.emit_backward(gy)
        for op in order:
            if getattr(op, "partner", None) is not None and op.x.needs_grad:
                assert (op._pair_gy is None) == (op.partner._pair_gy is None), "paired head towers: one has a gradient, the other not"
        # graph inputs whose gradient the caller wants (a head-only plan's feature): their fan-in goes in front of the tail, whose
        # GroupNorm parameter sums may use coefficients that fan-in's prologue computes
        for t in getattr(self, "grad_leaves", ()):
            self.finalize_grad(t)
        self.emit_tail(L_FULL)

    def emit_tail(self, lane=L_FULL):
        """The deferred items as two launches (stage a: class reductions + GroupNorm parameter sums; stage b: folds)."""
        import os
        self.flush_deferred()
        a, b = self.tail_a, self.tail_b
        self.tail_a, self.tail_b = [], []
        stages = [a, b]
        if self.opt.no_tail:   # A/B: one launch per item, like a per-layer plan
            stages = [[it] for it in a] + [[it] for it in b]
        for items in stages:
            if not items:
                continue
            bufs = [t for it in items for t in it[1:] if hasattr(t, "data_ptr")]
            outs = {"class_reduce": (5,), "gn_param": (4, 5), "wgrad_fold": (11, 12)}
            writes = [it[i] for it in items for i in outs[it[0]] if it[i] is not None]
            wk = set(t.data_ptr() for t in writes)
            self.emit_bwd(self.be.tail(items), lane, [t for t in bufs if t.data_ptr() not in wk], writes, "tail")


class StemOp:
    def __init__(self, g, x_f32, y, wname, bname):
        self.g, self.x, self.y, self.wname, self.bname = g, x_f32, y, wname, bname

    def inputs(self):
        return []

    def emit_backward(self, gy):
        g = self.g
        scratch = g.be.alloc((g.be.stem_bwd_blocks(), self.y.c, 2), "f32")
        g.emit_bwd(g.be.stem_bwd(self.x, gy, scratch, g.pgrad[self.wname], g.pgrad[self.bname], 0), g.wg_lane_of(gy),
                   [self.x, gy], [scratch, g.pgrad[self.wname], g.pgrad[self.bname]], "stem_bwd")


class ConvOp:
    def __init__(self, g, name, x, y, geom, wname, bname, gn, relu, residual, out_fp32, ci_real, co_real):
        self.g, self.name, self.x, self.y, self.geom = g, name, x, y, geom
        self.wname, self.bname, self.gn, self.relu, self.residual, self.out_fp32 = wname, bname, gn, relu, residual, out_fp32
        self.ci_real, self.co_real = ci_real, co_real
        self.mr = None

    def inputs(self):
        return [self.x] + ([self.residual] if self.residual is not None else [])

    def emit_forward(self):
        g, be, ge = self.g, self.g.be, self.geom
        w = g.param(self.wname)
        bias = g.param(self.bname) if self.bname else None
        ntap = ge.ks ** 3
        if self.gn:
            gamma, beta = g.param(self.gn[0]), g.param(self.gn[1])
            assert self.x.c == self.ci_real, (self.name, self.x.c, self.ci_real)
            stats = g.ensure_stats(self.x)
            nw = g.n
            self.mr = be.alloc((g.n, GROUPS if self.ci_real >= GROUPS else 1, 2), "f32")
        else:
            gamma = beta = stats = None
            nw = 1
        self.groups = GROUPS if self.ci_real >= GROUPS else 1
        # A GroupNorm conv on the LDS-tiled kernel folds the norm in the kernel's own prologue (rtp_conv_gn_fused): no fold launch
        # between two convs of a chain, no per-sample weight / bias-table buffers (options.fused_fold = 0: the separate launch, A/B)
        self.fold_fused = bool(self.gn and not self.out_fp32 and ge.ks == 3 and ge.stride == 1 and ge.pad == 1 and ge.ci == 32
                               and self.ci_real == 32 and ge.co in (16, 32) and self.x.cs == 32 and self.x.co == 0
                               and ge.di % 2 == 0 and ge.hi % 4 == 0 and ge.wi % 16 == 0 and not ge.w_ci_total
                               and hasattr(be, "conv_gn_fused") and bool(g.opt.fused_fold))
        self.wf = None if self.fold_fused else be.alloc((nw, ntap, ge.co, ge.ci), "bf16")
        need_btab = (bool(self.gn) or bias is not None) and not self.fold_fused
        self.btab = be.alloc((nw, 64, ge.co), "f32") if need_btab else None
        # training plans also need the data-gradient packing of the same weights
        need_dgrad = g.train and (self.x.needs_grad or bool(self.gn))
        self.wd = be.alloc((ntap, ge.ci, pad_to(ge.co, 32)), "bf16") if need_dgrad else None
        lane = g.lane_of(self.y)
        # Weight packing that does not depend on activations (everything of a conv without GroupNorm, the data-gradient
        # packing of every conv) is recorded for the ONE launch that opens the step (Graph.forward_list); only the
        # GroupNorm fold, which needs this step's statistics, is a launch of its own in front of its conv.
        if self.gn:
            if not self.fold_fused:
                g.emit_fwd(be.fold_fwd(w, bias, gamma, beta, stats, self.x.stats_split, self.groups, GN_EPS,
                                       ge, self.ci_real, self.co_real, self.wf, self.btab, self.mr, None),
                           lane, [stats], [self.wf, self.btab, self.mr], "fold:" + self.name)
            if self.wd is not None:
                g.head.append(("fold_fwd", w, None, None, None, None, 0, self.groups, GN_EPS, ge, self.ci_real,
                               self.co_real, None, None, None, self.wd))
        else:
            g.head.append(("fold_fwd", w, bias, None, None, None, 0, self.groups, GN_EPS, ge, self.ci_real,
                           self.co_real, self.wf, self.btab, None, self.wd))
        # Convs on the LDS-tiled kernel also emit (sum y, sum y^2) per channel from their epilogue, so a GroupNorm
        # consumer of y needs no statistics pass (ensure_stats finds them)
        fstats = None
        # wide 3x3x3 convs (64 / 128 channels: the feat64 backbone; stride 1 and the stride-2 forward) run as channel slices of the
        # LDS-tiled kernels, the partial sums carried in an fp32 workspace (rtp_conv_igemm_ws)
        self.sliced_fwd = (not self.fold_fused) and hasattr(be, "conv_sliced_ok") and be.conv_sliced_ok(self.x, ge, False)
        skw = dict(ws=True) if self.sliced_fwd else {}
        S = 0 if (self.out_fp32 or ge.co != self.y.c or not self.want_stats) else be.conv_stats_nsplit(
            self.x, g.with_width(ge, "conv:" + self.name), False, **skw)
        if S > 0:
            self.y.stats_split = S
            self.y.stats = be.alloc((g.n, S, self.y.c, 2), "f32")
            fstats = (None, self.y.stats)
        if self.fold_fused:
            self.wt = be.alloc((ntap, ge.co, ge.ci), "f32")   # tap-major fp32 copy of the master, made by the step's opening launch
            g.head.append(("pack_wt", w, self.co_real, ge.co, ge.ci, ntap, self.wt))
            g.emit_fwd(be.conv_gn_fused(self.x, self.wt, bias, gamma, beta, stats, self.x.stats_split, self.groups, GN_EPS, self.co_real,
                                        self.mr, self.residual, self.y, g.with_width(ge, "conv:" + self.name), self.relu,
                                        self.y.stats if fstats else None),
                       lane, [self.x, self.wt, stats, self.residual], [self.y, self.mr, self.y.stats if fstats else None],
                       "conv:" + self.name)
        else:
            kw = dict(ws=g.slice_ws(lane, g.n * self.y.vox * 32)) if self.sliced_fwd else {}
            g.emit_fwd(be.conv(self.x, self.wf, nw > 1, self.btab, self.residual, self.y, g.with_width(ge, "conv:" + self.name), self.relu,
                               False, self.out_fp32, fstats, **kw),
                       lane, [self.x, self.wf, self.btab, self.residual, kw.get("ws")],
                       [self.y, self.y.stats if fstats else None, kw.get("ws")], "conv:" + self.name)
        self.alg_flops = 2 * g.n * ge.do * ge.ho * ge.wo * self.co_real * self.ci_real * ntap
        g.flops["conv_fwd"] += self.alg_flops
        # mirrors the dispatch predicate of rtp_conv_tiled_try (csrc/conv_tiled.hip)
        brick = ge.ks == 3 and ge.stride == 1 and ge.di % 2 == 0 and ge.hi % 4 == 0 and ge.wi % 16 == 0
        s2_fwd = (ge.ks == 3 and ge.stride == 2 and ge.ci == 32 and ge.co == 32 and ge.di == 2 * ge.do and ge.hi == 2 * ge.ho
                  and ge.wi == 2 * ge.wo and ge.ho % 2 == 0 and min(ge.do, ge.ho, ge.wo) >= 2 and self.x.cs % 32 == 0
                  and hasattr(be, "conv_sliced_ok"))   # mirrors rtp_conv_s2_fwd_try (csrc/conv_s2_tiled.hip)
        self.tiled_fwd = (brick and ge.ci == 32 and ge.co in (16, 32) and self.x.cs == 32) or self.sliced_fwd or s2_fwd
        self.tiled_bwd = brick and ge.ci == 32 and pad_to(ge.co, 32) == 32
        # ... and of rtp_dgrad_s2_try (csrc/dgrad_s2_tiled.hip): stride-2 data gradients that write a 32-channel tensor of twice
        # the output's size
        self.s2_bwd = (ge.ks == 3 and ge.stride == 2 and ge.ci == 32 and pad_to(ge.co, 32) == 32 and ge.di == 2 * ge.do
                       and ge.hi == 2 * ge.ho and ge.wi == 2 * ge.wo and ge.ho % 2 == 0 and ge.wo % 16 == 0
)
        # (the 64 -> 64 stride-1 layers the sliced route hands to csrc/conv64_tiled.hip are a kernel family of their own)
        self.c64 = ge.ks == 3 and ge.stride == 1 and ge.ci == 64 and ge.co == 64
        fam_f = "conv64" if (self.sliced_fwd and self.c64) else "conv_tiled" if self.tiled_fwd else "conv_generic"
        g.flops[fam_f] += self.alg_flops
        esz = 4 if self.out_fp32 else 2
        self.bytes_fwd = 2 * g.n * self.x.vox * ge.ci + esz * g.n * self.y.vox * self.y.c + (
            2 * g.n * self.y.vox * self.y.c if self.residual is not None else 0)
        g.alg_bytes[fam_f] += self.bytes_fwd
        # the tiled kernel at its dominant geometry (csrc/conv_tiled.hip: 32 output channels, >= 2^20 voxels per launch)
        g.cost["conv:" + self.name] = (self.alg_flops, self.bytes_fwd)
        self.full_fwd = self.tiled_fwd and ge.co == 32 and g.n * self.y.vox >= (1 << 20)
        if self.full_fwd:
            g.flops["conv_tiled_full"] += self.alg_flops
            g.alg_bytes["conv_tiled_full"] += self.bytes_fwd

    def emit_backward(self, gy: View):
        g, be, ge, x = self.g, self.g.be, self.geom, self.x
        if self.residual is not None and self.residual.needs_grad:
            self.residual.contribs.append((gy, None))
        w = g.params[self.wname]
        if self._fusable(gy):
            return self._emit_backward_fused(gy)
        # ---- data gradient (and GroupNorm backward terms)
        if x.needs_grad or self.gn:
            cok = pad_to(ge.co, 32)
            assert gy.c >= cok, (self.name, gy.c, cok)
            wd = self.wd
            dxh_buf = be.alloc((g.n, x.d, x.h, x.w, ge.ci), "bf16")
            dxh = View(dxh_buf, g.n, x.d, x.h, x.w, ge.ci, 0, ge.ci)
            # GroupNorm backward needs P = sum dxhat and Q = sum dxhat*x per (sample, channel): the tiled kernel
            # accumulates them in its epilogue, otherwise a chan_stats pass over (dxhat, x) follows
            self.sliced_bwd = hasattr(be, "conv_sliced_ok") and be.conv_sliced_ok(gy, ge, True)
            S = be.conv_stats_nsplit(gy, g.with_width(ge, "dgrad:" + self.name), True, **(dict(ws=True) if self.sliced_bwd else {})) if self.gn else 0
            pq = be.alloc((g.n, S or x.stats_split, ge.ci, 2), "f32") if self.gn else None
            lane = g.lane_of(self.y)   # a stride-2 conv's data gradient runs with the LOWER-resolution group
            kw = dict(ws=g.slice_ws(lane, g.n * x.vox * 32)) if self.sliced_bwd else {}
            g.emit_bwd(be.conv(gy, wd, False, None, None, dxh, g.with_width(ge, "dgrad:" + self.name), False, True, False,
                               (x, pq) if S else None, **kw),
                       lane, [gy, wd, x if S else None, kw.get("ws")], [dxh, pq if S else None, kw.get("ws")], "dgrad:" + self.name)
            g.flops["conv_dgrad"] += self.alg_flops
            if self.sliced_bwd:
                self.tiled_bwd = True
            fam_b = "conv64" if (self.sliced_bwd and self.c64) else "conv_tiled" if (self.tiled_bwd or self.s2_bwd) else "conv_generic"
            g.flops[fam_b] += self.alg_flops
            # data gradient: read gy, write dxhat, (GroupNorm: read x for Q)
            nb = 2 * g.n * (gy.vox * pad_to(ge.co, 32) + x.vox * ge.ci * (2 if self.gn else 1))
            g.alg_bytes[fam_b] += nb
            if self.tiled_bwd and ge.ci == 32 and g.n * x.vox >= (1 << 20):   # transposed: the kernel's Cout is the conv's Cin
                g.flops["conv_tiled_full"] += self.alg_flops
                g.alg_bytes["conv_tiled_full"] += nb
                g.flops["conv_tiled_full_bwd"] += self.alg_flops
                g.alg_bytes["conv_tiled_full_bwd"] += nb
            if self.gn:
                if not S:
                    S = x.stats_split
                    g.emit_bwd(be.chan_stats(dxh, x, S, pq), lane, [dxh, x], [pq], "pq:" + self.name)
                coeff = be.alloc((g.n * ge.ci * 5,), "f32")  # [n][c][3] coefficients + [n][c][2] scratch
                # no launch yet: the fan-in pass of x computes them in its prologue when it can (LazyCoeff); the parameter sums
                # of the deferred tail read the partials that pass writes
                lz = LazyCoeff(g, self.name, lane, pq, S, self.mr, g.params[self.gn[0]], g.n, self.ci_real, self.groups, x.vox, coeff)
                g._lazy_by_coeff[coeff.data_ptr()] = lz
                g.tail_a.append(("gn_param", coeff, g.n, self.ci_real, g.pgrad[self.gn[0]], g.pgrad[self.gn[1]], 0))
                if x.needs_grad and ge.ci == self.ci_real:
                    x.contribs.append((dxh, lz))
                else:           # nobody downstream will ask for them (or padded channels): compute now for the parameter sums
                    lz.materialise()
                    if x.needs_grad:
                        x.contribs.append((dxh, coeff))
            elif x.needs_grad:
                x.contribs.append((dxh, None))
        self._emit_wgrad_unfused(gy)

    def _emit_wgrad_unfused(self, gy: View):
        g, be, ge, x = self.g, self.g.be, self.geom, self.x
        # ---- weight gradient
        S = be.wgrad_nsplit(g.with_width(ge, "wgrad:" + self.name)) if (x.cs % 32 == 0 and x.co % 8 == 0 and gy.cs % 32 == 0
                                                                        and (ge.ci > 32 or (x.cs == 32 and x.co == 0))) else 0
        self.tiled_wgrad = S > 0
        S = S or wgrad_split(gy.vox)
        co32 = pad_to(ge.co, 32)
        assert gy.c == co32, (self.name, gy.c, co32)
        gp = be.alloc((g.n, S, ge.ks ** 3, co32, ge.ci), "f32")
        # The weight-gradient chain (wgrad -> class sums -> un-fold) only feeds the optimiser, so it runs on its own
        # lane beside the rest of the backward sweep.
        wl = g.wg_lane_of(gy)
        # a conv with bias and without GroupNorm on the tiled weight-gradient kernel: its loader waves sum gy as a by-product (tg),
        # the bias gradient is read off those sums in the tail -- no class-sum pass over gy (the head towers: 84 MB each)
        tg = self._bias_tg(S) if self.tiled_wgrad else None
        gw = g.with_width(ge, "wgrad:" + self.name)
        if tg is not None:
            g.emit_bwd(be.wgrad_tg(gy, x, gw, S, gp, tg), wl, [gy, x], [gp, tg], "wgrad:" + self.name)
        else:
            g.emit_bwd(be.wgrad(gy, x, gw, S, gp), wl, [gy, x], [gp], "wgrad:" + self.name)
        g.flops["wgrad"] += self.alg_flops
        g.flops["wgrad_tiled" if self.tiled_wgrad else "wgrad_generic"] += self.alg_flops
        g.alg_bytes["wgrad_tiled" if self.tiled_wgrad else "wgrad_generic"] += 2 * g.n * (gy.vox * co32 + x.vox * ge.ci) + (
            4 * g.n * S * ge.ks ** 3 * co32 * ge.ci)
        # Everything after the correlation itself (class-sum reduction, slab fold + GroupNorm un-fold) only feeds the
        # optimiser: recorded here, run once for all layers at the end of the sweep (Graph.emit_tail).
        csum = g.class_sums_for(self.y, gy, wl, self.name) if ((self.gn or self.bname) and tg is None) else None
        g.tail_b.append(("wgrad_fold", gp, S, csum, self.mr, g.params[self.gn[0]] if self.gn else None,
                         g.params[self.gn[1]] if self.gn else None, self.groups, ge, self.ci_real,
                         self.co_real, g.pgrad[self.wname], g.pgrad[self.bname] if self.bname else None, 0, tg))

    def _bias_tg(self, S):
        """Subset-sum buffer for rtp_wgrad_tg when this conv qualifies (bias, no GroupNorm, 32 -> <= 32 channels, stride 1, tiled)."""
        g, be, ge = self.g, self.g.be, self.geom
        if (self.gn or not self.bname or not hasattr(be, "wgrad_tg") or ge.ci != 32 or pad_to(ge.co, 32) != 32 or ge.stride != 1
                or ge.ks != 3 or self.x.cs != 32 or self.x.co != 0):
            return None
        return be.alloc((g.n, S, 27, 32), "f32")


    # ------------------------------------------------------------------ fused backward (no fan-in pass)
    def _fusable(self, gy):
        """This conv's data gradient can write the FINISHED gradient of x: it runs on the LDS-tiled kernel, it is the last
        contribution to x in the sweep, and x's other contributions fit the kernel's epilogue."""
        g, be, ge, x = self.g, self.g.be, self.geom, self.x
        # The fused route for the stride-2 data gradients (Q from the generic slabs by rtp_qpart_from_slabs, P from gy's class
        # sums) is built and tested but OFF by default: measured 6.41 ms/step against 6.31 with the same kernel writing dxhat +
        # statistics and the fan-in pass kept (the slab contraction and the early class sums cost more than the two passes
        # they remove), and P inherits the LDS-atomic order of the class-sum scan (1e-6 instead of 1e-9 run to run).
        s2 = self.s2_bwd and bool(g.opt.fused_s2)
        if not (g.fused_dgrad and (self.tiled_bwd or s2) and x.needs_grad and g.first_consumer.get(id(x)) is self):
            return False
        if self.residual is x or x.c != 32 or x.cs != 32 or x.co != 0 or gy.c < 32:
            return False
        if self.gn and (self.ci_real != 32 or ge.ci != self.ci_real):   # the fused prologue's GroupNorm algebra is 32 real channels
            return False
        if len(x.contribs) > 3 or any(v.c < 32 or (cf is not None and v.c != 32) for v, cf in x.contribs):
            return False
        if ge.stride == 2:   # Q from the generic weight-gradient kernel's slabs (rtp_qpart_from_slabs), P from gy's class sums
            return gy.c == 32 and hasattr(be, "qpart_from_slabs") and be.conv_dgrad_fused_ok(gy, ge)
        if self.gn and not (be.wgrad_nsplit(g.with_width(ge, "wgrad:" + self.name)) > 0 and gy.c == 32 and pad_to(ge.co, 32) == 32):
            return False   # Q comes from the tiled weight-gradient kernel's slabs
        return be.conv_tiled_ok(gy, ge, True)

    def _emit_backward_fused(self, gy: View):
        """GroupNorm conv:  class sums of gy (P) | weight gradient + slab contraction (Q)  ->  coefficients  ->  data
        gradient whose epilogue evaluates A*dxhat + B*x + C, adds x's other contributions and applies x's ReLU mask.
        Conv without GroupNorm (head towers): the data gradient adds the other contributions and the mask."""
        g, be, ge, x = self.g, self.g.be, self.geom, self.x   # (the residual's contribution was added by emit_backward)
        lane, wl = g.lane_of(self.y), g.wg_lane_of(gy)
        co32 = pad_to(ge.co, 32)
        assert gy.c == co32, (self.name, gy.c, co32)
        ntap = ge.ks ** 3
        # ---- per-boundary-class sums of gy: bias / un-fold need them, and now P does too.
        # A stride-1 GroupNorm conv: the weight-gradient kernel's loader waves sum gy over the volume / faces / edges / corners
        # (tg), from which the data gradient's prologue derives P and the class sums -- no pass over gy, no launch between the
        # two kernels.  (Rounds 2-3 kept two intermediate routes for A/B -- coefficients by a kernel of their own, class sums + P by
        # a scan kernel -- both slower on the main chain; removed in round 6 with their switches.)
        from_wgrad = bool(self.gn) and ge.stride == 1
        csum = tg = None
        if ge.stride == 2:
            csum = g.class_sums_for(self.y, gy, wl, self.name, early=True) if (self.gn or self.bname) else None
        elif from_wgrad:
            tg = be.alloc((g.n, be.wgrad_nsplit(g.with_width(ge, "wgrad:" + self.name)), 27, 32), "f32")   # one partial table per weight-gradient slab
            csum = be.alloc((g.n, 64, gy.c), "f32")
        S = be.wgrad_nsplit(g.with_width(ge, "wgrad:" + self.name)) if x.cs == 32 and x.co == 0 else 0
        self.tiled_wgrad = S > 0
        S = S or wgrad_split(gy.vox)
        btg = self._bias_tg(S) if (self.tiled_wgrad and not self.gn and ge.stride == 1) else None
        if csum is None and tg is None and (self.gn or self.bname) and btg is None:
            csum = g.class_sums_for(self.y, gy, wl, self.name, early=bool(self.gn))
        # ---- weight gradient (for a GroupNorm conv it now precedes the data gradient: its slabs give Q)
        gp = be.alloc((g.n, S, ntap, co32, ge.ci), "f32")
        coeff = gnq = None
        gw, gd = g.with_width(ge, "wgrad:" + self.name), g.with_width(ge, "dgrad:" + self.name)
        if self.gn and ge.stride == 2:
            # the generic weight-gradient kernel's slabs, contracted with the weights by a small launch; P in the data gradient's
            # prologue from gy's boundary-class sums
            qpart = be.alloc((g.n, S, ge.ci), "f32")
            g.emit_bwd(be.wgrad(gy, x, gw, S, gp), lane, [gy, x], [gp], "wgrad:" + self.name)
            g.emit_bwd(be.qpart_from_slabs(gp, g.n, S, ntap, co32, ge.ci, self.wd, qpart), lane, [gp, self.wd], [qpart],
                       "qslab:" + self.name)
            coeff = be.alloc((g.n * ge.ci * 5,), "f32")
            gnq = dict(qpart=qpart, q_nsplit=S, p=None, tg=None, csum_out=None, csum=csum, mr=self.mr, gamma=g.params[self.gn[0]],
                       groups=self.groups, coeff_out=coeff)
            g.tail_a.append(("gn_param", coeff, g.n, self.ci_real, g.pgrad[self.gn[0]], g.pgrad[self.gn[1]], 0))
        elif self.gn:
            qpart = be.alloc((g.n, S, ge.ci), "f32")
            g.emit_bwd(be.wgrad_q(gy, x, gw, S, gp, self.wd, qpart, tg), lane, [gy, x, self.wd], [gp, qpart, tg], "wgrad:" + self.name)
            coeff = be.alloc((g.n * ge.ci * 5,), "f32")
            # Q, P (from tg) and the coefficients in the data gradient's own prologue
            gnq = dict(qpart=qpart, q_nsplit=S, p=None, tg=tg, csum_out=csum, mr=self.mr,
                       gamma=g.params[self.gn[0]], groups=self.groups, coeff_out=coeff)
            g.tail_a.append(("gn_param", coeff, g.n, self.ci_real, g.pgrad[self.gn[0]], g.pgrad[self.gn[1]], 0))
        elif btg is not None:
            g.emit_bwd(be.wgrad_tg(gy, x, gw, S, gp, btg), wl, [gy, x], [gp, btg], "wgrad:" + self.name)
        else:
            g.emit_bwd(be.wgrad(gy, x, gw, S, gp), wl, [gy, x], [gp], "wgrad:" + self.name)
        g.flops["wgrad"] += self.alg_flops
        g.flops["wgrad_tiled" if self.tiled_wgrad else "wgrad_generic"] += self.alg_flops
        g.alg_bytes["wgrad_tiled" if self.tiled_wgrad else "wgrad_generic"] += 2 * g.n * (gy.vox * co32 + x.vox * ge.ci) + (
            4 * g.n * S * ntap * co32 * ge.ci)
        g.tail_b.append(("wgrad_fold", gp, S, csum, self.mr, g.params[self.gn[0]] if self.gn else None,
                         g.params[self.gn[1]] if self.gn else None, self.groups, ge, self.ci_real,
                         self.co_real, g.pgrad[self.wname], g.pgrad[self.bname] if self.bname else None, 0, btg))
        # ---- data gradient -> finished gradient of x
        terms = [(v, cf.materialise() if isinstance(cf, LazyCoeff) else cf) for v, cf in x.contribs]
        dx_buf = be.alloc((g.n, x.d, x.h, x.w, ge.ci), "bf16")
        dx = View(dx_buf, g.n, x.d, x.h, x.w, ge.ci, 0, ge.ci)
        reads = [gy, self.wd, x, coeff] + [v for v, _ in terms] + [cf for _, cf in terms]
        # x's producer will want the per-boundary-class sums of this gradient: emit the per-channel totals here
        tot, prod = None, x.producer
        prod_from_wgrad = isinstance(prod, ConvOp) and prod.gn and prod.tiled_bwd and g.fused_dgrad
        if (isinstance(prod, (ConvOp, SplitConvOp)) and (prod.gn or prod.bname) and hasattr(be, "class_sums_boundary")
                and not prod_from_wgrad):
            ts = be.conv_stats_nsplit(gy, g.with_width(ge, "dgrad:" + self.name), True)
            if ts > 0:
                tot = be.alloc((g.n, ts, 32), "f32")
                x.grad_tot = (ts, tot)
        if gnq is not None:
            reads += [gnq["qpart"], gnq["p"], gnq["tg"], gnq.get("csum"), self.mr]
            g.emit_bwd(be.conv_dgrad_fused(gy, self.wd, x, None, terms, x.relu, dx, gd, tot, gnq), lane, reads,
                       [dx_buf, tot, coeff, gnq["csum_out"]], "dgrad:" + self.name)
        else:
            g.emit_bwd(be.conv_dgrad_fused(gy, self.wd, x, coeff, terms, x.relu, dx, gd, tot), lane, reads, [dx_buf, tot],
                       "dgrad:" + self.name)
        x.grad, x.final, x.contribs = dx, True, []
        g.flops["conv_dgrad"] += self.alg_flops
        g.flops["conv_tiled"] += self.alg_flops
        # read gy + x (+ the other contributions), write dx
        nb = 2 * g.n * (gy.vox * co32 + x.vox * ge.ci * (2 + len(terms)))
        g.alg_bytes["conv_tiled"] += nb
        g.cost["dgrad:" + self.name] = (self.alg_flops, nb)
        if g.n * x.vox >= (1 << 20):
            g.flops["conv_tiled_full"] += self.alg_flops
            g.alg_bytes["conv_tiled_full"] += nb
            g.flops["conv_tiled_full_bwd"] += self.alg_flops
            g.alg_bytes["conv_tiled_full_bwd"] += nb


class SplitConvOp:
    """Conv3d(Cin = 32*K, Cout in {16, 32}, 3x3x3, bias) [+ ReLU] as K input-channel slices on the LDS-tiled kernels.
    Forward: slice k adds the fp32 partial sum of the slices before it (in place, one fp32 buffer); the last slice adds
    the bias, applies the ReLU and writes bf16.  Backward: K independent data-gradient launches, each writing its
    32-channel slice of the input gradient, and K weight-gradient launches over the matching slices of x."""

    def __init__(self, g, name, x, y, gs, wname, bname, relu, ci_real, co_real, sources=None):
        self.g, self.name, self.x, self.y, self.gs = g, name, x, y, gs
        self.wname, self.bname, self.relu, self.ci_real, self.co_real = wname, bname, relu, ci_real, co_real
        self.gn = None
        # sources: [(activation, real channels, first weight input channel)] -- separate 32-channel tensors standing in for the
        # slices of one wide tensor (Graph.conv_cat); None: the K = Cin / 32 slices of x
        self.sources = sources
        self.K = len(sources) if sources else ci_real // 32
        # partner: the OTHER head tower's first conv over the same feature (Graph.conv_pair): the two run as ONE 64-wide launch per
        # 64-channel slice of the feature (rtp_conv64_blocks), forward and data gradient; the weight gradients stay per tower
        self.partner = None
        self._pair_gy = None

    def inputs(self):
        return [s[0] for s in self.sources] if self.sources else [self.x]

    def slice_real(self, k):
        return self.sources[k][1] if self.sources else 32

    def slice_geom(self, k):
        s = self.gs
        off = self.sources[k][2] if self.sources else 32 * k
        return Geom(s.n, s.di, s.hi, s.wi, s.do, s.ho, s.wo, 32, s.co, s.ks, s.stride, s.pad, self.ci_real, off)

    def x_slice(self, k):
        if self.sources:
            x = self.sources[k][0]
            return View(x.buf, x.n, x.d, x.h, x.w, x.cs, 0, 32)
        x = self.x
        return View(x.buf, x.n, x.d, x.h, x.w, x.cs, 32 * k, 32)

    def emit_forward(self, launches=True):
        """launches=False: the weight images only (the launches come from emit_pair_forward)."""
        g, be, gs = self.g, self.g.be, self.gs
        w = g.param(self.wname)
        bias = g.param(self.bname) if self.bname else None
        ntap = gs.ks ** 3
        need_dgrad = g.train and any(a.needs_grad for a in self.inputs())
        self.wf, self.wd, self.btab = [], [], None
        lane = g.lane_of(self.y)
        acc = accv = None
        if launches:
            acc = be.alloc((g.n, self.y.vox, gs.co), "f32")
            accv = View(acc.view(g.n, self.y.d, self.y.h, self.y.w, gs.co), g.n, self.y.d, self.y.h, self.y.w, gs.co, 0, gs.co)
        for k in range(self.K):
            last = k == self.K - 1
            gk = self.slice_geom(k)
            wf = be.alloc((1, ntap, gs.co, 32), "bf16")
            wd = be.alloc((ntap, 32, pad_to(gs.co, 32)), "bf16") if need_dgrad else None
            bt = be.alloc((1, 64, gs.co), "f32") if (last and bias is not None) else None
            self.wf.append(wf)
            self.wd.append(wd)
            if bt is not None:
                self.btab = bt
            g.head.append(("fold_fwd", w, bias if bt is not None else None, None, None, None, 0, 1, GN_EPS, gk, self.slice_real(k),
                           self.co_real, wf, bt, None, wd))
            if not launches:
                continue
            g.emit_fwd(be.conv(self.x_slice(k), wf, False, bt, None, self.y if last else accv, gk, self.relu and last, False,
                               not last, None, (acc, gs.co) if k > 0 else None),
                       lane, [self.inputs()[k] if self.sources else self.x, wf, bt, acc if k > 0 else None],
                       [self.y if last else acc], "conv:%s.%d" % (self.name, k))
        self.alg_flops = 2 * g.n * gs.do * gs.ho * gs.wo * self.co_real * self.ci_real * ntap
        g.flops["conv_fwd"] += self.alg_flops
        g.flops["conv_tiled" if launches else "conv64"] += self.alg_flops   # (no launches of its own: paired on the 64-wide kernel)
        nb = 2 * g.n * (self.x.vox * (32 * self.K if self.sources else self.ci_real) + self.y.vox * self.y.c)
        g.alg_bytes["conv_tiled" if launches else "conv64"] += nb
        self.full = launches and gs.co == 32 and g.n * self.y.vox >= (1 << 20)   # the profiling family of these launches (conv_tiled.hip)
        if self.full:
            g.flops["conv_tiled_full"] += self.alg_flops
            g.alg_bytes["conv_tiled_full"] += nb

    def emit_pair_forward(self):
        """Both towers (self, self.partner) as ONE launch per 64-channel slice of the feature: tower h = output half h, the partial
        sums of a feature wider than 64 channels travel through one fp32 buffer [n][voxels][64]."""
        g, be, gs, a, b, x = self.g, self.g.be, self.gs, self, self.partner, self.x
        lane = g.lane_of(a.y)
        M = self.K // 2
        acc = be.alloc((g.n, a.y.vox, 64), "f32") if M > 1 else None
        g64 = g.with_width(Geom(gs.n, gs.di, gs.hi, gs.wi, gs.do, gs.ho, gs.wo, 64, 64, 3, 1, 1), "conv:%s+%s" % (a.name, b.name))
        ys = [View(t.y.buf, g.n, t.y.d, t.y.h, t.y.w, t.y.cs, 0, 32) for t in (a, b)]
        for m in range(M):
            last = m == M - 1
            xs = [View(x.buf, x.n, x.d, x.h, x.w, x.cs, 64 * m + 32 * k, 32) for k in range(2)]
            blocks = [[(t.wf[2 * m + k], 0) for k in range(2)] for t in (a, b)]
            bts = [(a.btab, 0), (b.btab, 0)] if (last and a.btab is not None) else None
            g.emit_fwd(be.conv64_blocks(xs, blocks, 32, gs.co * 32, bts, gs.co, None, ys if last else None, g64, a.relu and last, False,
                                        acc, m > 0, not last),
                       lane, [x, acc if m > 0 else None] + [t.wf[2 * m + k] for t in (a, b) for k in range(2)] + ([a.btab, b.btab] if bts else []),
                       [a.y, b.y] if last else [acc], "conv:%s+%s.%d" % (a.name, b.name, m))

    def _emit_pair_dgrad(self, gys):
        """The data gradients of both towers as ONE launch per 64 input channels, summed: input half t = tower t's output gradient."""
        g, be, gs, a, b, x = self.g, self.g.be, self.gs, self, self.partner, self.x
        lane = g.lane_of(a.y)
        towers = (a, b) if a._pair_first else (b, a)   # creation order: the order the forward used
        gv = [View(gys[id(t)].buf, g.n, x.d, x.h, x.w, gys[id(t)].cs, gys[id(t)].co, 32) for t in towers]
        dxb = be.alloc((g.n, x.d, x.h, x.w, self.ci_real), "bf16")
        g64 = g.with_width(Geom(gs.n, gs.di, gs.hi, gs.wi, gs.do, gs.ho, gs.wo, 64, 64, 3, 1, 1), "dgrad:%s+%s" % (towers[0].name, towers[1].name))
        for j in range(self.K // 2):
            blocks = [[(t.wd[2 * j + hh], 0) for t in towers] for hh in range(2)]
            ys = [View(dxb, g.n, x.d, x.h, x.w, self.ci_real, 64 * j + 32 * hh, 32) for hh in range(2)]
            g.emit_bwd(be.conv64_blocks(gv, blocks, 32, 32 * 32, None, 32, None, ys, g64, False, True), lane,
                       [gys[id(t)] for t in towers] + [t.wd[2 * j + hh] for t in towers for hh in range(2)], [dxb],
                       "dgrad:%s+%s.%d" % (towers[0].name, towers[1].name, j))
        x.contribs.append((View(dxb, g.n, x.d, x.h, x.w, self.ci_real, 0, self.ci_real), None))
        for t in towers:
            g.flops["conv_dgrad"] += t.alg_flops
            g.flops["conv64"] += t.alg_flops
            g.alg_bytes["conv64"] += 2 * g.n * (x.vox * 32 + x.vox * self.ci_real // 2)

    def emit_backward(self, gy: View):
        g, be, gs, x = self.g, self.g.be, self.gs, self.x
        co32 = pad_to(gs.co, 32)
        assert gy.c >= co32, (self.name, gy.c, co32)
        lane = g.lane_of(self.y)
        if self.partner is not None and x.needs_grad:
            # the tower the sweep reaches second emits the shared data-gradient launches; each tower its own weight gradients
            self._pair_gy = gy
            if self.partner._pair_gy is not None:
                self._emit_pair_dgrad({id(self): gy, id(self.partner): self.partner._pair_gy})
        elif self.sources:   # separate tensors: a data gradient per source that wants one
            for k, (a, _real, _off) in enumerate(self.sources):
                if not a.needs_grad:
                    continue
                dk_buf = be.alloc((g.n, a.d, a.h, a.w, 32), "bf16")
                dk = View(dk_buf, g.n, a.d, a.h, a.w, 32, 0, 32)
                g.emit_bwd(be.conv(gy, self.wd[k], False, None, None, dk, self.slice_geom(k), False, True, False), lane,
                           [gy, self.wd[k]], [dk_buf], "dgrad:%s.%d" % (self.name, k))
                a.contribs.append((dk, None))
            if any(a.needs_grad for a in self.inputs()):
                g.flops["conv_dgrad"] += self.alg_flops
                g.flops["conv_tiled"] += self.alg_flops
        elif x.needs_grad:
            dxb = be.alloc((g.n, x.d, x.h, x.w, self.ci_real), "bf16")
            for k in range(self.K):
                dk = View(dxb, g.n, x.d, x.h, x.w, self.ci_real, 32 * k, 32)
                g.emit_bwd(be.conv(gy, self.wd[k], False, None, None, dk, self.slice_geom(k), False, True, False), lane,
                           [gy, self.wd[k]], [dxb], "dgrad:%s.%d" % (self.name, k))
            x.contribs.append((View(dxb, g.n, x.d, x.h, x.w, self.ci_real, 0, self.ci_real), None))
            g.flops["conv_dgrad"] += self.alg_flops
            g.flops["conv_tiled"] += self.alg_flops
            nb = 2 * g.n * (gy.vox * co32 + x.vox * self.ci_real)
            g.alg_bytes["conv_tiled"] += nb
            if g.n * x.vox >= (1 << 20):
                g.flops["conv_tiled_full"] += self.alg_flops
                g.alg_bytes["conv_tiled_full"] += nb
                g.flops["conv_tiled_full_bwd"] += self.alg_flops
                g.alg_bytes["conv_tiled_full_bwd"] += nb
        wl = g.wg_lane_of(gy)
        csum = g.class_sums_for(self.y, gy, wl, self.name) if self.bname else None
        for k in range(self.K):
            gk = self.slice_geom(k)
            S = be.wgrad_nsplit(gk) or wgrad_split(gy.vox)
            gp = be.alloc((g.n, S, gs.ks ** 3, co32, 32), "f32")
            g.emit_bwd(be.wgrad(gy, self.x_slice(k), gk, S, gp), wl, [gy, self.inputs()[k] if self.sources else x], [gp],
                       "wgrad:%s.%d" % (self.name, k))
            first = k == 0 and self.bname
            g.tail_b.append(("wgrad_fold", gp, S, csum if first else None, None, None, None, 1, gk, self.slice_real(k), self.co_real,
                             g.pgrad[self.wname], g.pgrad[self.bname] if first else None, 0))
            g.alg_bytes["wgrad_tiled"] += 2 * g.n * (gy.vox * co32 + x.vox * 32) + 4 * g.n * S * gs.ks ** 3 * co32 * 32
        g.flops["wgrad"] += self.alg_flops
        g.flops["wgrad_tiled"] += self.alg_flops


class CoSplitConvOp:
    """Conv3d(32, Cout in (32, 48], 3x3x3, bias) with fp32 output as two output-channel slices (32 + the rest, padded to
    16) on the LDS-tiled kernels.  Forward and weight gradient are independent per slice (parameter / gradient views);
    the data gradient contracts over the output channels, so its two launches chain through an fp32 partial sum."""

    def __init__(self, g, name, x, y, wname, bname, relu, co_real):
        self.g, self.name, self.x, self.y = g, name, x, y
        self.wname, self.bname, self.relu, self.co_real = wname, bname, relu, co_real
        self.gn = None
        self.slices = [(0, 32), (32, co_real - 32)]

    def inputs(self):
        return [self.x]

    def geom(self, c):
        x, y = self.x, self.y
        return Geom(self.g.n, x.d, x.h, x.w, y.d, y.h, y.w, 32, pad_to(c, 16), 3, 1, 1)

    def emit_forward(self):
        g, be, x, y = self.g, self.g.be, self.x, self.y
        w, bias = g.param(self.wname), g.param(self.bname)
        need_dgrad = g.train and x.needs_grad
        lane = g.lane_of(y)
        self.wd = []
        for k, (a, c) in enumerate(self.slices):
            gk = self.geom(c)
            wf = be.alloc((1, 27, gk.co, 32), "bf16")
            bt = be.alloc((1, 64, gk.co), "f32")
            wd = be.alloc((27, 32, 32), "bf16") if need_dgrad else None
            self.wd.append(wd)
            g.head.append(("fold_fwd", w[a:a + c], bias[a:a + c], None, None, None, 0, 1, GN_EPS, gk, 32, c, wf, bt, None, wd))
            yk = View(y.buf, y.n, y.d, y.h, y.w, y.cs, a, gk.co)
            g.emit_fwd(be.conv(x, wf, False, bt, None, yk, gk, self.relu, False, True), lane, [x, wf, bt], [y],
                       "conv:%s.co%d" % (self.name, k))
        self.alg_flops = 2 * g.n * y.vox * self.co_real * 32 * 27
        g.flops["conv_fwd"] += self.alg_flops
        g.flops["conv_tiled"] += self.alg_flops
        g.alg_bytes["conv_tiled"] += 2 * g.n * x.vox * 32 + 4 * g.n * y.vox * y.c
        self.big = g.n * y.vox >= (1 << 20)
        if self.big:   # the 32-channel slice runs in the dominant-geometry family, the 16-channel slice in the other
            g.flops["conv_tiled_full"] += self.alg_flops * 32 // self.co_real
            g.alg_bytes["conv_tiled_full"] += 2 * g.n * x.vox * 32 + 4 * g.n * y.vox * 32

    def emit_backward(self, gy: View):
        g, be, x, y = self.g, self.g.be, self.x, self.y
        assert gy.c >= 64, (self.name, gy.c)
        lane = g.lane_of(y)
        gys = [View(gy.buf, gy.n, gy.d, gy.h, gy.w, gy.cs, gy.co + 32 * k, 32) for k in range(2)]
        if x.needs_grad:
            acc = be.alloc((g.n, x.vox, 32), "f32")
            accv = View(acc.view(g.n, x.d, x.h, x.w, 32), g.n, x.d, x.h, x.w, 32, 0, 32)
            dxb = be.alloc((g.n, x.d, x.h, x.w, 32), "bf16")
            dxh = View(dxb, g.n, x.d, x.h, x.w, 32, 0, 32)
            g.emit_bwd(be.conv(gys[0], self.wd[0], False, None, None, accv, self.geom(self.slices[0][1]), False, True, True),
                       lane, [gy, self.wd[0]], [acc], "dgrad:%s.co0" % self.name)
            g.emit_bwd(be.conv(gys[1], self.wd[1], False, None, None, dxh, self.geom(self.slices[1][1]), False, True, False,
                               None, (acc, 32)), lane, [gy, self.wd[1], acc], [dxb], "dgrad:%s.co1" % self.name)
            x.contribs.append((dxh, None))
            g.flops["conv_dgrad"] += self.alg_flops
            g.flops["conv_tiled"] += self.alg_flops
            g.alg_bytes["conv_tiled"] += 2 * g.n * (gy.vox * 64 + x.vox * 32)
            if self.big:
                g.flops["conv_tiled_full"] += self.alg_flops
                g.alg_bytes["conv_tiled_full"] += 2 * g.n * (gy.vox * 64 + x.vox * 32)
                g.flops["conv_tiled_full_bwd"] += self.alg_flops
                g.alg_bytes["conv_tiled_full_bwd"] += 2 * g.n * (gy.vox * 64 + x.vox * 32)
        wl = g.wg_lane_of(gy)
        for k, (a, c) in enumerate(self.slices):
            gk = self.geom(c)
            S = be.wgrad_nsplit(gk) or wgrad_split(gy.vox)
            gp = be.alloc((g.n, S, 27, 32, 32), "f32")
            g.emit_bwd(be.wgrad(gys[k], x, gk, S, gp), wl, [gy, x], [gp], "wgrad:%s.co%d" % (self.name, k))
            cs_split = cls_split(gy.d, gy.h)
            cs_scratch = be.alloc((g.n, cs_split, 64, 32), "f32")
            csum = be.alloc((g.n, 64, 32), "f32")
            g.emit_bwd(be.class_sums(gys[k], cs_split, cs_scratch, None), wl, [gy], [cs_scratch], "cls:%s.co%d" % (self.name, k))
            g.tail_a.append(("class_reduce", cs_scratch, cs_split, g.n, 32, csum))
            g.tail_b.append(("wgrad_fold", gp, S, csum, None, None, None, 1, gk, 32, c, g.pgrad[self.wname][a:a + c],
                             g.pgrad[self.bname][a:a + c], 0))
            g.alg_bytes["wgrad_tiled"] += 2 * g.n * (gy.vox * 32 + x.vox * 32) + 4 * g.n * S * 27 * 32 * 32
        g.flops["wgrad"] += self.alg_flops
        g.flops["wgrad_tiled"] += self.alg_flops


class DcnAdaptOp:
    """graph.Graph.dcn_adapt: one forward launch sequence and one backward launch sequence of the native DCN operator."""

    def __init__(self, g, name, x, off, y, prefix):
        self.g, self.name, self.x, self.off, self.y, self.prefix = g, name, x, off, y, prefix
        self.gn = self.bname = None

    def inputs(self):
        return [self.x, self.off]

    def emit_backward(self, gy: View):
        g, x, off = self.g, self.x, self.off
        gx_buf = g.be.alloc((g.n, x.d, x.h, x.w, x.c), "bf16")
        gx = View(gx_buf, g.n, x.d, x.h, x.w, x.c, 0, x.c)
        oc = pad_to(pad_to(off.c_real, 16), 32)   # the offset conv's data gradient contracts full 32-channel K steps
        go_buf = g.be.alloc((g.n, x.d, x.h, x.w, oc), "bf16")
        go = View(go_buf, g.n, x.d, x.h, x.w, oc, 0, oc)
        gw = g.pgrad[self.prefix + ".conv_adaption.weight"]
        g.emit_bwd(self.make_bwd(gy, gx, go, gw), getattr(self, "lane", g.lane_of(self.y)),
                   [gy, x, off] + list(getattr(self.make_bwd, "planes", ())), [gx_buf, go_buf, gw], "dcn_bwd:" + self.name)
        if x.needs_grad:
            x.contribs.append((gx, None))
        off.contribs.append((go, None))


class ConcatOp:
    """Graph.concat: channel concatenation with upsampling; the adjoint is a channel slice (and the upsample adjoint)."""

    def __init__(self, g, terms, y):
        self.g, self.terms, self.y = g, terms, y

    def inputs(self):
        return list(self.terms)

    def emit_backward(self, gy: View):
        g, be = self.g, self.g.be
        off = 0
        for t in self.terms:
            gk = View(gy.buf, gy.n, gy.d, gy.h, gy.w, gy.cs, gy.co + off, t.c)
            off += t.c
            if not t.needs_grad:
                continue
            if t.dims == self.y.dims:
                t.contribs.append((gk, None))
            else:
                glow_buf = be.alloc((g.n, t.d, t.h, t.w, t.c), "bf16")
                glow = View(glow_buf, g.n, t.d, t.h, t.w, t.c, 0, t.c)
                g.emit_bwd(be.upsample_bwd(gk, glow), g.lane_of(glow), [gy], [glow], "upbwd:" + t.name)
                t.contribs.append((glow, None))


class FuseOp:
    def __init__(self, g, terms, y):
        self.g, self.terms, self.y = g, terms, y

    def inputs(self):
        return list(self.terms)

    def emit_backward(self, gy: View):
        g, be = self.g, self.g.be
        for t in self.terms:
            if not t.needs_grad:
                continue
            if t.dims == self.y.dims:
                t.contribs.append((gy, None))
            else:
                glow_buf = be.alloc((g.n, t.d, t.h, t.w, t.c), "bf16")
                glow = View(glow_buf, g.n, t.d, t.h, t.w, t.c, 0, t.c)
                g.emit_bwd(be.upsample_bwd(gy, glow), g.lane_of(glow), [gy], [glow], "upbwd:" + t.name)
                t.contribs.append((glow, None))
