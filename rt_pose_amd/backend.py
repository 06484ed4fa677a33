"""HipBackend: binds the graph's kernel calls to librtp_hip.so through ctypes.

PyTorch is used only for device memory (torch.zeros / torch.empty on cuda:N) and for the stream handle; every
method marshals its arguments ONCE and returns a closure f(stream_ptr) that performs the launch.
There is deliberately no CPU path here: constructing the backend without a GPU or without the built library
raises.
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import RtpAct, RtpConvGeom, RtpTerm, check

_DT = {"bf16": torch.bfloat16, "f32": torch.float32, "i64": torch.int64, "u8": torch.uint8, "i32": torch.int32}


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _act(v):
    if v is None:
        return None
    return RtpAct(v.buf.data_ptr(), v.cs, v.co, v.c)


def _geom(g):
    return RtpConvGeom(g.n, g.di, g.hi, g.wi, g.do, g.ho, g.wo, g.ci, g.co, g.ks, g.stride, g.pad, g.w_ci_total,
                       g.w_ci_off, getattr(g, "wgs", 0))


class _MultiLaunch:
    """A shared launch (rtp_multi_*): callable like every other launch closure; owns the handle and its device parameter block."""

    def __init__(self, lib, handle, params, fns):
        self.lib, self.handle, self.params, self.fns = lib, handle, params, fns

    def __call__(self, s):
        check(self.lib.rtp_multi_launch(self.handle, s), "rtp_multi_launch")

    def __del__(self):
        try:
            if self.handle >= 0:
                # rtp_multi_free requires that no launch of the handle is in flight (include/rtp.h): wait for the device first
                if torch.cuda.is_available():
                    torch.cuda.synchronize(self.params.device)
                self.lib.rtp_multi_free(self.handle)
                self.handle = -1
        except Exception:   # interpreter shutdown: the library may be gone already
            pass


class _LaneEvent:
    """hipEvent_t behind the C ABI (rtp_event_*): record(stream handle) / wait(stream handle), both c_void_p stream pointers."""
    __slots__ = ("lib", "h")

    def __init__(self, lib, system_fence):
        self.lib = lib
        h = C.c_void_p()
        check(lib.rtp_event_create(C.byref(h), int(system_fence)), "rtp_event_create")
        self.h = h

    def record(self, stream_ptr):
        check(self.lib.rtp_event_record(self.h, stream_ptr), "rtp_event_record")

    def wait(self, stream_ptr):
        check(self.lib.rtp_stream_wait_event(stream_ptr, self.h), "rtp_stream_wait_event")

    def __del__(self):
        try:
            if self.h:
                self.lib.rtp_event_destroy(self.h)
                self.h = None
        except Exception:
            pass


class HipBackend:
    name = "hip"

    def __init__(self, device=None):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.RtpError("rt_pose_amd needs an MI355X (no GPU visible) -- there is no CPU fallback")
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self.bytes = 0

    # -------------------------------------------------------------- memory (plumbing)
    def alloc(self, shape, dtype):
        t = torch.zeros(shape, dtype=_DT[dtype], device=self.device)
        self.bytes += t.numel() * t.element_size()
        return t

    def stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # one HIP stream per lane of the launch plan (rt_pose_amd/lanes.py); lane 0 is the caller's current stream
    def lane_streams(self, n):
        cur = torch.cuda.current_stream(self.device)
        if len(getattr(self, "_lanes", ())) < n - 1:
            # HIP deals streams onto GPU_MAX_HW_QUEUES (4) hardware queues in creation order, and which lanes share a queue decides
            # how their kernels interleave with the main lane's persistent ones (measured: 4 queues 6.1 ms/step, 3: 6.4, 5+: 9.0;
            # side lanes at default priority, the step stream high: the other combinations measured 3-8 % slower)
            self._lanes = [torch.cuda.Stream(self.device) for _ in range(1, n)]
        streams = [cur] + self._lanes[:n - 1]
        return streams, [C.c_void_p(st.cuda_stream) for st in streams]

    def new_event(self, system_fence=False):
        """An ordering event of the lane plan.  Events BETWEEN launches of a plan are device-scope (csrc/lane_events.hip: no
        system-scope cache writeback / invalidate when recorded); the few events that JOIN the side lanes back into the caller's
        stream at the end of a plan (system_fence=True) are ordinary HIP events: what the side lanes wrote is then visible to
        whatever the caller queues next -- a D2H copy of the losses, a collective's peer reads -- by the HIP contract itself."""
        return _LaneEvent(self.lib, bool(system_fence))

    def stem_bwd_blocks(self):
        return self.lib.rtp_stem_bwd_blocks()

    # -------------------------------------------------------------- conv family
    def chan_stats(self, a, b, nsplit, out):
        fn, aa, bb, o = self.lib.rtp_chan_stats, _act(a), _act(b), _ptr(out)
        n, vox = a.n, a.vox
        keep = (a, b, out)
        return lambda s: check(fn(aa, bb, n, vox, nsplit, o, s), "rtp_chan_stats") or keep and None

    def fold_fwd(self, w, bias, gamma, beta, stats, nsplit, groups, eps, geom, ci_real, co_real, wf, btab, mr, wd=None):
        fn, g = self.lib.rtp_fold_fwd, _geom(geom)
        args = (_ptr(w), _ptr(bias), _ptr(gamma), _ptr(beta), _ptr(stats), nsplit, groups, eps, g, ci_real, co_real,
                _ptr(wf), _ptr(btab), _ptr(mr), _ptr(wd))
        keep = (w, bias, gamma, beta, stats, wf, btab, mr, wd)
        return lambda s: check(fn(*args, s), "rtp_fold_fwd") or keep and None

    def pack_dgrad_w(self, w, geom, ci_real, co_real, wd):
        fn, g = self.lib.rtp_pack_dgrad_w, _geom(geom)
        args = (_ptr(w), g, ci_real, co_real, _ptr(wd))
        keep = (w, wd)
        return lambda s: check(fn(*args, s), "rtp_pack_dgrad_w") or keep and None

    def multi(self, fns):
        """fns: launch closures of this backend that each issue ONE stride-1 LDS-tiled kernel launch of the same variant on the same
        samples (conv_gn_fused / conv / conv_dgrad_fused, or wgrad / wgrad_q / wgrad_tg).  -> ONE closure that issues them as a
        shared launch (include/rtp.h: rtp_multi_*), or None when they cannot share one (the callers keep the separate launches).
        The handle and its device parameter block live as long as the returned closure (released by _MultiLaunch.__del__)."""
        lib = self.lib
        check(lib.rtp_multi_begin(), "rtp_multi_begin")
        try:
            for f in fns:
                f(None)          # recorded by the entry points, not launched
        except BaseException as e:   # whatever a closure raises: the capture must not stay open on this thread
            lib.rtp_multi_abort()
            if isinstance(e, _lib.RtpError):
                return None
            raise
        params = torch.zeros(int(lib.rtp_multi_param_bytes()), dtype=torch.uint8, device=self.device)
        h = C.c_int(-1)
        if lib.rtp_multi_end(_ptr(params), params.numel(), C.byref(h)) != 0 or h.value < 0:
            return None
        return _MultiLaunch(lib, h.value, params, tuple(fns))

    def conv_tiled_ok(self, x, geom, transposed):
        return bool(self.lib.rtp_conv_tiled_ok(_act(x), _geom(geom), int(transposed)))

    def conv_sliced_ok(self, x, geom, transposed):
        """A wide 3x3x3 stride-1 conv (Cin = 32 K, Cout = 32 J) that rtp_conv_igemm_ws runs as channel slices of the LDS-tiled kernel."""
        return bool(self.lib.rtp_conv_sliced_ok(_act(x), _geom(geom), int(transposed)))

    def conv(self, x, wf, per_sample, btab, res, y, geom, relu, transposed, y_fp32, stats=None, acc=None, ws=None):
        """stats = (stat_x | None, out [n, S, cout, 2]) with S = conv_stats_nsplit(...) > 0: the conv also emits the
        per-channel statistics of y (rtp_conv_igemm_stats).
        acc = (fp32 [n, vox, acc_cs], acc_cs): partial result of earlier input-channel slices (rtp_conv_igemm_acc).
        ws = fp32 [n * output voxels * 32] scratch: conv_sliced_ok geometries run as channel slices (rtp_conv_igemm_ws)."""
        g = _geom(geom)
        if ws is not None:
            assert acc is None
            fn = self.lib.rtp_conv_igemm_ws
            args = (_act(x), _ptr(wf), int(per_sample), _ptr(btab), _act(res), _act(y), g, int(relu), int(transposed),
                    int(y_fp32), _act(stats[0]) if stats is not None else None, _ptr(stats[1]) if stats is not None else None,
                    _ptr(ws))
            keep = (x, wf, btab, res, y, stats, ws)
            return lambda s: check(fn(*args, s), "rtp_conv_igemm_ws") or keep and None
        if acc is not None:
            assert stats is None
            fn = self.lib.rtp_conv_igemm_acc
            args = (_act(x), _ptr(wf), int(per_sample), _ptr(btab), _act(res), _act(y), g, int(relu), int(transposed),
                    int(y_fp32), _ptr(acc[0]), int(acc[1]))
            keep = (x, wf, btab, res, y, acc)
            return lambda s: check(fn(*args, s), "rtp_conv_igemm_acc") or keep and None
        args = (_act(x), _ptr(wf), int(per_sample), _ptr(btab), _act(res), _act(y), g, int(relu), int(transposed),
                int(y_fp32))
        keep = (x, wf, btab, res, y, stats)
        if stats is not None:
            fn = self.lib.rtp_conv_igemm_stats
            args = args + (_act(stats[0]), _ptr(stats[1]))
            return lambda s: check(fn(*args, s), "rtp_conv_igemm_stats") or keep and None
        fn = self.lib.rtp_conv_igemm
        return lambda s: check(fn(*args, s), "rtp_conv_igemm") or keep and None

    def conv_gn_fused(self, x, wt, bias, gamma, beta, stats, nsplit, groups, eps, co_real, mr, res, y, geom, relu, stat_out=None):
        """rtp_conv_gn_fused: GroupNorm fold in the conv kernel's prologue (no fold launch); the LDS-tiled geometries only.
        wt: the weights in tap-major fp32 order [27][co_pad][32] (tail item "pack_wt")."""
        fn, g = self.lib.rtp_conv_gn_fused, _geom(geom)
        fs = _lib.RtpGnFold(_ptr(wt), _ptr(bias), _ptr(gamma), _ptr(beta), _ptr(stats), int(nsplit), int(groups), int(co_real),
                            float(eps), _ptr(mr))
        args = (_act(x), C.byref(fs), _act(res), _act(y), g, int(relu), _ptr(stat_out))
        keep = (x, wt, bias, gamma, beta, stats, mr, res, y, stat_out, fs)
        return lambda s: check(fn(*args, s), "rtp_conv_gn_fused") or keep and None

    def conv64_blocks(self, xs, wblocks, w_row_stride, w_tap_stride, btabs, bt_cs, ress, ys, geom, relu, transposed,
                      acc=None, acc_in=False, acc_out=False):
        """rtp_conv64_blocks (include/rtp.h): a 64-wide stride-1 3x3x3 conv whose operands are given per 32-channel half.
        xs / ys / ress: pairs of Views (the 32 channels at .co; ys None with acc_out, ress None: no residual);
        wblocks[h][k] = (bf16 tensor, element offset) of the weight block [27][32 rows of output half h][32 columns of input half k];
        btabs = pair of (fp32 tensor, element offset) or None; acc: fp32 [n, voxels, 64] partial sums of a chain over slices."""
        c = _lib.RtpConv64()
        keep = [xs, wblocks, btabs, ress, ys, acc]
        for h in range(2):
            c.x[h] = xs[h].buf.data_ptr() + 2 * xs[h].co
            c.x_cs[h] = xs[h].cs
            for k in range(2):
                t, off = wblocks[h][k]
                c.w[h][k] = t.data_ptr() + 2 * off
            if btabs is not None:
                t, off = btabs[h]
                c.btab[h] = t.data_ptr() + 4 * off
            if ress is not None:
                c.res[h] = ress[h].buf.data_ptr() + 2 * ress[h].co
                c.r_cs[h] = ress[h].cs
            if ys is not None:
                c.y[h] = ys[h].buf.data_ptr() + 2 * ys[h].co
                c.y_cs[h] = ys[h].cs
        c.w_row_stride, c.w_tap_stride, c.w_sample_stride, c.w_per_sample = int(w_row_stride), int(w_tap_stride), 0, 0
        c.bt_cs = int(bt_cs)
        c.acc = acc.data_ptr() if acc is not None else None
        c.acc_in, c.acc_out, c.relu, c.transposed = int(acc_in), int(acc_out), int(relu), int(transposed)
        fn, g = self.lib.rtp_conv64_blocks, _geom(geom)
        return lambda s: check(fn(C.byref(c), g, s), "rtp_conv64_blocks") or keep and None

    def conv_stats_nsplit(self, x, geom, transposed, ws=False):
        """ws: the conv will be launched with a slice workspace (conv(..., ws=...))."""
        fn = self.lib.rtp_conv_stats_nsplit_ws if ws else self.lib.rtp_conv_stats_nsplit
        return fn(_act(x), _geom(geom), int(transposed))

    def wgrad(self, gy, x, geom, nsplit, gp):
        fn, g = self.lib.rtp_wgrad, _geom(geom)
        args = (_act(gy), _act(x), g, nsplit, _ptr(gp))
        keep = (gy, x, gp)
        return lambda s: check(fn(*args, s), "rtp_wgrad") or keep and None

    def wgrad_tg(self, gy, x, geom, nsplit, gp, tg):
        """rtp_wgrad on the tiled kernel + the subset sums of gy (tg [n, nsplit, 27, 32]): bias gradients without a class-sum pass."""
        fn, g = self.lib.rtp_wgrad_tg, _geom(geom)
        args = (_act(gy), _act(x), g, nsplit, _ptr(gp), _ptr(tg))
        keep = (gy, x, gp, tg)
        return lambda s: check(fn(*args, s), "rtp_wgrad_tg") or keep and None

    def wgrad_nsplit(self, geom):
        return self.lib.rtp_wgrad_nsplit(_geom(geom))

    def wgrad_q(self, gy, x, geom, nsplit, gp, wd, qpart, tg=None):
        """rtp_wgrad_q: the tiled weight-gradient correlation + each slab's contraction with the data-gradient weights
        (+ tg [n][27][32]: subset sums of gy, accumulated with atomics into a buffer the caller zeroes)."""
        fn, g = self.lib.rtp_wgrad_q, _geom(geom)
        args = (_act(gy), _act(x), g, nsplit, _ptr(gp), _ptr(wd), _ptr(qpart), _ptr(tg))
        keep = (gy, x, gp, wd, qpart, tg)
        return lambda s: check(fn(*args, s), "rtp_wgrad_q") or keep and None

    def qpart_from_slabs(self, gp, n, nsplit, ntap, co32, ci, wd, qpart):
        fn, args = self.lib.rtp_qpart_from_slabs, (_ptr(gp), n, nsplit, ntap, co32, ci, _ptr(wd), _ptr(qpart))
        keep = (gp, wd, qpart)
        return lambda s: check(fn(*args, s), "rtp_qpart_from_slabs") or keep and None

    def conv_dgrad_fused_ok(self, gy, geom):
        return bool(self.lib.rtp_conv_dgrad_fused_ok(_act(gy), _geom(geom)))

    def zero_f32(self, t):
        fn, args = self.lib.rtp_zero_f32, (_ptr(t), t.numel())
        return lambda s: check(fn(*args, s), "rtp_zero_f32") or t is None

    def gn_bwd_coeffs_cls(self, qpart, q_nsplit, cls_part, cls_nsplit, csum_out, wd, mr, gamma, geom, ci_real, co_real, groups,
                          coeff):
        fn, g = self.lib.rtp_gn_bwd_coeffs_cls, _geom(geom)
        args = (_ptr(qpart), q_nsplit, _ptr(cls_part), cls_nsplit, _ptr(csum_out), _ptr(wd), _ptr(mr), _ptr(gamma), g, ci_real,
                co_real, groups, _ptr(coeff))
        keep = (qpart, cls_part, csum_out, wd, mr, gamma, coeff)
        return lambda s: check(fn(*args, s), "rtp_gn_bwd_coeffs_cls") or keep and None

    def gn_bwd_p(self, cls_part, cls_nsplit, csum_out, wd, geom, ci_real, co_real, p_out):
        fn, g = self.lib.rtp_gn_bwd_p, _geom(geom)
        args = (_ptr(cls_part), cls_nsplit, _ptr(csum_out), _ptr(wd), g, ci_real, co_real, _ptr(p_out))
        keep = (cls_part, csum_out, wd, p_out)
        return lambda s: check(fn(*args, s), "rtp_gn_bwd_p") or keep and None

    def conv_dgrad_fused(self, gy, wd, x, coeff, terms, mask, dx, geom, tot_out=None, gn=None):
        """rtp_conv_dgrad_fused: terms = [(View, coeff | None)] contributions of x's other consumers (<= 3);
        tot_out fp32 [n][conv_stats_nsplit(gy, geom, True)][32]: per-channel totals of the stored dx;
        gn = dict(qpart, q_nsplit, p, mr, gamma, groups, coeff_out): coefficients computed in the kernel (coeff is None)."""
        fn, g = self.lib.rtp_conv_dgrad_fused, _geom(geom)
        arr = self._terms(terms, False) if terms else None
        gs = None
        if gn is not None:
            dp = lambda k: gn[k].data_ptr() if gn.get(k) is not None else None
            gs = _lib.RtpGnBwd(dp("qpart"), gn["q_nsplit"], dp("p"), dp("tg"), dp("csum_out"), dp("csum"), dp("mr"), dp("gamma"),
                               gn["groups"], dp("coeff_out"))
        args = (_act(gy), _ptr(wd), _act(x), _ptr(coeff), C.byref(gs) if gs is not None else None, arr, len(terms), int(mask),
                _act(dx), g, _ptr(tot_out))
        keep = (gy, wd, x, coeff, terms, dx, arr, tot_out, gn, gs)
        return lambda s: check(fn(*args, s), "rtp_conv_dgrad_fused") or keep and None

    def class_sums_p(self, gy, nsplit, scratch, tot_part, tot_nsplit, csum_out, wd, geom, ci_real, co_real, p_out):
        """rtp_class_sums_p: class sums (boundary-only when tot_part is given) + P of the GroupNorm backward, one launch."""
        fn = self.lib.rtp_class_sums_p
        counters = self.alloc((gy.n,), "i32")
        args = (_act(gy), gy.n, gy.d, gy.h, gy.w, nsplit, _ptr(scratch), _ptr(tot_part), tot_nsplit, _ptr(csum_out), _ptr(wd),
                _geom(geom) if geom is not None else None, ci_real, co_real, _ptr(p_out), _ptr(counters))
        keep = (gy, scratch, tot_part, csum_out, wd, p_out, counters)
        return lambda s: check(fn(*args, s), "rtp_class_sums_p") or keep and None

    def class_sums_boundary(self, gy, nsplit, scratch, tot_part, tot_nsplit, out):
        fn = self.lib.rtp_class_sums_boundary
        args = (_act(gy), gy.n, gy.d, gy.h, gy.w, nsplit, _ptr(scratch), _ptr(tot_part), tot_nsplit, _ptr(out))
        keep = (gy, scratch, tot_part, out)
        return lambda s: check(fn(*args, s), "rtp_class_sums_boundary") or keep and None

    def class_sums(self, gy, nsplit, scratch, out):
        fn = self.lib.rtp_class_sums
        args = (_act(gy), gy.n, gy.d, gy.h, gy.w, nsplit, _ptr(scratch), _ptr(out))
        keep = (gy, scratch, out)
        return lambda s: check(fn(*args, s), "rtp_class_sums") or keep and None

    def class_sums_reduce(self, scratch, nsplit, n, c, out):
        fn = self.lib.rtp_class_sums_reduce
        args = (_ptr(scratch), nsplit, n, c, _ptr(out))
        keep = (scratch, out)
        return lambda s: check(fn(*args, s), "rtp_class_sums_reduce") or keep and None

    def wgrad_fold(self, gp, nsplit, csum, mr, gamma, beta, groups, geom, ci_real, co_real, dw, dbias, acc):
        fn, g = self.lib.rtp_wgrad_fold, _geom(geom)
        args = (_ptr(gp), nsplit, _ptr(csum), _ptr(mr), _ptr(gamma), _ptr(beta), groups, g, ci_real, co_real,
                _ptr(dw), _ptr(dbias), int(acc))
        keep = (gp, csum, mr, gamma, beta, dw, dbias)
        return lambda s: check(fn(*args, s), "rtp_wgrad_fold") or keep and None

    def gn_bwd_coeffs(self, pq, nsplit, mr, gamma, n, c, groups, vox, coeff, dgamma, dbeta, acc):
        fn = self.lib.rtp_gn_bwd_coeffs
        args = (_ptr(pq), nsplit, _ptr(mr), _ptr(gamma), n, c, groups, vox, _ptr(coeff), _ptr(dgamma), _ptr(dbeta),
                int(acc))
        keep = (pq, mr, gamma, coeff, dgamma, dbeta)
        return lambda s: check(fn(*args, s), "rtp_gn_bwd_coeffs") or keep and None

    def tail(self, items):
        """One launch for a list of independent deferred items (rtp_tail_*): tuples ("class_reduce", scratch, nsplit,
        n, c, out) | ("wgrad_fold", <wgrad_fold args>) | ("gn_param", coeff, n, c, dgamma, dbeta, acc) |
        ("fold_fwd", <fold_fwd args>) | ("pack_wt", w, co_real, co_pad, ci, ntap, wt)."""
        lib = self.lib
        nb = lib.rtp_tail_desc_bytes()
        host = C.create_string_buffer(nb * len(items))
        starts, shm, keep = [0], 0, []
        for i, it in enumerate(items):
            d = C.c_void_p(C.addressof(host) + i * nb)
            blocks, sb = C.c_int(0), C.c_int(0)
            kind, a = it[0], it[1:]
            if kind == "class_reduce":
                rc = lib.rtp_tail_desc_class_reduce(_ptr(a[0]), a[1], a[2], a[3], _ptr(a[4]), d, C.byref(blocks), C.byref(sb))
            elif kind == "wgrad_fold" and len(a) == 14 and a[13] is not None:   # bias from the weight-gradient kernel's subset sums
                gp, nsplit, csum, mr, gamma, beta, groups, geom, ci_real, co_real, dw, dbias, acc, tg = a
                assert csum is None and mr is None
                rc = lib.rtp_tail_desc_wgrad_fold_tg(_ptr(gp), nsplit, _ptr(tg), _geom(geom), ci_real, co_real, _ptr(dw), _ptr(dbias),
                                                     int(acc), d, C.byref(blocks), C.byref(sb))
            elif kind == "wgrad_fold":
                gp, nsplit, csum, mr, gamma, beta, groups, geom, ci_real, co_real, dw, dbias, acc = a[:13]
                rc = lib.rtp_tail_desc_wgrad_fold(_ptr(gp), nsplit, _ptr(csum), _ptr(mr), _ptr(gamma), _ptr(beta), groups,
                                                  _geom(geom), ci_real, co_real, _ptr(dw), _ptr(dbias), int(acc), d,
                                                  C.byref(blocks), C.byref(sb))
            elif kind == "fold_fwd":
                w, bias, gamma, beta, stats, nsplit, groups, eps, geom, ci_real, co_real, wf, btab, mr, wd = a
                rc = lib.rtp_tail_desc_fold_fwd(_ptr(w), _ptr(bias), _ptr(gamma), _ptr(beta), _ptr(stats), nsplit, groups,
                                                eps, _geom(geom), ci_real, co_real, _ptr(wf), _ptr(btab), _ptr(mr),
                                                _ptr(wd), d, C.byref(blocks), C.byref(sb))
            elif kind == "pack_wt":
                w, co_real, co_pad, ci, ntap, wt = a
                rc = lib.rtp_tail_desc_pack_wt(_ptr(w), co_real, co_pad, ci, ntap, _ptr(wt), d, C.byref(blocks), C.byref(sb))
            elif kind == "gn_param":
                rc = lib.rtp_tail_desc_gn_param(_ptr(a[0]), a[1], a[2], _ptr(a[3]), _ptr(a[4]), int(a[5]), d,
                                                C.byref(blocks), C.byref(sb))
            else:
                raise ValueError(kind)
            check(rc, "rtp_tail_desc_" + kind)
            starts.append(starts[-1] + blocks.value)
            shm = max(shm, sb.value)
            keep.append(a)
        descs = torch.frombuffer(bytearray(host.raw), dtype=torch.uint8).to(self.device)
        st = torch.tensor(starts, dtype=torch.int32, device=self.device)
        fn, args = lib.rtp_tail_launch, (_ptr(descs), _ptr(st), len(items), starts[-1], shm)
        keep = (keep, descs, st)
        return lambda s: check(fn(*args, s), "rtp_tail_launch") or keep and None

    # -------------------------------------------------------------- point-wise family
    @staticmethod
    def _terms(terms, with_dims):
        arr = (RtpTerm * len(terms))()
        for i, t in enumerate(terms):
            v, coeff = t if isinstance(t, tuple) else (t, None)
            coeff = getattr(coeff, "tensor", coeff)   # graph.LazyCoeff: the buffer the coefficients live in
            arr[i].t = _act(v)
            arr[i].coeff = coeff.data_ptr() if coeff is not None else None
            arr[i].d, arr[i].h, arr[i].w = (v.d, v.h, v.w) if with_dims else (0, 0, 0)
        return arr

    def grad_combine(self, terms, x, relu_src, out, cls=None):
        """cls = (nsplit, scratch): also emit per-boundary-class partial sums of the result (rtp_grad_combine_cls)."""
        arr = self._terms(terms, False)
        lazies = [cf for _, cf in terms if hasattr(cf, "pq")]   # graph.LazyCoeff terms still to be computed
        if cls is not None:
            lz = None
            if lazies:   # ... by this launch's prologue
                lz = (_lib.RtpGnLazy * len(terms))()
                for i, (_, cf) in enumerate(terms):
                    if hasattr(cf, "pq"):
                        lz[i].pq, lz[i].nsplit, lz[i].mr, lz[i].gamma, lz[i].groups = (cf.pq.data_ptr(), cf.nsplit, cf.mr.data_ptr(),
                                                                                      cf.gamma.data_ptr(), cf.groups)
            fn = self.lib.rtp_grad_combine_cls_lazy
            args = (arr, len(terms), lz, _act(x), _act(relu_src), _act(out), out.n, out.d, out.h, out.w, cls[0], _ptr(cls[1]))
            keep = (terms, x, relu_src, out, cls, lz)
            return lambda s: check(fn(*args, s), "rtp_grad_combine_cls_lazy") or keep and None
        assert not lazies, "lazy GroupNorm coefficients need the class-sum combine (graph.finalize_grad materialises them otherwise)"
        fn = self.lib.rtp_grad_combine
        args = (arr, len(terms), _act(x), _act(relu_src), _act(out), out.n, out.vox)
        keep = (terms, x, relu_src, out)
        return lambda s: check(fn(*args, s), "rtp_grad_combine") or keep and None

    grad_combine_lazy_ok = True   # rtp_grad_combine_cls_lazy: GroupNorm-backward coefficients in the combine's prologue

    @staticmethod
    def grad_combine_cls_ok(c):
        """Channel counts rtp_grad_combine_cls accepts."""
        return c <= 64 and c % 8 == 0 and 64 % (c // 8) == 0

    def fuse_stats_nsplit(self, out):
        """Statistics partials per sample rtp_fuse_sum_stats writes for this row (0: not offered)."""
        return int(self.lib.rtp_fuse_stats_nsplit(out.n, out.c, out.d, out.h, out.w))

    def fuse_sum(self, terms, bias, out, relu, stats=None):
        """stats = (nsplit, fp32 [n, nsplit, c, 2]): the row also emits the statistics of what it stores (rtp_fuse_sum_stats)."""
        fn, arr = self.lib.rtp_fuse_sum_stats, self._terms(terms, True)
        args = (arr, len(terms), _ptr(bias), _act(out), out.n, out.d, out.h, out.w, int(relu),
                _ptr(stats[1]) if stats else None, stats[0] if stats else 0)
        keep = (terms, bias, out, stats)
        return lambda s: check(fn(*args, s), "rtp_fuse_sum_stats") or keep and None

    def upsample_bwd(self, ghi, glow):
        fn = self.lib.rtp_upsample_bwd
        scratch = self.alloc((self.lib.rtp_upsample_bwd_scratch_floats(ghi.n, ghi.c, ghi.d, ghi.h, ghi.w, glow.d, glow.h, glow.w),), "f32")
        args = (_act(ghi), ghi.d, ghi.h, ghi.w, _act(glow), glow.d, glow.h, glow.w, ghi.n, _ptr(scratch))
        keep = (ghi, glow, scratch)
        return lambda s: check(fn(*args, s), "rtp_upsample_bwd") or keep and None

    def stem_fwd(self, x, w, b, y):
        fn = self.lib.rtp_stem_fwd
        args = (_ptr(x), _ptr(w), _ptr(b), _act(y), y.n, y.vox)
        keep = (x, w, b, y)
        return lambda s: check(fn(*args, s), "rtp_stem_fwd") or keep and None

    def stem_stats_nsplit(self, n, c, vox):
        return self.lib.rtp_stem_stats_nsplit(n, c, vox)

    def stem_fwd_stats(self, x, w, b, y, stats, nsplit):
        """rtp_stem_fwd_stats: the stem with the statistics of its (stored) output as an epilogue, [n, nsplit, c, 2]."""
        fn = self.lib.rtp_stem_fwd_stats
        args = (_ptr(x), _ptr(w), _ptr(b), _act(y), y.n, y.vox, _ptr(stats), nsplit)
        keep = (x, w, b, y, stats)
        return lambda s: check(fn(*args, s), "rtp_stem_fwd_stats") or keep and None

    def stem_bwd(self, x, gy, scratch, dw, db, acc):
        fn = self.lib.rtp_stem_bwd
        args = (_ptr(x), _act(gy), gy.n, gy.vox, _ptr(scratch), _ptr(dw), _ptr(db), int(acc))
        keep = (x, gy, scratch, dw, db)
        return lambda s: check(fn(*args, s), "rtp_stem_bwd") or keep and None

    def pack_ncdhw(self, x, y, c):
        fn = self.lib.rtp_pack_ncdhw
        args = (_ptr(x), _act(y), y.n, c, y.vox)
        keep = (x, y)
        return lambda s: check(fn(*args, s), "rtp_pack_ncdhw") or keep and None

    def unpack_ncdhw(self, x, y, c):
        fn = self.lib.rtp_unpack_ncdhw
        args = (_act(x), _ptr(y), x.n, c, x.vox)
        keep = (x, y)
        return lambda s: check(fn(*args, s), "rtp_unpack_ncdhw") or keep and None

    # -------------------------------------------------------------- deformable feature adaption (BASELINE config 4)
    def dcn_cl_forward(self, x, off_act, w_ad, y, relu=True):
        """rtp_dcn_cl_forward: y = [relu] DeformConv3x3_dg4(x, off) per (frame, z) slice on the plan's layout (x, y bf16 channels-last
        Views, off_act an fp32 channels-last View with >= 72 channels, w_ad fp32 [32, 32, 3, 3])."""
        fn = self.lib.rtp_dcn_cl_forward
        oa = RtpAct(off_act.buf.data_ptr(), off_act.cs, off_act.co, off_act.c)
        args = (_act(x), oa, _ptr(w_ad), _act(y), x.n * x.d, x.h, x.w, int(relu))
        keep = (x, off_act, w_ad, y)
        return lambda s: check(fn(*args, s), "rtp_dcn_cl_forward") or keep and None

    def dcn_adapt(self, x, off_act, koff, w_ad, y, dg=4):
        """The deformable half of FeatureAdaption (center_head.py:24-62) on a 5-D feature with Z folded into the batch:
        y = relu(DeformConv3x3(x2, offset)) for x2 = [N*D, C, H, W].  The offsets are an activation of the plan (the 1x1
        offset conv runs on the plan's own MFMA conv with fp32 output, channels-last [.., >= koff]); the bf16 channels-last
        tensors are handed to the fp32 NCHW operator of include/rtp.h section D and back (rtp_unpack_ncdhw[_f32],
        rtp_pack_ncdhw_ex).  Returns (forward closure, make_backward(gy, gx, goff, gw_ad))."""
        lib = self.lib
        N, C, H, W = x.n * x.d, x.c, x.h, x.w
        vox = H * W
        f32 = lambda *shape: self.alloc(shape, "f32")
        xf, off, yf = f32(N, C, H, W), f32(N, koff, H, W), f32(N, C, H, W)
        step = 64
        while N % step:
            step //= 2
        ws3 = f32(max(1, lib.rtp_dcn_workspace_bytes(step, C, H, W, C, 3, 3, H, W) // 4))
        xa, ya = _act(x), _act(y)
        keep = [x, y, off_act, w_ad, xf, off, yf, ws3]
        # forward on the plan's own layout (rtp_dcn_cl_forward: four 16-byte corner loads per sample instead of sixteen planar ones,
        # no unpack / transpose / pack passes); the backward operator still wants fp32 planes and unpacks them itself then.
        # PlanOptions.dcn_cl = 0: the fp32 NCHW operator for the forward too
        from .options import PlanOptions
        cl = (PlanOptions.from_env().dcn_cl and C == 32 and dg == 4 and koff == 72 and vox % 16 == 0
              and x.cs % 8 == 0 and x.co % 8 == 0)
        oa = RtpAct(off_act.buf.data_ptr(), off_act.cs, off_act.co, off_act.c)

        state = {"prepared": False}   # True: the plan runs `prep` as a launch of its own (beside the forward) before the backward

        def unpack_inputs(s):
            check(lib.rtp_unpack_ncdhw(xa, _ptr(xf), N, C, vox, s), "rtp_unpack_ncdhw")
            check(lib.rtp_unpack_ncdhw_f32(_ptr(off_act.buf), off_act.cs, off_act.co, _ptr(off), N, koff, vox, s), "rtp_unpack_ncdhw_f32")

        def fwd(s):
            if cl:
                check(lib.rtp_dcn_cl_forward(xa, oa, _ptr(w_ad), ya, N, H, W, 1, s), "rtp_dcn_cl_forward")
                return keep and None
            unpack_inputs(s)
            check(lib.rtp_deform_conv_forward(_ptr(xf), _ptr(w_ad), _ptr(off), _ptr(yf), _ptr(ws3), N, C, H, W, C, 3, 3, 1, 1, 1, 1,
                                              1, 1, 1, dg, step, s), "rtp_deform_conv_forward")
            check(lib.rtp_pack_ncdhw_ex(_ptr(yf), None, ya, N, C, vox, 1, s), "rtp_pack_ncdhw_ex")
            return keep and None

        def make_backward(gy, gx, goff_v, gw_ad):
            gyf, gi, goff = f32(N, C, H, W), f32(N, C, H, W), f32(N, koff, H, W)
            gya, gxa, goa = _act(gy), _act(gx), _act(goff_v)
            zero = [gi, goff, gw_ad]
            keepb = [gy, gx, goff_v, gyf] + zero

            def bwd(s):
                # the plan owns the three gradient buffers: the overwriting entry (no zero fills; the reference's wrapper
                # allocates zeros and accumulates, deform_conv.py:75-76, 86)
                if cl and not state["prepared"]:
                    unpack_inputs(s)
                check(lib.rtp_unpack_ncdhw(gya, _ptr(gyf), N, C, vox, s), "rtp_unpack_ncdhw")
                check(lib.rtp_deform_conv_backward_overwrite(_ptr(xf), _ptr(off), _ptr(gyf), _ptr(gi), _ptr(goff), _ptr(w_ad),
                                                             _ptr(gw_ad), _ptr(ws3), N, C, H, W, C, 3, 3, 1, 1, 1, 1, 1, 1, 1, dg,
                                                             1.0, step, s),
                      "rtp_deform_conv_backward_overwrite")
                check(lib.rtp_pack_ncdhw_ex(_ptr(gi), None, gxa, N, C, vox, 0, s), "rtp_pack_ncdhw_ex")
                check(lib.rtp_pack_ncdhw_ex(_ptr(goff), None, goa, N, koff, vox, 0, s), "rtp_pack_ncdhw_ex")
                return keepb and None
            return bwd

        if cl:
            def prep(s):
                unpack_inputs(s)
                return keep and None

            def use_prep():
                state["prepared"] = True
                return prep
            make_backward.prep = use_prep   # the backward operator's fp32 planes as a launch the plan places where it likes
        make_backward.planes = (xf, off)    # what `prep` writes and the backward operator reads: the plan tracks them as buffers
        return fwd, make_backward

    # -------------------------------------------------------------- head: loss / decode / optimiser
    def focal_scratch(self, n):
        return self.alloc((n * self.lib.rtp_focal_blocks() * 2,), "f32")

    def focal_loss(self, logits, target, ind, mask, cat, ncls, gscale, scratch, out_loss, ghm, write_pad=True):
        """write_pad=False: the gradient rows' padding channels are not stored (ghm zeroed once, written by nothing else)."""
        fn = self.lib.rtp_focal_loss_ex
        n, vox, m = logits.n, logits.vox, ind.shape[1]
        args = (_ptr(logits.buf), logits.cs, _ptr(target), _ptr(ind), _ptr(mask), _ptr(cat), n, ncls, vox, m,
                float(gscale), _ptr(scratch), _ptr(out_loss), _act(ghm), int(write_pad))
        keep = (logits, target, ind, mask, cat, scratch, out_loss, ghm)
        return lambda s: check(fn(*args, s), "rtp_focal_loss_ex") or keep and None

    def reg_loss(self, reg, target, ind, mask, code_w, nreg, gscale, out, greg, prev=None):
        """prev (int64 [n, m], initialised to -1; greg zero-initialised and written by nothing else): rtp_reg_loss_sparse -- only
        the previous call's voxels are cleared instead of zero-filling the gradient tensor."""
        n, vox, m = reg.n, reg.vox, ind.shape[1]
        args = (_ptr(reg.buf), reg.cs, _ptr(target), _ptr(ind), _ptr(mask), _ptr(code_w), n, nreg, vox, m,
                float(gscale), _ptr(out), _act(greg))
        keep = (reg, target, ind, mask, code_w, out, greg, prev)
        if prev is not None:
            fn, args = self.lib.rtp_reg_loss_sparse, args + (_ptr(prev),)
            return lambda s: check(fn(*args, s), "rtp_reg_loss_sparse") or keep and None
        fn = self.lib.rtp_reg_loss
        return lambda s: check(fn(*args, s), "rtp_reg_loss") or keep and None

    def decode_scratch(self, n, ncls):
        return self.alloc((self.lib.rtp_decode_scratch_floats(n, ncls),), "f32")

    def decode(self, hm, reg, ncls, nreg, scale_xyz, origin_xyz, scratch, out):
        fn = self.lib.rtp_decode
        sc = (C.c_float * 3)(*scale_xyz)
        og = (C.c_float * 3)(*origin_xyz)
        args = (_ptr(hm.buf), hm.cs, _ptr(reg.buf), reg.cs, hm.n, ncls, nreg, hm.d, hm.h, hm.w, sc, og, _ptr(scratch),
                _ptr(out))
        keep = (hm, reg, scratch, out, sc, og)
        return lambda s: check(fn(*args, s), "rtp_decode") or keep and None

    def sqnorm_blocks(self):
        return self.lib.rtp_sqnorm_blocks()

    def sqnorm(self, g, n, hyper, partial):
        fn = self.lib.rtp_sqnorm
        args = (_ptr(g), n, _ptr(hyper), _ptr(partial))
        keep = (g, hyper, partial)
        return lambda s: check(fn(*args, s), "rtp_sqnorm") or keep and None

    def adam_step(self, p, g, m, v, n, hyper, partial, mode, norm_out):
        fn = self.lib.rtp_adam_step
        args = (_ptr(p), _ptr(g), _ptr(m), _ptr(v), n, _ptr(hyper), _ptr(partial), mode, _ptr(norm_out))
        keep = (p, g, m, v, hyper, partial, norm_out)
        return lambda s: check(fn(*args, s), "rtp_adam_step") or keep and None

    # -------------------------------------------------------------- diagnostics
    def prof_enable(self, family, on=True):
        check(self.lib.rtp_prof_enable(family, int(on)), "rtp_prof_enable")

    def prof_collect(self, family):
        ms, cnt = C.c_float(0), C.c_int(0)
        check(self.lib.rtp_prof_collect(family, C.byref(ms), C.byref(cnt)), "rtp_prof_collect")
        return ms.value, cnt.value
