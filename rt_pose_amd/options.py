"""Build-time choices of the launch plan, in ONE place.

Rounds 1-5 grew one environment variable per experiment (72 RTP_* switches, most of them A/B losers); round 6 removed the losers
and folded what is left -- alternative routes that the tests still exercise, and the few levers a deployment may want -- into this
object.  A plan is built with `PlanOptions()` (the measured best) unless the caller passes its own; the ONE environment variable
`RTP_PLAN="key=value;key=value"` overrides fields for experiments (tools/, A/B scripts, the plan-logic tests), read when a plan is
built, never at import.

    RTP_PLAN="defer_wg=;lazy_coef=0"  python bench.py --steps 20

Field                default   meaning (alternatives are kept because a test compares them with the default route)
-------------------  --------  -------------------------------------------------------------------------------------------------
width_hints          None      "tag-prefix=workgroups;..." for the persistent tiled launches (None: engine.DEFAULT_WIDTH_HINTS; "": none)
lanes                None      lane -> stream map "0,1,2,3,4,5" (None: the four-stream map, six streams for channel-sliced heads)
graph_lanes          0,1,1,2,2,1   lane -> stream fold of the captured HIP graph (four-stream capture crashes in ROCm 7.2)
ar_buckets           1         gradient all-reduce buckets (2: the transition2 .. pose_head suffix is reduced from inside the backward)
merge_head           1         the two head towers' launches pairwise in one shared launch (rtp_multi_*)
pair_heads           1         SepHead's two first convs over a 128- / 256-channel feature as 64-wide launches (conv64_tiled.hip)
fwd_row0_first       1         the main lane's fuse-row feeders ahead of the other rows' chains in the forward list
bwd_f10_first        1         stage 3's row-1 stride-2 data gradient hoisted ahead of the row-2 chain
bwd_sink_wg          1         lower levels' weight gradients behind the other launches of their stage (four-stream map)
defer_wg             3         lanes whose launches are queued onto the main lane in front of its fan-ins ("" = none)
lazy_coef            1         GroupNorm-backward coefficients in the fan-in pass's prologue instead of a launch of their own
fuse_stats           1         fuse rows emit the statistics of what they store (no rtp_chan_stats pass)
fused_fold           1         GroupNorm fold in the tiled conv's own prologue (rtp_conv_gn_fused) instead of a fold launch
no_tail              0         1: one launch per deferred tail item instead of the batched tail (A/B, test_plan_logic)
fused_s2             0         1: the fused route for stride-2 data gradients (built and tested, slower: DESIGN.md 8)
dcn_cl               1         the DCN head's forward on the plan's own layout (rtp_dcn_cl_forward); 0: the fp32 NCHW operator
"""
import os

_DEFAULTS = dict(width_hints=None, lanes=None, graph_lanes="0,1,1,2,2,1", ar_buckets=1, merge_head=1, pair_heads=1,
                 fwd_row0_first=1, bwd_f10_first=1, bwd_sink_wg=1, defer_wg="3", lazy_coef=1, fuse_stats=1, fused_fold=1, no_tail=0,
                 fused_s2=0, dcn_cl=1)
_STRINGS = {"width_hints", "lanes", "graph_lanes", "defer_wg"}


class PlanOptions:
    __slots__ = tuple(_DEFAULTS)

    def __init__(self, **kw):
        for k, v in _DEFAULTS.items():
            setattr(self, k, v)
        self.update(kw)

    def update(self, kw):
        for k, v in kw.items():
            if k not in _DEFAULTS:
                raise KeyError("PlanOptions has no field %r (fields: %s)" % (k, ", ".join(_DEFAULTS)))
            setattr(self, k, v if (k in _STRINGS or v is None) else int(v))
        return self

    @classmethod
    def from_env(cls, spec=None):
        """PlanOptions() overridden by RTP_PLAN (or `spec`): "key=value;key=value"; a bare key means key=1."""
        o = cls()
        spec = os.environ.get("RTP_PLAN", "") if spec is None else spec
        for item in (p.strip() for p in spec.split(";")):
            if item:
                k, _, v = item.partition("=")
                o.update({k.strip(): (v.strip() if "=" in item else "1")})
        return o

    def int_list(self, field):
        v = getattr(self, field)
        return None if v is None else [int(x) for x in str(v).split(",") if x.strip() != ""]

    def __repr__(self):
        diff = {k: getattr(self, k) for k in _DEFAULTS if getattr(self, k) != _DEFAULTS[k]}
        return "PlanOptions(%s)" % ", ".join("%s=%r" % kv for kv in diff.items())
