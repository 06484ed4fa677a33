"""Lane scheduler (rt_pose_amd/lanes.py): the cross-stream waits derived from read/write sets must order every
conflicting pair of launches exactly as the serial list does -- checked on random plans and on the real hr3d plan."""
import random

import torch

from rt_pose_amd import lanes
from rt_pose_amd.lanes import Launch, plan_waits


class _Buf:
    def __init__(self, k):
        self.k = k

    def data_ptr(self):
        return self.k


def _relane(L, lane):
    c = Launch(None, lane)
    c.reads, c.writes = L.reads, L.writes
    return c


def _happens_before(launches, waits):
    """reach[i] = set of launches guaranteed complete before i starts (stream order + event waits, transitively)."""
    last_on_lane, reach = {}, []
    for i, L in enumerate(launches):
        r = set()
        p = last_on_lane.get(L.lane)
        if p is not None:
            r |= reach[p] | {p}
        for j in waits[i]:
            assert j < i and launches[j].lane != L.lane
            r |= reach[j] | {j}
        reach.append(r)
        last_on_lane[L.lane] = i
    return reach


def _check(launches, lane_of=None):
    waits, record = plan_waits(launches, lanes.NLANES, lane_of)
    if lane_of is not None:
        launches = [Launch(None, l) for l in lane_of] and [_relane(L, l) for L, l in zip(launches, lane_of)]
    reach = _happens_before(launches, waits)
    for i, a in enumerate(launches):
        for j in range(i):
            b = launches[j]
            conflict = (set(a.reads) & set(b.writes)) or (set(a.writes) & set(b.reads)) or (set(a.writes) & set(b.writes))
            if conflict:
                assert j in reach[i], (j, i, conflict)
    for i, w in enumerate(waits):
        for j in w:
            assert record[j]
    return sum(len(w) for w in waits)


def test_random_plans_are_ordered():
    rng = random.Random(0)
    for trial in range(30):
        bufs = [_Buf(k) for k in range(rng.randint(3, 12))]
        ls = []
        for _ in range(rng.randint(5, 80)):
            ls.append(Launch(None, rng.randrange(lanes.NLANES), rng.sample(bufs, rng.randint(0, 3)), rng.sample(bufs, rng.randint(0, 2))))
        _check(ls)


def test_no_waits_within_one_lane_and_minimal_chain():
    a, b, c = _Buf(1), _Buf(2), _Buf(3)
    ls = [Launch(None, 0, [], [a]), Launch(None, 0, [a], [b]), Launch(None, 1, [b], [c]), Launch(None, 1, [a, c], [])]
    waits, record = plan_waits(ls)
    assert waits == [[], [], [1], []]          # the second lane-1 launch is covered by stream order + the first wait
    assert record == [False, True, False, False]


def test_hr3d_plan_hazards_are_covered():
    from tests.emu_backend import EmuBackend
    from rt_pose_amd import configs
    from rt_pose_amd.engine import PoseEngine, FlatParams
    from rt_pose_amd.trainer import init_state_dict
    be = EmuBackend()
    s = configs.spec("hr3d")
    shapes = configs.param_shapes("hr3d")
    flat = FlatParams(shapes, be.alloc)
    flat.load_state_dict(init_state_dict(shapes, 0))
    eng = PoseEngine(be, flat.values, s["arch"], s["final_fuse"], s["heads"], s["weight"], s["code_weights"], 1,
                     (8, 16, 32), train=True, pgrads=flat.grads)
    assert {L.lane for L in eng.fwd} >= {lanes.L_FULL, lanes.L_MID, lanes.L_LOW}
    # (the full-resolution weight-gradient lane L_WG is issued on the main lane by default: graph.Graph.emit_bwd, PlanOptions.defer_wg)
    assert {L.lane for L in eng.bwd} >= {lanes.L_FULL, lanes.L_MID, lanes.L_LOW, lanes.L_WG_LOW}
    assert any(L.tag.startswith("wgrad:head.") and L.lane == lanes.L_FULL for L in eng.bwd)
    nf, nb = _check(eng.fwd), _check(eng.bwd)
    # and under the default lane -> stream mapping the engine replays with
    assert eng.fwd_plan.waits == plan_waits(eng.fwd, lanes.NLANES, eng.fwd_plan.lane_of)[0]
    _check(eng.fwd, eng.fwd_plan.lane_of)
    _check(eng.bwd, eng.bwd_plan.lane_of)
    assert 0 < nf < len(eng.fwd) and 0 < nb < len(eng.bwd)     # far fewer events than launches
    # every launch declares what it writes (a launch with no write set could never be ordered)
    assert all(L.writes for L in eng.fwd + eng.bwd)


def test_merge_launches_keeps_every_dependency():
    """lanes.merge_launches (horizontal fusion of independent launches): a side-lane launch joins its main-lane twin, the list is
    re-sorted so that every dependency of the original order still holds (the transition conv that feeds the side launch moves in
    front of the merged launch), launches that depend on each other are never merged, and a backend that refuses leaves the list alone."""
    from rt_pose_amd.lanes import Launch, merge_launches, _order_preds

    class T:
        def __init__(self, i):
            self.i = i

        def data_ptr(self):
            return self.i

        def numel(self):
            return 1

        def element_size(self):
            return 1

    class BE:
        def __init__(self, refuse=()):
            self.refuse, self.asked = set(refuse), []

        def multi(self, fns):
            names = tuple(f.__name__ for f in fns)
            self.asked.append(names)
            if names in self.refuse:
                return None
            return lambda s: [f(s) for f in fns]

    t = [T(i) for i in range(10)]
    ran = []

    def mk(name):
        def f(s):
            ran.append(name)
        f.__name__ = name
        return f

    def build():
        return [Launch(mk("a1"), 0, [t[0]], [t[1]], "conv:a1"), Launch(mk("a2"), 0, [t[1]], [t[2]], "conv:a2"),
                Launch(mk("t"), 1, [t[0]], [t[3]], "conv:t"), Launch(mk("b1"), 1, [t[3]], [t[4]], "conv:b1"),
                Launch(mk("b2"), 1, [t[4]], [t[5]], "conv:b2"), Launch(mk("f"), 0, [t[2], t[5]], [t[6]], "fuse")]

    L = build()
    new, merged = merge_launches(L, BE(), [("conv:a1", "conv:b1"), ("conv:a2", "conv:b2"), ("conv:a1", "conv:a2"), ("conv:x", "conv:b1")])
    assert merged == [("conv:a1", "conv:b1"), ("conv:a2", "conv:b2")]
    assert [x.tag for x in new] == ["conv:t", "conv:a1+b1", "conv:a2+b2", "fuse"]
    assert all(x.lane == 0 for x in new[1:3]) and set(new[1].reads) == {0, 3} and set(new[1].writes) == {1, 4}
    for x in new:
        x.fn(None)
    assert ran == ["t", "a1", "b1", "a2", "b2", "f"]
    # dependent launches (a2 reads what a1 writes) are refused even when the backend would take them
    be = BE()
    new2, merged2 = merge_launches(build(), be, [("conv:a1", "conv:a2")])
    assert merged2 == [] and [x.tag for x in new2] == [x.tag for x in build()] and be.asked == []
    # ... and so is a pair that would close a cycle THROUGH other launches: t feeds b1, a2 depends on a1 -> (t, a2)? independent: fine;
    # (a1, b2) with (a2, b1) already merged would need a1 < a2 = b1 < b2 = a1
    new3, merged3 = merge_launches(build(), BE(), [("conv:a2", "conv:b1"), ("conv:a1", "conv:b2")])
    assert merged3 == [("conv:a2", "conv:b1")]
    preds = _order_preds(build())
    where = {}
    for k, x in enumerate(new3):
        for name in x.tag.replace("conv:", "").split("+"):
            where[name] = k
    names = ["a1", "a2", "t", "b1", "b2", "fuse"]
    for i, ps in enumerate(preds):
        for j in ps:
            assert where[names[j]] <= where[names[i]]
    # a backend that cannot build the shared launch: nothing changes
    new4, merged4 = merge_launches(build(), BE(refuse={("a1", "b1")}), [("conv:a1", "conv:b1")])
    assert merged4 == [] and [x.tag for x in new4] == [x.tag for x in build()]


def test_width_hint_rules_reach_the_right_launches():
    """Width rules are resolved when the plan is built (graph.Graph.with_width): first matching tag prefix wins, the width travels
    as an explicit field of that launch's geometry (Geom.wgs -> RtpConvGeom::wgs), every other launch keeps wgs = 0 -- no table on
    the C side that could outlive the plan."""
    from rt_pose_amd.engine import PoseEngine, DEFAULT_WIDTH_HINTS, parse_width_hints
    from rt_pose_amd.graph import Graph, Geom

    rules = parse_width_hints("conv:s3.b0=192;wgrad:s3.b0=176;dgrad:s3.b0.c3=208;conv:s3=64;conv:s4=0")
    assert rules == [("conv:s3.b0", 192), ("wgrad:s3.b0", 176), ("dgrad:s3.b0.c3", 208), ("conv:s3", 64)]
    g = Graph.__new__(Graph)
    g.width_rules, g.widths_applied = rules, []
    ge = Geom(8, 16, 64, 160, 16, 64, 160, 32, 32, 3, 1, 1)
    got = {t: g.with_width(ge, t).wgs for t in ("conv:s3.b0.c2", "conv:s3.b1.c2", "fuse:s3.row0", "wgrad:s3.b0.c2", "dgrad:s3.b0.c3",
                                                "dgrad:s3.b0.c2")}
    assert got == {"conv:s3.b0.c2": 192, "conv:s3.b1.c2": 64, "fuse:s3.row0": 0, "wgrad:s3.b0.c2": 176, "dgrad:s3.b0.c3": 208,
                   "dgrad:s3.b0.c2": 0}
    assert ge.wgs == 0 and g.with_width(ge, "fuse:s3.row0") is ge, "the shared geometry object is never modified"
    assert g.widths_applied == [("conv:s3.b0.c2", 192), ("conv:s3.b1.c2", 64), ("wgrad:s3.b0.c2", 176), ("dgrad:s3.b0.c3", 208)]
    assert parse_width_hints("") == [] and parse_width_hints(None) == []
    # the shipped default names launches that exist in the hr3d plan
    pre = [r.split("=")[0] for r in DEFAULT_WIDTH_HINTS.split(";")]
    from tests.emu_backend import EmuBackend
    from rt_pose_amd import configs
    from rt_pose_amd.engine import FlatParams
    from rt_pose_amd.trainer import init_state_dict
    be = EmuBackend()
    s = configs.spec("hr3d")
    shapes = configs.param_shapes("hr3d")
    flat = FlatParams(shapes, be.alloc)
    flat.load_state_dict(init_state_dict(shapes, 0))
    real = PoseEngine(be, flat.values, s["arch"], s["final_fuse"], s["heads"], s["weight"], s["code_weights"], 8, (4, 8, 16), train=True, pgrads=flat.grads)
    tags = [x.tag for x in real.fwd + real.bwd]
    assert all(any(t.startswith(p) for t in tags) for p in pre), [p for p in pre if not any(t.startswith(p) for t in tags)]
    # ... and the eight-sample plan (the bench's batch) carries them: every rule reached at least one launch
    assert real.width_hints and all(any(t.startswith(p) for t, _ in real.width_hints) for p in pre)


def test_four_stream_map_is_chosen_where_it_pays_and_orders_every_hazard():
    """PoseEngine picks lanes.LANE_MAP_4 (four streams = four hardware queues) for plans without channel-sliced heads -- any batch:
    B = 4 / 8 / 16 measured, round 5 -- and one stream per lane otherwise; the waits derived under the four-stream map still order
    every read/write hazard of both lists, and the default width rules reach the same launches whatever the batch."""
    from tests.emu_backend import EmuBackend
    from rt_pose_amd import configs
    from rt_pose_amd.engine import PoseEngine, FlatParams
    from rt_pose_amd.trainer import init_state_dict

    def build(name, batch):
        be = EmuBackend()
        s = configs.spec(name)
        shapes = configs.param_shapes(name)
        flat = FlatParams(shapes, be.alloc)
        flat.load_state_dict(init_state_dict(shapes, 0))
        return PoseEngine(be, flat.values, s["arch"], s["final_fuse"], s["heads"], s["weight"], s["code_weights"], batch, (4, 8, 16),
                          train=True, pgrads=flat.grads)

    eng = build("hr3d", 8)
    assert eng.lane_map == lanes.LANE_MAP_4 and eng.fwd_plan.lane_map == lanes.LANE_MAP_4 and eng.bwd_plan.lane_map == lanes.LANE_MAP_4
    assert len(set(eng.bwd_plan.lane_of)) <= 4
    _check(eng.fwd, eng.fwd_plan.lane_of)
    _check(eng.bwd, eng.bwd_plan.lane_of)
    for b in (2, 4, 16):
        other = build("hr3d", b)
        assert other.lane_map == lanes.LANE_MAP_4
        assert [t for t, _ in other.width_hints] == [t for t, _ in eng.width_hints] and other.width_hints
        _check(other.bwd, other.bwd_plan.lane_of)
    assert build("hr3d_one_hm_doppler", 8).lane_map == lanes.LANE_MAP


def test_merge_launches_groups_of_three_and_four():
    """merge_launches with more than two launches per group (the head towers' four weight gradients, a four-launch group): everything
    joins the group's first launch, reads / writes are the union, a group with one dependent member is refused as a whole."""
    from rt_pose_amd.lanes import Launch, merge_launches, _order_preds

    class T:
        def __init__(self, i):
            self.i = i

        def data_ptr(self):
            return self.i

        def numel(self):
            return 1

        def element_size(self):
            return 1

    class BE:
        def __init__(self):
            self.asked = []

        def multi(self, fns):
            self.asked.append(tuple(f.__name__ for f in fns))
            return lambda s: [f(s) for f in fns]

    t = [T(i) for i in range(12)]
    ran = []

    def mk(name):
        def f(s):
            ran.append(name)
        f.__name__ = name
        return f

    def build():   # four independent weight gradients w0..w3 of one producer p, a consumer c of all, and d that depends on w0
        return [Launch(mk("p"), 0, [t[0]], [t[1]], "p"), Launch(mk("w0"), 0, [t[1]], [t[2]], "wgrad:w0"), Launch(mk("d"), 0, [t[2]], [t[6]], "wgrad:d"),
                Launch(mk("w1"), 3, [t[1]], [t[3]], "wgrad:w1"), Launch(mk("w2"), 3, [t[1]], [t[4]], "wgrad:w2"), Launch(mk("w3"), 4, [t[1]], [t[5]], "wgrad:w3"),
                Launch(mk("c"), 0, [t[2], t[3], t[4], t[5], t[6]], [t[7]], "tail")]

    be = BE()
    new, merged = merge_launches(build(), be, [("wgrad:w0", "wgrad:w1", "wgrad:w2", "wgrad:w3")])
    assert merged == [("wgrad:w0", "wgrad:w1", "wgrad:w2", "wgrad:w3")] and be.asked == [("w0", "w1", "w2", "w3")]
    assert [x.tag for x in new] == ["p", "wgrad:w0+w1+w2+w3", "wgrad:d", "tail"]
    m = new[1]
    assert m.lane == 0 and set(m.reads) == {1} and set(m.writes) == {2, 3, 4, 5}
    for x in new:
        x.fn(None)
    assert ran == ["p", "w0", "w1", "w2", "w3", "d", "c"]
    # a group with a member that depends on another member (d reads what w0 writes) is refused as a whole, smaller groups still go
    be = BE()
    new2, merged2 = merge_launches(build(), be, [("wgrad:w0", "wgrad:w1", "wgrad:d"), ("wgrad:w1", "wgrad:w2", "wgrad:w3")])
    assert merged2 == [("wgrad:w1", "wgrad:w2", "wgrad:w3")] and be.asked == [("w1", "w2", "w3")]
    assert [x.tag for x in new2] == ["p", "wgrad:w0", "wgrad:d", "wgrad:w1+w2+w3", "tail"]
    assert all(x.lane == 3 for x in new2 if x.tag.startswith("wgrad:w1"))
