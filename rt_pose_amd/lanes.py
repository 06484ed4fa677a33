"""Multi-stream replay of a static launch list.

HRNet keeps four resolution branches alive; between fuse points they are independent, and the weight-gradient chains
of the backward sweep only feed the optimiser.  The full-resolution kernels fill the chip, the low-resolution ones
(1-40 workgroups) do not, so the plan assigns every launch a LANE (= HIP stream): one per resolution group plus
lanes for the weight-gradient chains.  List order is a valid serial order; this module derives, once, the
cross-lane event waits that preserve every read-after-write / write-after-read / write-after-write ordering of that
serial order (vector clocks keep only the waits not already implied by stream order or earlier waits).  Replay is then
`wait events -> launch -> record event` per entry, capturable into a HIP graph like any stream fork/join.
"""

# lane ids
L_FULL, L_MID, L_LOW, L_WG, L_WG_LOW, L_LOW3 = 0, 1, 2, 3, 4, 5
NLANES = 6
# Lanes -> streams.  Measured on MI355X (hr3d, B=8, ms/step) with the main stream at high priority: one stream 8.9;
# "0,1,1,2,2" 7.1; one stream per lane "0,1,2,3,4" 6.96.  PlanOptions.lanes overrides for experiments.
import os
LANE_MAP = [0, 1, 2, 3, 4, 5]
# FOUR streams -- as many as HIP has hardware queues (rt_pose_amd.pin_hw_queues), so that no two streams share a queue and run each
# other's launches in submission order: the lowest level's lane shares a stream with the lower levels' weight gradients, the full-
# resolution weight-gradient lane (what is left of it once the head towers' launches are deferred onto the main lane: the stem's)
# with the level-2 lane.  Measured (round 4, hr3d, B = 8, width hints in place): 5.50 -> 5.38 ms per step (-2.2 %, three same-box
# triples; "0,1,2,3,3,3" 5.40, "0,1,2,0,3,3" 5.43, "0,1,2,3,2,3" 5.55, "0,1,2,2,3,1" 5.59); hr3d_dcn 11.51 -> 11.47.  The configs whose
# heads are channel-sliced keep their weight gradients on that lane and lose with it (hr3d_one_hm_doppler 10.23 -> 10.34): they stay
# on one stream per lane (engine.PoseEngine picks; RTP_LANES overrides both).
LANE_MAP_4 = [0, 1, 2, 2, 3, 3]


BUF_BYTES = {}   # data_ptr -> bytes of every buffer a launch names (tools/plan_times.py prices launches with it)


def _key(t):
    if t is None:
        return None
    buf = getattr(t, "buf", t)
    if hasattr(buf, "numel"):
        BUF_BYTES[buf.data_ptr()] = buf.numel() * buf.element_size()
    return buf.data_ptr()


class Launch:
    """One kernel launch of the plan: fn(stream_ptr), its lane, and the buffers it reads / writes."""
    __slots__ = ("fn", "lane", "reads", "writes", "tag")

    def __init__(self, fn, lane=0, reads=(), writes=(), tag=""):
        self.fn, self.lane, self.tag = fn, lane, tag
        self.reads = tuple(k for k in (_key(t) for t in reads) if k is not None)
        self.writes = tuple(k for k in (_key(t) for t in writes) if k is not None)

    def __call__(self, s):
        return self.fn(s)


def plan_waits(launches, nlanes=NLANES, lane_of=None):
    """-> (waits, record): waits[i] = indices of earlier launches (on other lanes) launch i must wait for;
    record[j] = launch j's completion needs an event.  lane_of: optional list overriding each launch's lane."""
    lane_of = lane_of if lane_of is not None else [L.lane for L in launches]
    last_on = {}
    last_w, readers = {}, {}
    clock = [[-1] * nlanes for _ in range(nlanes)]  # clock[l][m]: newest launch on lane m known finished before lane l's next
    after = []                                      # clock snapshot implied by the completion of launch i
    waits, record = [], [False] * len(launches)
    for i, L in enumerate(launches):
        deps = set()
        for k in L.reads:
            j = last_w.get(k)
            if j is not None:
                deps.add(j)
        for k in L.writes:
            j = last_w.get(k)
            if j is not None:
                deps.add(j)
            deps.update(readers.get(k, ()))
        lane = lane_of[i]
        last_on[lane] = i
        vc = clock[lane]
        need = {}
        for j in deps:
            m = lane_of[j]
            if m != lane and j > vc[m]:
                need[m] = max(need.get(m, -1), j)
        w = []
        for m, j in sorted(need.items(), key=lambda kv: -kv[1]):
            if j > vc[m]:
                w.append(j)
                record[j] = True
                for t in range(nlanes):
                    vc[t] = max(vc[t], after[j][t])
        vc[lane] = i
        after.append(list(vc))
        waits.append(w)
        for k in L.writes:
            last_w[k] = i
            readers[k] = []
        for k in L.reads:
            if k not in L.writes:
                readers.setdefault(k, []).append(i)
    return waits, record


def _order_preds(launches):
    """Read-after-write / write-after-read / write-after-write predecessors of every launch in the list's own order."""
    last_w, readers = {}, {}
    preds = [set() for _ in launches]
    for i, L in enumerate(launches):
        for k in L.reads:
            j = last_w.get(k)
            if j is not None:
                preds[i].add(j)
        for k in L.writes:
            j = last_w.get(k)
            if j is not None:
                preds[i].add(j)
            preds[i].update(readers.get(k, ()))
        for k in L.writes:
            last_w[k] = i
            readers[k] = []
        for k in L.reads:
            if k not in L.writes:
                readers.setdefault(k, []).append(i)
        preds[i].discard(i)
    return preds


def main_row_first(launches):
    """Forward list: inside every fuse block, the launches that feed the MAIN lane's fuse row (row 0: the 1x1x1 convs `sN.f0j` of
    the lower branches, with their weight folds) are issued before the other rows' chains.

    Rows are created last-to-first (net.build_backbone: the backward's fusable-consumer rule wants that creation order), so on
    the side lanes -- FIFO streams -- the cheap feeders of row 0 used to sit behind the stride-2 chains of rows 2 and 1, and the
    main lane waited for them after its own convolutions (0.33 / 0.16 / 0.15 ms in front of the three `fuse:sN.row0` launches,
    profiles/r03_main_lane_trace.txt).  Only launches of one block are permuted, groups keep their internal order, and the
    result is checked against the dependency relations of the original order (any violation: the original list is returned)."""
    import re
    n = len(launches)
    preds = _order_preds(launches)
    order = list(range(n))
    pos = 0
    out = []
    stage_of = lambda tag: (re.search(r"[:.]s(\d)\.(f\d\d|row\d)", tag) or [None, None])[1]
    i = 0
    while i < n:
        st = stage_of(launches[i].tag)
        if st is None:
            out.append(i)
            i += 1
            continue
        j = i
        while j < n and stage_of(launches[j].tag) == st:   # the block's launches are contiguous
            j += 1
        blk = list(range(i, j))
        feed0 = [k for k in blk if re.search(r"s\d\.f0\d", launches[k].tag)]
        row0 = [k for k in blk if launches[k].tag.endswith("s%s.row0" % st)]
        rest = [k for k in blk if k not in feed0 and k not in row0]
        out += feed0 + row0 + rest
        i = j
    where = {k: p for p, k in enumerate(out)}
    for k in range(n):
        if any(where[q] > where[k] for q in preds[k]):
            return launches
    return [launches[k] for k in out]


def hoist_tagged(launches, pattern, before):
    """Moves the launches whose tag matches `pattern` (kept in order) in front of the first launch whose tag matches `before`,
    provided that respects the dependency relations of the original order (otherwise the list is returned unchanged)."""
    import re
    idx = [i for i, L in enumerate(launches) if re.search(pattern, L.tag)]
    tgt = next((i for i, L in enumerate(launches) if re.search(before, L.tag)), None)
    if not idx or tgt is None or idx[0] < tgt:
        return launches
    moved = set(idx)
    out = [i for i in range(tgt) if i not in moved] + idx + [i for i in range(tgt, len(launches)) if i not in moved]
    preds = _order_preds(launches)
    where = {k: p for p, k in enumerate(out)}
    for k in range(len(launches)):
        if any(where[q] > where[k] for q in preds[k]):
            return launches
    return [launches[k] for k in out]


def sink_lane_in_segments(launches, lane, boundary):
    """Backward list: inside every segment (the list is cut in front of each launch whose tag matches `boundary`: the main lane's
    fan-in passes, one per stage) the launches of `lane` -- the lower levels' weight-gradient chains, whose results only the deferred
    tail reads -- are issued AFTER the segment's other launches (both groups keep their order).

    Why: under the four-stream map that lane shares a stream with the level-3 lane, and a stream runs its launches in list order.  In
    creation order every weight gradient sits right behind its data gradient, so stage 4's level-3 chain (combine -> dgrad -> combine
    -> dgrad ... -> dgrad:t3, the head of the dependency chain the MAIN lane waits for in front of its stage-3 fan-in) queued behind
    weight gradients that were themselves waiting for other lanes' tensors (round 5 timeline: combine:s4.b3.c3 started 240 us after
    its inputs were ready).  The dependency relations of the original order are checked; the list is returned unchanged if the
    permutation would break one."""
    import re
    cuts = [i for i, L in enumerate(launches) if re.search(boundary, L.tag)]
    out, lo = [], 0
    for hi in cuts + [len(launches)]:
        seg = list(range(lo, hi))
        out += [i for i in seg if launches[i].lane != lane] + [i for i in seg if launches[i].lane == lane]
        lo = hi
    if out == list(range(len(launches))):
        return launches
    preds = _order_preds(launches)
    where = {k: p for p, k in enumerate(out)}
    for k in range(len(launches)):
        if any(where[q] > where[k] for q in preds[k]):
            return launches
    return [launches[k] for k in out]


def merge_launches(launches, backend, pairs):
    """Horizontal fusion (include/rtp.h: rtp_multi_*): independent launches of one LDS-tiled kernel variant become ONE launch.

    pairs: (tag_a, tag_b[, tag_c[, tag_d]]) candidates, tried in order.  Launches b, c, d (e.g. a side lane's: the level-1 branch of an
    HRNet stage) join launch a (the main lane's full-resolution launch of the same position) on a's lane; the merged launch reads and
    writes what all of them did.
    A pair is accepted if (1) neither launch depends on the other -- directly or through other launches, merged ones included --
    and (2) the backend can build the shared launch (same kernel variant, eight samples, tiled geometry: HipBackend.multi).
    The list is then re-sorted topologically over the read-after-write / write-after-read / write-after-write relations of the
    ORIGINAL order (earliest original position first among the ready launches), so every dependency of the original list holds in
    the new one by construction.  -> (new list, [(tag_a, tag_b), ...] merged)."""
    if not hasattr(backend, "multi") or not pairs:
        return launches, []
    dbg = (lambda *a: print("[merge]", *a)) if os.environ.get("RTP_MERGE_DEBUG") else (lambda *a: None)
    n = len(launches)
    preds = _order_preds(launches)
    by_tag = {}
    for i, L in enumerate(launches):
        by_tag.setdefault(L.tag, []).append(i)
    node_of = list(range(n))            # launch index -> its node (the smaller index of a merged pair)
    fns, merged = {}, []

    def topo():
        """Stable topological order of the nodes, or None if merging made a cycle."""
        members = {}
        for i in range(n):
            members.setdefault(node_of[i], []).append(i)
        indeg = {v: 0 for v in members}
        succ = {v: set() for v in members}
        for i in range(n):
            for j in preds[i]:
                a, b = node_of[j], node_of[i]
                if a != b and b not in succ[a]:
                    succ[a].add(b)
                    indeg[b] += 1
                elif a == b and i != j:
                    return None         # the two launches of a pair depend on each other
        import heapq
        ready = [v for v in members if indeg[v] == 0]
        heapq.heapify(ready)
        order = []
        while ready:
            v = heapq.heappop(ready)
            order.append(v)
            for w in succ[v]:
                indeg[w] -= 1
                if indeg[w] == 0:
                    heapq.heappush(ready, w)
        return (order, members) if len(order) == len(members) else None

    for group in pairs:     # (tag_a, tag_b[, tag_c[, tag_d]]): everything joins the first launch
        if any(len(by_tag.get(t, ())) != 1 for t in group):
            dbg("tags not found once:", *group)
            continue
        idx = [by_tag[t][0] for t in group]
        if any(node_of[i] != i for i in idx):
            continue
        ia = idx[0]
        keep = [node_of[i] for i in idx]
        for i in idx[1:]:
            node_of[i] = ia
        if topo() is None:
            dbg("dependent launches (directly or through others):", *group)
            for i, k in zip(idx, keep):
                node_of[i] = k
            continue
        f = backend.multi([launches[i].fn for i in idx])
        if f is None:
            dbg("the backend cannot share a launch for", *group)
            for i, k in zip(idx, keep):
                node_of[i] = k
            continue
        fns[ia] = f
        merged.append(tuple(group))
    if not merged:
        return launches, []
    order, members = topo()
    new = []
    for v in order:
        if len(members[v]) == 1:
            new.append(launches[v])
            continue
        mem = [launches[v]] + [launches[i] for i in members[v] if i != v]
        m = Launch.__new__(Launch)
        m.fn, m.lane = fns[v], mem[0].lane
        m.tag = mem[0].tag + "".join("+" + x.tag.split(":", 1)[-1] for x in mem[1:])
        m.reads = tuple(dict.fromkeys(k for x in mem for k in x.reads))
        m.writes = tuple(dict.fromkeys(k for x in mem for k in x.writes))
        new.append(m)
    return new, merged


class LanePlan:
    """A launch list bound to a backend, replayable on one stream or on one stream per lane."""

    def __init__(self, backend, launches, lane_map=None):
        self.be = backend
        self.launches = [x if isinstance(x, Launch) else Launch(x) for x in launches]
        self.lane_map = lane_map
        self.lane_of = [lane_map[L.lane] if lane_map is not None else L.lane for L in self.launches]
        self.lanes_used = sorted(set(self.lane_of))
        self.waits, self.record = plan_waits(self.launches, NLANES, self.lane_of)
        self._events = None

    def __len__(self):
        return len(self.launches)

    def run(self, stream_ptr, multi=True):
        """stream_ptr: the caller's (main) stream handle.  multi=False replays everything on it in list order."""
        be = self.be
        if not multi or len(self.lanes_used) <= 1 or not hasattr(be, "lane_streams"):
            for L in self.launches:
                L.fn(stream_ptr)
            return
        streams, ptrs = be.lane_streams(NLANES)    # [current, side...], their handles
        if self._events is None:
            self._events = [be.new_event() if r else None for r in self.record]
            self._start = be.new_event()
            # the joins at the plan boundary carry the system fence (backend.HipBackend.new_event); only the lanes in use get one
            self._ends = [be.new_event(system_fence=True) if l in self.lanes_used and l != 0 else None for l in range(NLANES)]
        ev = self._events
        side = [l for l in self.lanes_used if l != 0]
        # the backend's ordering events take stream HANDLES: record(ptr) / wait(ptr) (HipBackend: device-scope HIP events behind the
        # C ABI, csrc/lane_events.hip)
        self._start.record(ptrs[0])
        for l in side:
            self._start.wait(ptrs[l])
        waits, record, lane_of = self.waits, self.record, self.lane_of
        trace = getattr(self, "trace", None)   # tools/main_lane_trace.py, tools/lane_timeline.py: {launch index: (start, end) timing events}
        for i, L in enumerate(self.launches):
            lane = lane_of[i]
            sp = ptrs[lane]
            for j in waits[i]:
                ev[j].wait(sp)
            if trace is not None and i in trace:
                trace[i][0].record(streams[lane])
            L.fn(sp)
            if trace is not None and i in trace:
                trace[i][1].record(streams[lane])
            if record[i]:
                ev[i].record(sp)
        for l in side:
            self._ends[l].record(ptrs[l])
            self._ends[l].wait(ptrs[0])
