"""Single-node data-parallel training driver for HRRadarPose on MI355X.

One process per GPU (torchrun); rank r owns frames [r*b, (r+1)*b) of every global batch (the reference's
DistributedGroupSampler slice, datasets/loader/sampler.py:185-217); weights are replicated.  Per step:
    forward + losses + backward   -- ONE captured HIP graph replay (rt_pose_amd.engine.PoseEngine lists)
    all-reduce of the flat fp32 gradient buffer over RCCL/xGMI (a single collective: the whole model is 8-33 MB)
    clip + decoupled weight decay + Adam  -- a handful of fused launches over the flat buffers
GroupNorm needs no statistics exchange (per-sample), and the loss normalisers are per-rank sums exactly as in the
reference (centernet_loss.py:22,48-54), so nothing else crosses GPUs.  The reference's second, redundant
all-reduce (core/utils/dist_utils.py:45-57) is not reproduced.
"""
import math
import os
from collections import OrderedDict

import torch

from . import configs
from .backend import HipBackend
from .engine import FlatAdam, FlatParams, PoseEngine, one_cycle


def init_state_dict(shapes, seed=0):
    """Reference-default initialisation, seeded (identical on every rank): nn.Conv3d kaiming_uniform(a=sqrt(5)) +
    uniform bias, GroupNorm ones/zeros, head convs kaiming_normal(fan_out, relu) with zero bias except the
    heat-map tower (default conv init) whose last bias is -2.19 (center_head.py:94-99)."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for name, shape in shapes.items():
        if ".feature_adapt_" in name:   # FeatureAdaption (center_head.py:44-57): zero offset weights, default conv init otherwise
            wshape = shapes[name.rsplit(".", 1)[0] + ".weight"]
            bound = 1.0 / math.sqrt(wshape[1] * wshape[2] * wshape[3])
            if name.endswith("conv_offset.weight"):
                sd[name] = torch.zeros(shape)
            else:
                sd[name] = (torch.rand(shape, generator=g) * 2 - 1) * bound
            continue
        is_gn = len(shapes[name.rsplit(".", 1)[0] + ".weight"]) == 1
        if is_gn:
            sd[name] = torch.ones(shape) if name.endswith(".weight") else torch.zeros(shape)
            continue
        wshape = shapes[name.rsplit(".", 1)[0] + ".weight"]
        fan_in = wshape[1] * wshape[2] * wshape[3] * wshape[4]
        fan_out = wshape[0] * wshape[2] * wshape[3] * wshape[4]
        reg_tower = ".tasks." in name and ".hm." not in name
        if name.endswith(".weight"):
            if reg_tower:
                sd[name] = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_out)
            else:
                bound = 1.0 / math.sqrt(fan_in)
                sd[name] = (torch.rand(shape, generator=g) * 2 - 1) * bound
        elif name.endswith("hm.2.bias"):
            sd[name] = torch.full(shape, -2.19)
        elif reg_tower:
            sd[name] = torch.zeros(shape)
        else:
            bound = 1.0 / math.sqrt(fan_in)
            sd[name] = (torch.rand(shape, generator=g) * 2 - 1) * bound
    return sd


class DataParallelTrainer:
    def __init__(self, name="hr3d", batch_per_gpu=8, dims=configs.NATIVE_DIMS, total_steps=1000, lr_max=None,
                 device="cuda:0", rank=0, world_size=1, use_graph=True, seed=0, backend=None, process_group=None, ar_buckets=None,
                 stream=None, options=None):
        self.spec = s = configs.spec(name)
        self.name, self.rank, self.world = name, rank, world_size
        if backend is None and str(device).startswith("cuda"):
            from . import pin_hw_queues
            self.hw_queues = pin_hw_queues()   # before the HIP runtime initialises (no effect, and a warning, afterwards)
        self.be = backend if backend is not None else HipBackend(device)
        self.shapes = configs.param_shapes(name)
        self.flat = FlatParams(self.shapes, self.be.alloc)
        self.flat.load_state_dict(init_state_dict(self.shapes, seed))
        from .options import PlanOptions
        opt = self.options = options if options is not None else PlanOptions.from_env()
        # Gradient all-reduce in ONE bucket (the whole flat buffer after the backward sweep) or in TWO (options.ar_buckets / ar_buckets=2,
        # multi-rank eager mode only): the reference's DDP overlaps bucketed all-reduces with the tail of backward
        # (det3d/torchie/apis/train.py:284-291); here the deferred tail is flushed once early (graph.Graph.early_flush) so that
        # the gradients of transition2 .. pose_head -- a contiguous suffix of the flat buffer -- are reduced on the process
        # group's stream while stage 2 / layer 1 are still being swept.  One bucket stays the default (A/B: `allreduce_buckets`
        # in the bench line); with 8-33 MB of gradients per step the collective is a few percent of the step either way.
        if ar_buckets is None:
            ar_buckets = int(opt.ar_buckets)
        self.ar_buckets = 2 if (ar_buckets == 2 and world_size > 1 and not (use_graph and self.be.name == "hip")) else 1
        self.engine = PoseEngine(self.be, self.flat.values, s["arch"], s["final_fuse"], s["heads"], s["weight"],
                                 s["code_weights"], batch_per_gpu, dims, train=True, pgrads=self.flat.grads,
                                 test_cfg=configs.test_cfg(), lidar_channels=s.get("lidar_channels", 0),
                                 early_flush=self.ar_buckets == 2, options=opt)
        self._ar_pending = None
        if self.ar_buckets == 2:
            self._install_early_bucket()
        self.opt = FlatAdam(self.be, self.flat, self.engine.live_params)
        self.total_steps, self.lr_max = total_steps, lr_max if lr_max is not None else s["lr_max"]
        self.step_idx = 0
        self.pg = process_group
        self.use_graph = use_graph and self.be.name == "hip"
        self._graph = None
        if self.use_graph and opt.lanes is None:
            # hipStreamEndCapture crashes (ROCm 7.2) on a capture that forks into all six lane streams; three streams
            # capture and replay fine, so graph mode folds the lanes: {full}, {mid, low, lowest}, {weight gradients}
            from .lanes import LanePlan
            gmap = opt.int_list("graph_lanes")   # (A/B: "0,1,2,2,3,3" = lanes.LANE_MAP_4)
            self.engine.fwd_plan = LanePlan(self.be, self.engine.fwd, gmap)
            self.engine.bwd_plan = LanePlan(self.be, self.engine.bwd, gmap)
        # All work of a step goes to ONE explicit (non-default) HIP stream: a graph launched on the legacy NULL stream
        # was observed NOT to be ordered against the optimiser kernels queued behind it on ROCm 7.x.
        # High priority: this stream carries the full-resolution chain, the critical path of the lane plan (side lanes
        # keep the default priority, so their small kernels fill in around it instead of delaying it).
        prio = -1
        # (stream: reuse another trainer's step stream.  A second set of streams lands on the four hardware queues in a different
        # pattern -- which lanes share a queue decides how the persistent kernels time-share the CUs: measured 12.6 instead of
        # 10.8 ms/step for a second model built on streams of its own in the same process)
        self.stream = stream if stream is not None else (torch.cuda.Stream(self.be.device, priority=prio) if self.be.name == "hip" else None)
        if self.stream is not None:
            # construction-time work (zero fills, parameter upload, code weights) was queued on the current stream; the
            # step stream is non-blocking with respect to it, so order the first step behind it explicitly
            self.stream.wait_stream(torch.cuda.current_stream(self.be.device))
            self._step_done = torch.cuda.Event()
            self._ar_events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(64)]
            self._ar_count = 0

    def _install_early_bucket(self):
        """Split the flat gradient buffer at the first parameter the early tail flush finalises and queue that suffix's all-reduce
        as a launch of the backward list, right behind the flush (same lane: stream order makes it wait for exactly that flush)."""
        from .lanes import LanePlan, Launch, L_WG_LOW
        eng, g = self.engine, self.engine.graph
        # the flush is looked up by identity AFTER every re-ordering of the list (engine.py hoists launches by tag): its creation
        # index (g.early_tail_index) is a position in Graph.bwd, not in eng.bwd
        tail = getattr(g, "early_tail_launch", None)
        if tail is None or not any(L is tail for L in eng.bwd):
            self.ar_buckets = 1
            return
        idx = next(i for i, L in enumerate(eng.bwd) if L is tail)
        assert eng.bwd[idx].tag.startswith("tail"), "the early bucket must sit right behind a tail flush, found %r" % eng.bwd[idx].tag
        ptr2name = {t.data_ptr(): k for k, t in self.flat.grads.items()}
        early = {ptr2name[w] for L in eng.bwd[:idx + 1] for w in L.writes if w in ptr2name}
        late = {ptr2name[w] for L in eng.bwd[idx + 1:] for w in L.writes if w in ptr2name}
        assert early and not (early & late), "a parameter gradient is written on both sides of the early flush"
        split = min(self.flat.offsets[k] for k in early)
        assert all(self.flat.offsets[k] < split for k in late), "the early bucket is not a suffix of the flat gradient buffer"
        self.ar_split = split
        bucket = self.flat.g[split:]
        grads = [self.flat.grads[k] for k in early]

        def kick(stream_ptr):
            import torch.distributed as dist
            if stream_ptr is not None and self.be.name == "hip":
                raw = stream_ptr.value if hasattr(stream_ptr, "value") else int(stream_ptr)   # c_void_p(0).value is None: the null stream
                ext = torch.cuda.ExternalStream(raw or 0, device=self.be.device)
                with torch.cuda.stream(ext):   # the collective is ordered behind this lane's work and runs on the group's own stream
                    self._ar_pending = dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
            else:
                self._ar_pending = dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        eng.bwd.insert(idx + 1, Launch(kick, L_WG_LOW, grads, grads, "allreduce:early"))
        eng.bwd_plan = LanePlan(self.be, eng.bwd, eng.bwd_plan.lane_map)   # (the map the plan it replaces was built with)

    # ------------------------------------------------------------------ one step
    def _on_stream(self):
        import contextlib
        return torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def load(self, example):
        with self._on_stream():
            self.engine.load_input(example["rdr"]["rdr_tensor"])
            if self.engine.lidar_in is not None:
                self.engine.load_lidar(example["rdr"]["lidar_grid"])
            self.engine.load_targets(example["rdr"])

    def _fwd_bwd(self):
        self.engine.run_forward()
        self.engine.run_loss_backward()

    def _capture(self):
        # run once eagerly (lazy initialisation happens outside capture), then record the fwd+loss+bwd lists
        self._fwd_bwd()
        torch.cuda.synchronize(self.be.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=self.stream):
            self._fwd_bwd()
        self._graph = g

    def step(self, example=None):
        """forward + loss + backward + gradient all-reduce + optimiser step on the currently loaded batch."""
        if example is not None:
            self.load(example)
        with self._on_stream():
            self._step()
            if self.stream is not None:
                self._step_done.record(self.stream)   # feed_raw after a plain step() waits on a recorded event

    def _step(self):
        if self.use_graph:
            if self._graph is None:
                self._capture()
            self._graph.replay()
        else:
            self._fwd_bwd()
        scale = 1.0
        if self.world > 1:
            import torch.distributed as dist
            # ONE collective over the flat fp32 gradient buffer, queued the moment the backward's last launch (the batched
            # slab folds, which finish every weight gradient at once) is queued; bracketed by events on the step stream
            # (the process group's internal stream is ordered against it on both sides) for allreduce_ms()
            timed = self.stream is not None
            if timed:
                e0, e1 = self._ar_events[self.step_idx % len(self._ar_events)]
                e0.record(self.stream)
            if self.ar_buckets == 2:   # the suffix is already on its way (queued behind the early tail flush); now the prefix
                dist.all_reduce(self.flat.g[:self.ar_split], op=dist.ReduceOp.SUM, group=self.pg)
                if self._ar_pending is not None:
                    self._ar_pending.wait()   # (stream-ordered on the GPU: the step stream waits for the early bucket)
                    self._ar_pending = None
            else:
                dist.all_reduce(self.flat.g, op=dist.ReduceOp.SUM, group=self.pg)
            if timed:
                e1.record(self.stream)
                self._ar_count += 1
            scale = 1.0 / self.world
        lr, beta1 = one_cycle(self.step_idx, self.total_steps, self.lr_max)
        self.opt.set_hyper(lr, beta1, grad_scale=scale)
        self.opt.run()
        self.step_idx += 1

    # ------------------------------------------------------------------ raw-input feeding
    def attach_input_pipeline(self, pipe):
        """pipe: rt_pose_amd.input_pipeline.DeviceInputPipeline built on this trainer's engine."""
        self.pipe = pipe
        self._fed = None

    def feed_raw(self, cubes_f16, poses):
        """Queue the NEXT batch from raw fp16 cubes + key-points: its H2D copy overlaps the step in flight, its two
        preparation kernels run right after that step."""
        self._fed = self.pipe.submit(cubes_f16, poses, after=self._step_done if self.step_idx > 0 else None)

    def step_fed(self):
        """One training step on the batch queued by feed_raw."""
        with self._on_stream():
            self.stream.wait_event(self._fed)
            self._step()
            self._step_done.record(self.stream)

    def losses(self):
        return self.engine.losses()

    def log_vars(self):
        """(loss, log_vars) of the last step under the reference's log key names (rt_pose_amd.train_log.parse_losses:
        loss, hm_loss, loc_loss, coor_{x,y,z}_offset_<joint>, num_positive) -- feed them to train_log.TextLogger.  One host sync."""
        from .train_log import engine_losses_as_lists, parse_losses
        return parse_losses(engine_losses_as_lists(self.losses()))

    def allreduce_ms(self):
        """Mean duration of the gradient all-reduce over the last (up to 64) steps, by events on the step stream; None when
        single-rank.  Synchronises the device."""
        if self.world <= 1 or self.stream is None or self._ar_count == 0:
            return None
        torch.cuda.synchronize(self.be.device)
        k = min(self._ar_count, len(self._ar_events))
        idx = [(self.step_idx - 1 - i) % len(self._ar_events) for i in range(k)]
        return sum(self._ar_events[i][0].elapsed_time(self._ar_events[i][1]) for i in idx) / k

    def forward_only(self):
        with self._on_stream():
            self.engine.run_forward()
