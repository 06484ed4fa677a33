"""Loss logging with the reference's key names and record layout (the half that is not the checkpoint).

Host-side plumbing only -- no kernels.  What it reproduces, so that dashboards / parsers written for the reference's
`<timestamp>.log.json` and console lines read this framework's output unchanged:

  parse_losses    det3d/torchie/trainer/trainer.py:70-89 (parse_second_losses): the detector's loss dict
                  {loss, hm_loss, loc_loss, loc_loss_elem, num_positive} -> flat log_vars; the per-coordinate regression losses
                  are named coor_{x,y,z}_offset_<joint> in that order (3 names for hr3d's `reg`, 45 for the one-heat-map heads)
  LogBuffer       det3d/torchie/trainer/log_buffer.py: per-key history, count-weighted mean over the last n updates
  TextLogger      det3d/torchie/trainer/hooks/logger/text.py:42-149 + logger/base.py:36-47: every `interval` iterations one
                  console line "Epoch [e/E][i/I]\\tlr: ..., eta: ..., time: ..." , one "task : [...], key: value, ..." line per task
                  and one JSON record {mode, epoch, iter, lr, time, data_time, memory, <timers>, <log_vars>} (floats rounded to 5)

Pinned by tests/golden/train_log_golden.json, captured by running the reference's own classes (tests/golden/gen_golden_log.py).
"""
import datetime
import json
import os
from collections import OrderedDict

LOC_LOSS_ELEM_NAMES = ["coor_%s_offset_%d" % (ax, j) for j in range(15) for ax in "xyz"]
_TIMERS = ("time", "data_time", "transfer_time", "forward_time", "loss_parse_time")
_NOT_PER_TASK = ("mode", "Epoch", "iter", "lr", "memory", "epoch") + _TIMERS


def _scalar(v):
    return float(v.item()) if hasattr(v, "item") else float(v)


def parse_losses(losses):
    """losses: the detector's dict of per-task lists (RadarPoseNet.forward(return_loss=True) / PoseEngine.losses() wrapped in
    lists).  -> (loss to back-propagate = sum over tasks, OrderedDict of python floats keyed like the reference's log_vars)."""
    log_vars = OrderedDict()
    total = sum(losses["loss"])
    for name, value in losses.items():
        if name == "loc_loss_elem":
            for j, item in enumerate(value[0]):
                log_vars[LOC_LOSS_ELEM_NAMES[j]] = _scalar(item)
        elif name in ("num_pos", "num_neg"):
            log_vars[name] = value
        else:
            log_vars[name] = _scalar(value[0])
    return total, log_vars


def engine_losses_as_lists(d):
    """PoseEngine.losses() / a trainer's losses() (device scalars, one task) in the reference's per-task-list form."""
    return OrderedDict((k, [v]) for k, v in d.items())


class LogBuffer:
    """History of logged variables with windowed, count-weighted averages (`output` holds the last averages)."""

    def __init__(self):
        self.val_history, self.n_history, self.output, self.ready = OrderedDict(), OrderedDict(), OrderedDict(), False

    def clear(self):
        self.val_history.clear()
        self.n_history.clear()
        self.clear_output()

    def clear_output(self):
        self.output.clear()
        self.ready = False

    def update(self, variables, count=1):
        for k, v in variables.items():
            self.val_history.setdefault(k, []).append(v)
            self.n_history.setdefault(k, []).append(count)

    def average(self, n=0):
        """Mean of the latest n updates of every key (all of them for n = 0), weighted by the counts they came with."""
        for k, vals in self.val_history.items():
            v, c = vals[-n:], self.n_history[k][-n:]
            if v and isinstance(v[0], (list, tuple)):
                self.output[k] = [sum(col) / len(v) for col in zip(*v)]
            else:
                self.output[k] = sum(a * b for a, b in zip(v, c)) / sum(c)
        self.ready = True


def _round5(x):
    if isinstance(x, list):
        return [_round5(v) for v in x]
    return round(x, 5) if isinstance(x, float) else x


def _fmt4(x):
    if isinstance(x, list):
        return [_fmt4(v) for v in x]
    return "{:.4f}".format(x) if isinstance(x, float) else x


class TextLogger:
    """Console lines + JSON records every `interval` training iterations.

    class_names: per-task lists (CenterHead.class_names); json_path: the `<timestamp>.log.json` file (None: no file);
    sink: callable taking one console line (default: print)."""

    def __init__(self, class_names, max_epochs, iters_per_epoch, interval=10, json_path=None, sink=print, start_iter=0, rank=0):
        self.class_names, self.max_epochs, self.iters_per_epoch = class_names, max_epochs, iters_per_epoch
        self.interval, self.json_path, self.sink, self.start_iter, self.rank = interval, json_path, sink, start_iter, rank
        self.max_iters = max_epochs * iters_per_epoch
        self.buffer = LogBuffer()
        self.time_sec_tot = 0.0

    def before_epoch(self):
        self.buffer.clear()

    def after_train_iter(self, epoch, inner_iter, it, lr, log_vars, timers=None, memory_mb=None, num_samples=-1):
        """epoch / inner_iter / it: 0-based counters as the reference's trainer holds them when its hooks fire; lr: the first
        parameter group's learning rate; timers: dict with the five timer keys (train mode) or None; -> the lines emitted."""
        if timers:
            self.buffer.update({k: timers[k] for k in _TIMERS if k in timers})
        self.buffer.update(log_vars, num_samples)
        if (inner_iter + 1) % self.interval:
            return []
        self.buffer.average(self.interval)
        out = self.buffer.output
        rec = OrderedDict(mode="train" if "time" in out else "val", epoch=epoch + 1, iter=inner_iter + 1, lr=lr)
        if rec["mode"] == "train":
            rec["time"], rec["data_time"] = out["time"], out["data_time"]
            if memory_mb is not None:
                rec["memory"] = memory_mb
        for k, v in out.items():
            if k not in ("time", "data_time"):
                rec[k] = v
        lines = self._lines(rec, it)
        for ln in lines:
            self.sink(ln)
        if self.json_path and self.rank == 0:
            with open(self.json_path, "a+") as f:
                json.dump(OrderedDict((k, _round5(v)) for k, v in rec.items()), f)
                f.write("\n")
        self.buffer.clear_output()
        return lines

    def _lines(self, rec, it):
        if rec["mode"] == "train":
            head = "Epoch [{}/{}][{}/{}]\tlr: {:.5f}, ".format(rec["epoch"], self.max_epochs, rec["iter"], self.iters_per_epoch, rec["lr"])
            if "time" in rec:
                self.time_sec_tot += rec["time"] * self.interval
                avg = self.time_sec_tot / (it - self.start_iter + 1)
                head += "eta: {}, ".format(datetime.timedelta(seconds=int(avg * (self.max_iters - it - 1))))
                head += "time: {:.3f}, data_time: {:.3f}, transfer_time: {:.3f}, forward_time: {:.3f}, loss_parse_time: {:.3f} ".format(
                    rec["time"], rec["data_time"], rec["transfer_time"] - rec["data_time"],
                    rec["forward_time"] - rec["transfer_time"], rec["loss_parse_time"] - rec["forward_time"])
                head += "memory: {}, ".format(rec.get("memory"))
        else:
            head = "Epoch({}) [{}][{}]\t".format(rec["mode"], rec["epoch"] - 1, rec["iter"])
        lines = [head]
        for idx, names in enumerate(self.class_names):
            items = ["task : {}".format(names)]
            for k, v in rec.items():
                if k in _NOT_PER_TASK:
                    continue
                items.append("{}: {}".format(k, _fmt4(v[idx]) if isinstance(v, list) else _fmt4(v)))
            lines.append(", ".join(items) + ("\n" if idx == len(self.class_names) - 1 else ""))
        return lines
