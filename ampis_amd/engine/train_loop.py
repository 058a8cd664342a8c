"""Training loop pieces of the detectron2 surface AMPIS uses (ampis/data_utils.py:57,104-106,128-132):
EventStorage (`trainer.storage.put_scalar(s)`), the WarmupMultiStepLR schedule, and the checkpoint / writer hooks that
DefaultTrainer.build_hooks returns (the writer is LAST: AmpisTrainer inserts its LossEvalHook at index -1)."""
import bisect
import json
import logging
import os
import time
from collections import defaultdict

from .hooks import HookBase

logger = logging.getLogger("ampis_amd")


class EventStorage:
    def __init__(self, start_iter=0):
        self.iter = start_iter
        self._history = defaultdict(list)          # name -> [(value, iter)]
        self._latest = {}

    def put_scalar(self, name, value, smoothing_hint=True):
        value = float(value)
        self._history[name].append((value, self.iter))
        self._latest[name] = (value, self.iter)

    def put_scalars(self, *, smoothing_hint=True, **kwargs):
        for k, v in kwargs.items():
            self.put_scalar(k, v, smoothing_hint=smoothing_hint)

    def history(self, name):
        return self._history[name]

    def histories(self):
        return self._history

    def latest(self):
        return self._latest

    def step(self):
        self.iter += 1


def warmup_multistep_lr(it, base_lr, steps, gamma, warmup_iters, warmup_factor):
    """detectron2 solver/lr_scheduler.py WarmupMultiStepLR (linear warm-up)."""
    if it < warmup_iters:
        alpha = it / warmup_iters
        w = warmup_factor * (1 - alpha) + alpha
    else:
        w = 1.0
    return base_lr * w * gamma ** bisect.bisect_right(list(steps), it)


class PeriodicCheckpointer(HookBase):
    """model_{iter:07d}.pth every `period` iterations and model_final.pth at the end (names sort so that
    sorted(glob('*.pth'))[-1] is the final model, which notebook cell 24 relies on)."""

    def __init__(self, period, output_dir):
        self.period = int(period)
        self.output_dir = output_dir

    def after_step(self):
        t = self.trainer
        it = t.iter + 1
        if not t.is_main_process:
            return
        if self.period > 0 and it % self.period == 0 and it != t.max_iter:
            t.save_checkpoint(os.path.join(self.output_dir, f"model_{t.iter:07d}.pth"))
        if it >= t.max_iter:
            t.save_checkpoint(os.path.join(self.output_dir, "model_final.pth"))


class PeriodicWriter(HookBase):
    """metrics.json (one JSON object per write) + a log line, every `period` iterations and at the end."""

    def __init__(self, output_dir, period=20):
        self.path = os.path.join(output_dir, "metrics.json")
        self.period = period
        self._t0 = None

    def before_train(self):
        self._t0 = time.perf_counter()

    def after_step(self):
        t = self.trainer
        it = t.iter + 1
        if not t.is_main_process or not ((it % self.period == 0) or it == t.max_iter):
            return
        latest = {k: v for k, (v, i) in t.storage.latest().items()}
        latest["iteration"] = t.iter
        with open(self.path, "a") as f:
            f.write(json.dumps(latest, sort_keys=True) + "\n")
        el = time.perf_counter() - (self._t0 or time.perf_counter())
        losses = "  ".join(f"{k}: {v:.4g}" for k, v in sorted(latest.items()) if "loss" in k)
        logger.info(f"iter: {t.iter}  {losses}  lr: {latest.get('lr', 0):.5g}  elapsed: {el:.1f}s")
