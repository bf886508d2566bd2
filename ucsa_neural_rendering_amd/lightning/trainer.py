"""A thin stand-in for the PyTorch-Lightning 1.6 pieces the reference's entry
point uses (scripts/train_joint.py:146-181): ``Trainer(max_epochs=...)`` with
``fit / test / validate / predict`` and ``seed_everything``.  PyTorch-Lightning
is not installed on the MI355X image; Trainer internals are out of
scope, only the call order and hook names the LightningModule relies on are
kept.  One process per GPU; under ``torch.distributed`` the NeRF gradient
all-reduce lives in the module (``dist.allreduce_grads_``), not here.
"""
from __future__ import annotations

import json
import os
import random
import time

import numpy as np
import torch


def seed_everything(seed: int):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    return seed


class JsonlLogger:
    """Plain JSONL metrics logger (same metric names as the reference's
    self.log calls; WandB is out of scope).

    Values may be device tensors: they are kept as such and converted in
    batches (``flush``: one device read-back and one file append for up to
    ``flush_every`` records), so logging a loss does not synchronise the
    training step that produced it."""

    def __init__(self, save_dir=None, flush_every=512):
        self.path = os.path.join(save_dir, "metrics.jsonl") if save_dir else None
        self._history = []
        self._pending = []
        self.flush_every = int(flush_every)

    def log(self, name, value, step=None):
        if torch.is_tensor(value):
            value = value.detach()
        self._pending.append((name, value, step, time.time()))
        if len(self._pending) >= self.flush_every:
            self.flush()

    def flush(self):
        pend, self._pending = self._pending, []
        if not pend:
            return
        tens = [(i, v) for i, (_, v, _, _) in enumerate(pend) if torch.is_tensor(v)]
        vals = [None if torch.is_tensor(v) else float(v) for (_, v, _, _) in pend]
        by_dev = {}
        for i, v in tens:
            by_dev.setdefault(v.device, []).append((i, v))
        for items in by_dev.values():
            flat = torch.stack([v.reshape(()).double() for _, v in items]).tolist()
            for (i, _), f in zip(items, flat):
                vals[i] = f
        recs = [{"name": n, "value": vals[i], "step": st, "time": tm}
                for i, (n, _, st, tm) in enumerate(pend)]
        self._history.extend(recs)
        if self.path:
            with open(self.path, "a") as f:
                f.write("".join(json.dumps(r) + "\n" for r in recs))

    @property
    def history(self):
        self.flush()
        return self._history

    def log_hyperparams(self, params):
        pass


class Trainer:

    def __init__(self, max_epochs=1, default_root_dir=None, logger=None,
                 callbacks=None, check_val_every_n_epoch=1, device=None,
                 limit_batches=None, prefetch=0, **unused):
        self.max_epochs = int(max_epochs)
        self.root = default_root_dir
        self.logger = logger or JsonlLogger(default_root_dir)
        self.check_val_every_n_epoch = check_val_every_n_epoch
        self.device = torch.device(device or (
            "cuda" if torch.cuda.is_available() else "cpu"))
        self.limit_batches = limit_batches
        # `trainer: {prefetch: N}`: the next N batches are produced (PNG decode,
        # device rays, replay augmentation, host -> device copies) by a background
        # thread while the current step runs -- the role of the reference's
        # DataLoader workers (cfg `num_workers`), which cannot be forked here: the
        # datasets work on the device.  Off by default: the loader's random draws
        # then interleave with the step's on the shared generator (as with worker
        # processes, runs are no longer reproducible draw for draw).
        self.prefetch = int(prefetch or 0)
        self.current_epoch = 0
        self.global_step = 0

    # -- helpers -------------------------------------------------------------
    def _attach(self, model):
        model.trainer = self
        model.to(self.device)
        if not getattr(model, "_optimizers", None):
            model._optimizers = model.configure_optimizers()

    # per-item scalars the module only ever reads as Python numbers
    # (intrinsics -> float, H/W -> int, flags -> bool): left on the host, a
    # device copy would cost one read-back per use and training step
    _HOST_KEYS = frozenset(["intrinsics", "H", "W", "one_m_to_scene_uom",
                            "viewpoint_is_novel", "from_old_scene",
                            "current_index"])

    def _to_device(self, obj):
        if torch.is_tensor(obj):
            return obj.to(self.device, non_blocking=True)
        if isinstance(obj, dict):
            return {k: (v.cpu() if torch.is_tensor(v) and k in self._HOST_KEYS
                        else v if k in self._HOST_KEYS else self._to_device(v))
                    for k, v in obj.items()}
        if isinstance(obj, (list, tuple)):
            return type(obj)(self._to_device(v) for v in obj)
        return obj

    def _loaders(self, dl):
        return dl if isinstance(dl, (list, tuple)) else [dl]

    def _batches_serial(self, loader, limited=True):
        for i, b in enumerate(loader):
            if limited and self.limit_batches is not None and i >= self.limit_batches:
                break
            yield i, self._to_device(b)

    def _batches(self, loader, limited=True):
        if self.prefetch <= 0:
            yield from self._batches_serial(loader, limited)
            return
        import queue
        import threading
        q, done, failed = queue.Queue(maxsize=self.prefetch), object(), []
        stop = threading.Event()

        def work():
            try:
                if self.device.type == "cuda":
                    torch.cuda.set_device(self.device)     # thread-local in HIP
                for item in self._batches_serial(loader, limited):
                    while not stop.is_set():
                        try:
                            q.put(item, timeout=0.2)
                            break
                        except queue.Full:
                            continue
                    if stop.is_set():
                        return
            except BaseException as e:  # noqa: BLE001 -- re-raised in the consumer
                failed.append(e)
            finally:
                while True:
                    try:
                        q.put(done, timeout=0.2)
                        break
                    except queue.Full:
                        if stop.is_set():
                            break

        t = threading.Thread(target=work, name="ucsa-prefetch", daemon=True)
        t.start()
        try:
            while True:
                item = q.get()
                if item is done:
                    break
                yield item
        finally:
            stop.set()
        if failed:
            raise failed[0]

    def _flush_logs(self):
        flush = getattr(self.logger, "flush", None)
        if flush is not None:
            flush()

    # -- loops ---------------------------------------------------------------
    def fit(self, model, train_dataloaders=None, val_dataloaders=None):
        self._attach(model)
        for epoch in range(self.max_epochs):
            self.current_epoch = epoch
            model.train()
            model.on_train_epoch_start()
            set_epoch = getattr(getattr(train_dataloaders, "sampler", None),
                                "set_epoch", None)
            if set_epoch is not None:  # DistributedSampler: reshuffle per epoch
                set_epoch(epoch)
            for i, batch in self._batches(train_dataloaders):
                model.training_step(batch, i)
                self.global_step += 1
            model.on_train_epoch_end()
            self._flush_logs()
            if val_dataloaders is not None and (
                    epoch + 1) % self.check_val_every_n_epoch == 0:
                self.validate(model, dataloaders=val_dataloaders)

    def _eval_loop(self, model, dataloaders, kind):
        self._attach(model)
        model.eval()
        getattr(model, f"on_{kind}_epoch_start")()
        outs = []
        with torch.no_grad():
            # the predict pass writes the files the NEXT stage replays: it always
            # covers every frame, whatever `limit_batches` (a smoke-run knob) says
            for li, loader in enumerate(self._loaders(dataloaders)):
                for i, batch in self._batches(loader, limited=kind != "predict"):
                    outs.append(getattr(model, f"{kind}_step")(batch, i, li))
        res = getattr(model, f"on_{kind}_epoch_end")()
        self._flush_logs()
        model.train()
        return res if res is not None else outs

    def validate(self, model, dataloaders=None):
        return self._eval_loop(model, dataloaders, "validation")

    def test(self, model, dataloaders=None):
        return self._eval_loop(model, dataloaders, "test")

    def predict(self, model, dataloaders=None):
        return self._eval_loop(model, dataloaders, "predict")
