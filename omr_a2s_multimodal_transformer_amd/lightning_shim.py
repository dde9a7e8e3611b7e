"""Minimal LightningModule / Trainer surface so train.py/test.py-style callers run without `lightning`
(absent in this image and on the GPU box; SURVEY.md section 8b lists the attributes and hooks the reference uses:
train.py:115-158, test.py:62-76).  Only what the hot path's callers touch is provided.
"""
from __future__ import annotations

import inspect
from typing import Any, Dict, Iterable, List, Optional

import torch
import torch.nn as nn


class LightningModule(nn.Module):
    def __init__(self) -> None:
        super().__init__()
        self.hparams: Dict[str, Any] = {}
        self.logged_metrics: Dict[str, Any] = {}

    def save_hyperparameters(self) -> None:
        """Capture the calling __init__'s arguments (lightning's save_hyperparameters(), model.py:82,394)."""
        frame = inspect.currentframe().f_back
        args = inspect.getargvalues(frame)
        self.hparams = {k: args.locals[k] for k in args.args if k != "self"}

    def log(self, name: str, value, **kwargs) -> None:
        self.logged_metrics[name] = value

    @property
    def device(self) -> torch.device:
        flat = getattr(self, "_flat", None)
        if flat is not None:
            return flat.device
        try:
            return next(self.parameters()).device
        except StopIteration:
            return torch.device("cpu")

    def freeze(self) -> None:
        for p in self.parameters():
            p.requires_grad_(False)
        self.eval()

    # -- checkpoints: {"state_dict", "hyper_parameters"} like Lightning's .ckpt (split_multimodal_ckpt.py:9-16)
    def save_checkpoint(self, path: str, extra: Optional[Dict[str, Any]] = None) -> None:
        ck = {"state_dict": {k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()}, "hyper_parameters": dict(self.hparams)}
        if extra:
            ck.update(extra)
        torch.save(ck, path)

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path: str, map_location=None, **overrides):
        ck = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
        hp = dict(ck.get("hyper_parameters", {}))
        hp.update(overrides)  # e.g. ytest_i2w override in test.py:62
        model = cls(**hp)
        model.load_state_dict(ck["state_dict"])
        return model


class Trainer:
    """fit/test loops with the hook order Lightning uses for the reference's modules.  One process per GPU: when
    torch.distributed is initialised with more than one rank, fit() attaches the model's gradient reducer (ddp.GradReducer:
    rank-0 parameter broadcast, bucketed all-reduce overlapped with backward), steps the optimizer on the MEAN gradient
    (Lightning-DDP semantics) and the evaluation loops shard their samples over the ranks and all-reduce the metric counts."""

    def __init__(self, max_epochs: int = 1, check_val_every_n_epoch: int = 1, reducer=None, log_every: int = 0, **_ignored):
        self.max_epochs, self.check_val_every_n_epoch, self.reducer, self.log_every = max_epochs, check_val_every_n_epoch, reducer, log_every
        self.callback_metrics: Dict[str, Any] = {}

    @staticmethod
    def _world() -> int:
        import torch.distributed as dist
        return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1

    @staticmethod
    def _rank() -> int:
        import torch.distributed as dist
        return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0

    def _reducer_for(self, model):
        if self.reducer is None and self._world() > 1:
            self.reducer = getattr(model, "_reducer", None) or model.attach_reducer()
        return self.reducer

    def fit(self, model, train_dataloaders: Iterable, val_dataloaders: Optional[Iterable] = None) -> None:
        opt = model.configure_optimizers()
        reducer = self._reducer_for(model)
        for epoch in range(self.max_epochs):
            model.train()
            if hasattr(train_dataloaders, "set_epoch"):
                train_dataloaders.set_epoch(epoch)          # ddp.ShardedLoader: DistributedSampler semantics
            for i, batch in enumerate(train_dataloaders):
                batch = _to_device(batch, model.device)
                opt.zero_grad()
                loss = model.training_step(batch, i)
                loss.backward()
                if reducer is not None:
                    reducer.finish()
                    opt.step(grad_scale=reducer.grad_scale)      # all-reduced SUM -> mean (1/world folded into the Adam kernel)
                else:
                    opt.step()
                if self.log_every and i % self.log_every == 0:
                    print(f"epoch {epoch} step {i} train_loss {float(loss):.4f}")
            if val_dataloaders is not None and (epoch + 1) % self.check_val_every_n_epoch == 0:
                self.callback_metrics.update(self._eval(model, val_dataloaders, "val"))

    def _eval(self, model, loader: Iterable, name: str) -> Dict[str, float]:
        """Validation / test loop (bs = 1 samples, model.py:170-218).  Data parallel: rank r decodes samples r, r + world, ...;
        on_validation_epoch_end all-reduces the edit-distance counts, so every rank returns the single-process metrics."""
        model.eval()
        world, rank = self._world(), self._rank()
        if world > 1:
            self._reducer_for(model)
        with torch.no_grad():
            for i, batch in enumerate(loader):
                if i % world != rank:
                    continue
                model.validation_step(_to_device(batch, model.device), i)
            metrics = model.on_validation_epoch_end(name=name)
        return {f"{name}_{k}": v for k, v in metrics.items()}

    def test(self, model, dataloaders: Iterable) -> List[Dict[str, float]]:
        m = self._eval(model, dataloaders, "test")
        self.callback_metrics.update(m)
        return [m]


def _to_device(batch, device):
    """Float inputs go to the device (non-blocking); integer tensors (tokens, lengths) stay on the host: the token-noise step
    is host logic (model.py:152-160) and the modules upload what the kernels need with pinned, non-blocking copies -- moving
    y_in to the GPU here would force a blocking D2H read (a full stream drain) in every training step."""
    if isinstance(batch, torch.Tensor):
        return batch.to(device, non_blocking=True) if batch.is_floating_point() else batch
    if isinstance(batch, (list, tuple)):
        return type(batch)(_to_device(b, device) for b in batch)
    return batch
