"""Process-wide runtime state of the HIP path: dropout seed stream and flat-parameter management mixin."""
from __future__ import annotations

import random as _pyrandom
from typing import Optional

import torch
import torch.nn as nn

from .params import FlatParams, FusedAdam

# Dropout masks are counter-based (csrc/omr_common.h hash_u32): every dropout site draws a fresh 63-bit
# seed from this stream.  It is separate from Python's global `random`, whose draws the modules consume
# exactly as the reference does (encoder.py:102,160,219; model.py:568-575) so rank-consistent choices
# (dropout position / modality) stay in lock-step across data-parallel ranks.
_seed_stream = _pyrandom.Random(0x5EED)


def seed_dropout(seed: int, rank: int = 0) -> None:
    _seed_stream.seed(seed * 1000003 + rank)


_trace = None      # list of (kind, p, seed, channel_mode) while a dropout trace is being recorded


def next_seed(kind: str = "", p: float = 0.0, channel_mode: bool = False) -> int:
    """Seed of the next dropout site.  `kind` names the site's mask convention for the trace: "nhwc" (MixDropout /
    PositionalEncoding2D on NHWC maps; channel_mode = nn.Dropout2d), "rows" (elementwise over a row-major [rows, d] tensor:
    PE1D, dropout1/2/3, FFN) or "attn" (attention probabilities, omr_attn_dropout_mask)."""
    seed = _seed_stream.getrandbits(63)
    if _trace is not None:
        _trace.append((kind, float(p), seed, bool(channel_mode)))
    return seed


class trace_dropout:
    """Context manager that records every dropout site of the forward passes run inside it, in call order.  The masks are
    pure functions of (seed, element index), so a checker can materialise each site's mask afterwards (omr_dropout on a
    tensor of ones, omr_attn_dropout_mask) and inject it into the CPU oracle (tests/test_dropout_parity_gpu.py)."""

    def __enter__(self):
        global _trace
        self.prev, _trace = _trace, []
        return _trace

    def __exit__(self, *exc):
        global _trace
        _trace = self.prev
        return False


_relu_trace = None     # list of bool tensors (y > 0) while a ReLU trace is being recorded


def note_relu(y: torch.Tensor) -> None:
    """Called by every kernel wrapper whose epilogue applied a ReLU (functional.Conv3x3Fn / LinearFn) with its stored output."""
    if _relu_trace is not None:
        _relu_trace.append(y.detach() > 0)


class trace_relu:
    """Context manager that records, in call order, the ReLU mask (stored output > 0) of every ReLU'd kernel output of the
    forward passes run inside it: NHWC [B,H,W,C] maps for the encoder, [B,T,ff] for the decoder's feed-forward.  A checker
    injects them into oracle.ref_cpu.DropPlan(relu_fn=...) so that the CPU gradient is taken on the same piecewise-linear
    region as the HIP run (tests/test_dropout_parity_gpu.py).  Where a dropout is fused behind the ReLU the mask also
    excludes the dropped elements, whose gradient is zero either way."""

    def __enter__(self):
        global _relu_trace
        self.prev, _relu_trace = _relu_trace, []
        return _relu_trace

    def __exit__(self, *exc):
        global _relu_trace
        _relu_trace = self.prev
        return False


class WgradStream:
    """Side HIP stream for the decoder's weight-gradient GEMMs.  dW = dY^T X of a linear has no consumer inside backward
    and, at d_model = 256, fills a quarter of the CUs for a few tens of microseconds: issued on a second stream it runs
    under the data-gradient kernels of the main stream instead of between them.  Ordering: the side stream waits for the main
    stream's position at every hand-off (its inputs were produced there); the main stream waits for the side stream once,
    when the backward pass ends (autograd engine callback), and before anything else that touches the flat gradient buffer
    (all-reduce buckets, zero_grad, Adam -- they call join()).  A reference to every tensor read on the side stream is held
    until the side-stream kernel that reads it has finished, for two reasons: the caching allocator must not hand the memory
    out early, and autograd accumulates a second incoming gradient IN PLACE into a buffer nobody else
    references (a residual add hands the same gradient tensor to both of its inputs), and that in-place add on the main
    stream would otherwise race with the weight-gradient GEMM still reading the tensor on the side stream (seen as 4-30 %
    errors in the point_conv weight gradients in front of the encoder's residual adds when the side stream lags)."""

    enabled = True
    kinds = {"linear", "depthwise", "cross_kv", "conv", "prepare"}     # what takes the side stream (tests narrow this)
    _side = {}
    _pending = {}          # device index -> the stream that has to wait
    _hold = []             # (event recorded behind the side-stream kernel, the tensors it reads)
    _deferred = []         # weight-gradient problems of linear layers collected for one grouped launch (defer_linear)
    _callback = False      # the end-of-backward engine callback (join) is queued
    defer = True           # collect the linear layers' weight gradients instead of launching them one by one

    @classmethod
    def _release_finished(cls, everything: bool = False) -> None:
        if everything:
            cls._hold.clear()
        else:
            while cls._hold and cls._hold[0][0].query():
                cls._hold.pop(0)

    @classmethod
    def run(cls, kind, fn, *tensors) -> None:
        t0 = tensors[0]
        if not (cls.enabled and kind in cls.kinds and t0.is_cuda):
            fn()
            return
        dev = t0.device.index
        side = cls._side.get(dev)
        if side is None:
            side = cls._side[dev] = torch.cuda.Stream(device=t0.device)
        main = torch.cuda.current_stream(t0.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            fn()
            ev = torch.cuda.Event()
            ev.record(side)
        # no Tensor.record_stream: the references held below keep the operands' memory from being reused until the side-stream
        # kernel has finished (or the main stream has joined the side stream), without the allocator's per-block event traffic
        cls._release_finished()
        cls._hold.append((ev, tensors))
        cls._pending[dev] = main
        cls._at_end_of_backward()

    @classmethod
    def side_stream(cls, device) -> "torch.cuda.Stream":
        side = cls._side.get(device.index)
        if side is None:
            side = cls._side[device.index] = torch.cuda.Stream(device=device)
        return side

    @classmethod
    def prepare(cls, fn, device):
        """Run fn() -- work that depends on NOTHING the main stream has in flight (the attention dropout words of a layer: a
        function of the seed alone) -- on the side stream and make the main stream wait for it.  The host issues a step's
        launches far ahead of the GPU, so the side stream runs fn while the main stream is still busy with earlier kernels
        (the encoder's HBM-bound convolutions, the previous step's backward pass): its vector work hides there instead of
        sitting in front of the consumer.  The result is allocated from the side stream's pool and marked as used by the
        main stream, so the allocator recycles it only after the consumers that were queued when it was freed."""
        if not (cls.enabled and "prepare" in cls.kinds and device.type == "cuda"):
            return fn()
        side, main = cls.side_stream(device), torch.cuda.current_stream(device)
        with torch.cuda.stream(side):
            out = fn()
            ev = torch.cuda.Event()
            ev.record(side)
        main.wait_event(ev)
        out.record_stream(main)
        return out

    @classmethod
    def _at_end_of_backward(cls) -> None:
        """Queue join() to run when the backward pass ends (once per pass)."""
        if cls._callback:
            return
        try:
            torch.autograd.Variable._execution_engine.queue_callback(cls.join)
            cls._callback = True
        except RuntimeError:          # not inside a backward pass (direct call from a test): order it right away
            cls.join()

    @classmethod
    def defer_linear(cls, dy2d, x2d, dw2d, db, group=None) -> None:
        """dW = dY^T X (+ db) of a linear layer.  At d_model = 256 one such product is a handful of workgroups -- a latency
        chain that cannot fill the chip -- so the products of a backward pass are collected here and launched TOGETHER
        (omr_linear_wgrad_grouped: ~2 000 workgroups in one grid) when the pass reaches a flush point: a gradient-bucket
        boundary (ddp.GradBoundary) or its end.  The queue holds the operands alive until then."""
        from . import kernels as K
        if not (cls.defer and dy2d.is_cuda):
            cls.run("linear", lambda: K.linear_wgrad_grouped([(dy2d, x2d, dw2d, db, group)]), dy2d, x2d)
            return
        cls._deferred.append((dy2d, x2d, dw2d, db, group))
        cls._at_end_of_backward()

    @classmethod
    def flush(cls) -> None:
        """Launch what defer_linear collected (on the side stream when enabled for "linear")."""
        if not cls._deferred:
            return
        from . import kernels as K
        probs, cls._deferred = cls._deferred, []
        for dt in {p[0].dtype for p in probs}:
            same = [p for p in probs if p[0].dtype == dt]
            cls.run("linear", lambda: K.linear_wgrad_grouped(same), *[t for p in same for t in p[:2]])

    @classmethod
    def join(cls) -> None:
        cls._callback = False
        cls.flush()
        for dev, main in list(cls._pending.items()):
            main.wait_stream(cls._side[dev])
        cls._pending.clear()
        cls._release_finished(everything=True)      # the main stream is ordered behind every side-stream read now


    @classmethod
    def reset(cls) -> None:
        """Start of a step (FlatParams.zero_grad): nothing collected may survive into it.  After a backward pass that RAISED
        (an OOM, GradReducer's second-backward error) the engine never ran the queued join(): `_callback` would stay set --
        so the next pass would not queue its own join -- and the stale collected problems would be launched into the fresh
        gradient buffer.  Drop them unlaunched, then order the streams like join()."""
        cls._deferred = []
        cls.join()


class FlatModuleMixin:
    """Adds flat-buffer parameter storage (params.py) to a top-level nn.Module."""

    _flat: Optional[FlatParams] = None

    def compute_dtype(self) -> torch.dtype:
        return self._flat.compute_dtype if self._flat is not None else torch.float32

    def flatten_parameters(self, compute_dtype: Optional[torch.dtype] = None, device=None) -> FlatParams:
        """(Re)build the flat master/grad/compute buffers on `device` (default cuda:current) and re-point
        every nn.Parameter at its view.  Call after .load_state_dict on CPU, or to switch compute dtype."""
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if compute_dtype is None:
            compute_dtype = self._flat.compute_dtype if self._flat is not None else getattr(self, "_default_compute_dtype", torch.float32)
        named = [(n, p) for n, p in self.named_parameters()]
        self._flat = FlatParams(named, torch.device(device), compute_dtype)
        for mod in self.modules():
            for name, buf in list(mod._buffers.items()):
                if buf is not None and buf.device != self._flat.device:
                    mod._buffers[name] = buf.to(self._flat.device)
        return self._flat

    def ensure_flat(self) -> FlatParams:
        if self._flat is None:
            self.flatten_parameters()
        return self._flat

    def sync_compute_weights(self) -> None:
        if self._flat is not None:
            self._flat.sync_lowp()

    def zero_grad(self, set_to_none: bool = False) -> None:  # type: ignore[override]
        if self._flat is not None:
            self._flat.zero_grad()
        else:
            nn.Module.zero_grad(self, set_to_none)

    _touched = None     # top-level sub-modules whose parameters took part in the last forward (None = all); see FusedAdam

    def make_optimizer(self, lr: float = 1e-4) -> FusedAdam:
        return FusedAdam(self.ensure_flat(), lr=lr, touched_fn=lambda: self._touched)

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):  # type: ignore[override]
        out = nn.Module.load_state_dict(self, state_dict, strict=strict)
        self.sync_compute_weights()
        return out
