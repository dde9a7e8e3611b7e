"""Flat parameter storage sized for one HBM-resident model replica + fused Adam.

All parameters of a model live in ONE fp32 buffer (master weights) with parallel flat buffers for
gradients, Adam moments and (bf16 mode) the bf16 compute copy.  nn.Parameters are views into the master
buffer that keep the reference's logical shapes (state-dict contract) while the memory
is laid out for the kernels: conv weights [Cout,Cin,kh,kw] are stored [Cout,kh,kw,Cin] (the logical
tensor is the channels_last-strided permutation), so MFMA B-fragments are 16-byte k-contiguous loads.
One flat buffer means: Adam is a single kernel launch (omr_adam), zero_grad is one memset and the
data-parallel gradient all-reduce runs over a few large contiguous buckets (ddp.py).
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Tuple

import torch
import torch.nn as nn

from . import kernels as K

ALIGN = 64  # elements; keeps every view 256-byte aligned in fp32 and 128-byte aligned in bf16


def _phys_perm(ndim: int) -> Optional[Tuple[int, ...]]:
    """logical dims -> physical order: channel dim 1 moves last for conv weights (ndim >= 3)."""
    if ndim < 3:
        return None
    return (0,) + tuple(range(2, ndim)) + (1,)


def _inverse(perm: Tuple[int, ...]) -> Tuple[int, ...]:
    inv = [0] * len(perm)
    for i, p in enumerate(perm):
        inv[p] = i
    return tuple(inv)


# Parameters that are placed back to back (in layer order) so that one GEMM can run over all of them through a row-group
# view (omr_gemm row_group_*): the packed cross-attention in_proj matrices / biases of all decoder layers, whose K|V rows
# project the SAME encoder memory in every layer (functional.FusedCrossKVFn).  Placement is a memory-layout choice only:
# names, shapes and state-dict order are untouched.
_GROUPED_SUFFIXES = ("multihead_attn.in_proj_weight", "multihead_attn.in_proj_bias")


def _placement_order(named_params: List[Tuple[str, nn.Parameter]]) -> List[Tuple[str, nn.Parameter]]:
    order: List[Tuple[str, nn.Parameter]] = []
    done = set()
    for n, p in named_params:
        if n in done:
            continue
        suffix = next((s for s in _GROUPED_SUFFIXES if n.endswith(s)), None)
        if suffix is None:
            order.append((n, p))
            done.add(n)
            continue
        for n2, p2 in named_params:             # pull every member of the family here, in model (layer) order
            if n2.endswith(suffix) and n2 not in done and p2.shape == p.shape:
                order.append((n2, p2))
                done.add(n2)
    return order


class FlatParams:
    def __init__(self, named_params: List[Tuple[str, nn.Parameter]], device: torch.device, compute_dtype: torch.dtype):
        self.names = [n for n, _ in named_params]
        self.params = [p for _, p in named_params]
        self.device = device
        self.compute_dtype = compute_dtype
        self.offsets: Dict[str, Tuple[int, int]] = {}
        off = 0
        for n, p in _placement_order(named_params):
            self.offsets[n] = (off, p.numel())
            off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.total = off
        self.master = torch.zeros(off, dtype=torch.float32, device=device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=device)
        self.lowp = torch.zeros(off, dtype=torch.bfloat16, device=device) if compute_dtype == torch.bfloat16 else None
        self.exp_avg: Optional[torch.Tensor] = None
        self.exp_avg_sq: Optional[torch.Tensor] = None
        with torch.no_grad():
            for n, p in named_params:
                o, cnt = self.offsets[n]
                perm = _phys_perm(p.dim())
                phys_shape = tuple(p.shape) if perm is None else tuple(p.shape[i] for i in perm)

                def views(buf):
                    phys = buf[o:o + cnt].view(phys_shape)
                    logical = phys if perm is None else phys.permute(_inverse(perm))
                    return phys, logical

                phys, logical = views(self.master)
                logical.copy_(p.detach().to(device=device, dtype=torch.float32))
                p.data = logical
                p.omr_phys = phys
                gphys, glogical = views(self.grad)
                p.omr_grad = gphys
                p.grad = glogical
                p.omr_lowp = views(self.lowp)[0] if self.lowp is not None else None
        # Data-gradient weights of the dense 3x3 convs ([CIN,3,3,COUT], taps mirrored: the transposed conv's operand) in the
        # compute dtype: refreshed once per optimizer step by ONE launch (refresh_flips) instead of one re-layout launch in front
        # of every data gradient of the backward pass.  A per-parameter (torch version, update count) stamp guards manual edits.
        convs = [p for p in self.params if p.dim() == 4 and tuple(p.shape[2:]) == (3, 3) and p.shape[1] > 1]
        self.flip = torch.empty(sum(p.numel() for p in convs), dtype=compute_dtype, device=device)
        self.updates = 0                 # weight updates that bypass torch's version counter (omr_adam writes through raw pointers)
        self._flip_table = None
        self._flip_params = convs
        pairs, off = [], 0
        for p in convs:
            cout, cin = p.shape[0], p.shape[1]
            p.omr_flip = self.flip[off:off + p.numel()].view(cin, 3, 3, cout)
            p.omr_flat = self
            p.omr_flip_stamp = None
            off += p.numel()
            pairs.append((p.omr_phys if compute_dtype == torch.float32 else p.omr_lowp, p.omr_flip))
        self._flip_pairs = pairs
        self.sync_lowp()

    def sync_lowp(self) -> None:
        """Refresh the bf16 compute copy from the fp32 masters (after load_state_dict / manual edits)."""
        if self.lowp is not None:
            K.cast(self.master, torch.bfloat16, out=self.lowp)
        self.refresh_flips()

    def refresh_flips(self) -> None:
        """Re-lay the data-gradient weights of all dense 3x3 convs from the current compute weights (one launch)."""
        if not self._flip_pairs or not self.master.is_cuda:
            return
        if self._flip_table is None:
            self._flip_table = K.conv3x3_weight_flip_table(self._flip_pairs)
        K.conv3x3_weight_flip_grouped(self._flip_table)
        for p in self._flip_params:
            p.omr_flip_stamp = (p._version, self.updates)

    def flipped(self, p: nn.Parameter, dtype: torch.dtype) -> torch.Tensor:
        """Data-gradient weights of conv parameter p.  Current by construction after FusedAdam.step / sync_lowp; an in-place
        edit of the parameter through torch (copy_, a sub-module's load_state_dict) bumps its version counter and triggers a
        refresh here.  (In bf16 mode such an edit needs sync_compute_weights() anyway: the forward reads the bf16 copy.)"""
        if dtype != self.compute_dtype:
            return K.conv3x3_weight_flip(p.omr_phys if dtype == torch.float32 else p.omr_lowp)
        if p.omr_flip_stamp != (p._version, self.updates):
            self.refresh_flips()
        return p.omr_flip

    def zero_grad(self) -> None:
        from .runtime import WgradStream
        WgradStream.reset()         # also drops what a failed backward pass left collected
        from .functional import clear_pending_norm
        clear_pending_norm()
        self.grad.zero_()

    def slice_of(self, names: Iterable[str]) -> Tuple[int, int]:
        """[begin, end) element range of the flat buffers that covers the given parameters."""
        offs = [self.offsets[n] for n in names]
        b = min(o for o, _ in offs)
        e = max((o + c + ALIGN - 1) // ALIGN * ALIGN for o, c in offs)
        return b, e

    def module_ranges(self) -> "Dict[str, Tuple[int, int]]":
        """{top-level sub-module name: [begin, end)}: the contiguous range of the flat buffers each top-level sub-module
        (encoder, image_encoder, audio_encoder, decoder, cross_attn) owns.  Placement keeps every sub-module contiguous (the
        grouped cross-attention in_proj family lies inside `decoder`); checked here."""
        groups: Dict[str, List[str]] = {}
        for n in self.names:
            groups.setdefault(n.split(".", 1)[0], []).append(n)
        ranges = {g: self.slice_of(ns) for g, ns in groups.items()}
        spans = sorted(ranges.values())
        assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])), "top-level sub-modules overlap in the flat buffer"
        return ranges


class FusedAdam:
    """torch.optim.Adam(lr=1e-4, amsgrad=False) semantics (model.py:134-139,475-483) as one kernel launch per top-level
    sub-module over the flat buffers (one launch in the common case that every sub-module has gradients).
    Interface subset of torch.optim.Optimizer: step / zero_grad / param_groups / state_dict.

    Parameters WITHOUT a gradient are skipped like torch.optim.Adam skips `p.grad is None` (Lightning zeroes with
    set_to_none): no update from stale momentum, no moment decay, no step count.  In MultimodalTransformer the modality-drop
    steps (model.py:510-519) leave one encoder and cross_attn without gradients; the model reports the sub-modules that took
    part in the step through `touched_fn` and each sub-module keeps its own step count for the bias correction.  (Under data
    parallelism every rank takes the same branch BY CONSTRUCTION: the modality decision is drawn from a generator all ranks
    seed with rank 0's value -- model.MultimodalTransformer._draw_modality, ddp.GradReducer.broadcast_state -- so all ranks
    skip the same slices; tests/test_ddp_cpu.py runs it with ranks seeded differently.)"""

    def __init__(self, flat: FlatParams, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8, touched_fn=None):
        self.flat = flat
        self.param_groups = [dict(params=flat.params, lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False)]
        self.ranges = flat.module_ranges()
        self.steps = {g: 0 for g in self.ranges}
        self.touched_fn = touched_fn
        if flat.exp_avg is None:
            flat.exp_avg = torch.zeros_like(flat.master)
            flat.exp_avg_sq = torch.zeros_like(flat.master)

    @property
    def step_count(self) -> int:
        return max(self.steps.values())

    def zero_grad(self, set_to_none: bool = False) -> None:
        self.flat.zero_grad()

    def step(self, grad_scale: float = 1.0, touched=None) -> None:
        """touched: names of the top-level sub-modules that received gradients this step (None = all; default: ask the model)."""
        g = self.param_groups[0]
        f = self.flat
        if touched is None and self.touched_fn is not None:
            touched = self.touched_fn()
        names = list(self.ranges) if touched is None else [n for n in self.ranges if n in set(touched)]
        for n in names:
            self.steps[n] += 1
        from .runtime import WgradStream
        WgradStream.join()
        # merge neighbouring ranges that share a step count: one launch over the whole buffer in the common case
        runs: List[List[int]] = []
        for n in sorted(names, key=lambda k: self.ranges[k][0]):
            b, e = self.ranges[n]
            if runs and runs[-1][1] == b and runs[-1][2] == self.steps[n]:
                runs[-1][1] = e
            else:
                runs.append([b, e, self.steps[n]])
        for b, e, st in runs:
            K.adam_step(f.master[b:e], f.grad[b:e], f.exp_avg[b:e], f.exp_avg_sq[b:e], st, g["lr"], g["betas"], g["eps"], grad_scale,
                        p_lowp=None if f.lowp is None else f.lowp[b:e])
        f.updates += 1
        f.refresh_flips()           # the next backward pass finds its data-gradient weights ready

    def state_dict(self):
        return dict(step=self.step_count, steps=dict(self.steps), exp_avg=self.flat.exp_avg, exp_avg_sq=self.flat.exp_avg_sq, param_groups=[
            {k: v for k, v in self.param_groups[0].items() if k != "params"}])

    def load_state_dict(self, sd) -> None:
        self.steps = {g: int(sd.get("steps", {}).get(g, sd["step"])) for g in self.ranges}
        self.flat.exp_avg.copy_(sd["exp_avg"])
        self.flat.exp_avg_sq.copy_(sd["exp_avg_sq"])
