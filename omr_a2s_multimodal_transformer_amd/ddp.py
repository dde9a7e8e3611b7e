"""Data-parallel gradient averaging over RCCL/xGMI (backend "nccl" on ROCm) or gloo (CPU tests).

The reference has no explicit DDP code: multi-GPU training happens only through Lightning's default
DistributedDataParallel.  Here one process drives one GPU and gradients live in ONE flat
fp32 buffer (params.py), so the exchange step is an all-reduce(sum) of a few large contiguous buckets followed
by a 1/world scale that is folded into the fused Adam kernel (grad_scale) -- no per-tensor work at all.

Overlap with backward: autograd "bucket boundary" nodes (GradBoundary) are placed in the forward graph where
all parameters of a bucket have finished accumulating (decoder after the memory gradient is complete, encoder
stages after their block).  When backward reaches a boundary, the bucket's all-reduce is enqueued on a side
HIP stream behind an event recorded on the compute stream, so RCCL traffic over xGMI runs under the remaining
backward kernels.  finish() makes the compute stream wait for all outstanding buckets before Adam.

Unused parameters (MultimodalTransformer drops a modality in ~20 % of the steps, model.py:510-519): their slice of
the flat buffer simply stays zero and is reduced like any other -- every rank reduces the same buckets every
step, so there is nothing to negotiate and nothing can hang.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
from torch.autograd import Function

from .runtime import WgradStream


class GradReducer:
    def __init__(self, flat, process_group=None, buckets: Optional[Sequence[Tuple[int, int]]] = None, async_stream: bool = True):
        self.flat = flat
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.buckets: List[Tuple[int, int]] = list(buckets) if buckets else [(0, flat.total)]
        self.done = [False] * len(self.buckets)
        self._backwards = 0               # GradBoundary crossings since the last finish()
        self.handles: List = []
        self.use_stream = async_stream and flat.grad.is_cuda
        self.stream = torch.cuda.Stream(device=flat.device) if self.use_stream else None
        self.shared_rng = None            # random.Random every rank seeds with rank 0's value (broadcast_state): rank-proof host decisions

    def broadcast_state(self) -> None:
        """Rank 0's master parameters (and Adam moments, if an optimizer exists already) to every rank, then refresh the bf16
        compute copy: what DistributedDataParallel does at construction, so replicas are equal whatever each rank seeded or
        loaded."""
        if self.world == 1:
            return
        src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
        for t in (self.flat.master, self.flat.exp_avg, self.flat.exp_avg_sq):
            if t is not None:
                dist.broadcast(t, src=src, group=self.group)
        self.flat.sync_lowp()
        # Host decisions that select WHICH parameters a step touches (MultimodalTransformer's modality drop, model.py:561-575)
        # must come out the same on every rank: the all-reduced buckets carry every rank's gradients, so a rank that skipped
        # `audio_encoder` in its optimizer while another rank trained it would diverge silently.  The reference's ranks agree
        # only because train.py:17 seeds every process alike; here the ranks draw such decisions from ONE generator seeded
        # with rank 0's value -- a one-time broadcast, no per-step collective, no host synchronisation.
        import random
        seed = torch.tensor([random.getrandbits(62)], dtype=torch.int64, device=self.flat.master.device)
        dist.broadcast(seed, src=src, group=self.group)
        self.shared_rng = random.Random(int(seed.item()))

    @property
    def grad_scale(self) -> float:
        """Factor that turns the all-reduced SUM into Lightning-DDP's mean (applied inside omr_adam)."""
        return 1.0 / self.world

    def reduce_bucket(self, i: int) -> None:
        """Enqueue bucket i's all-reduce (idempotent per step: a second backward before finish() -- gradient accumulation --
        would add to a bucket that is already on the wire, so it is refused rather than silently mis-reduced)."""
        if self.world == 1:
            return
        if self.done[i]:
            if self._backwards > 1:
                raise RuntimeError("GradReducer: a second backward reached an already reduced bucket; call finish() (and the optimizer) "
                                   "after every backward -- gradient accumulation across backwards is not supported")
            return
        self.done[i] = True
        b, e = self.buckets[i]
        view = self.flat.grad[b:e]
        if self.use_stream:
            # Weight gradients collected / issued on the side stream belong to the bucket too: launch what is collected (one
            # grouped GEMM) and let the COMMUNICATION stream wait for the side stream -- the compute stream goes straight on
            # with the encoder's backward pass and only joins the side stream when the pass ends.
            WgradStream.flush()
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.flat.device))
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ev)
                side = WgradStream._side.get(self.flat.device.index)
                if side is not None:
                    self.stream.wait_stream(side)
                dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
        else:
            WgradStream.join()
            self.handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self) -> None:
        """Reduce whatever has not been reduced yet and order the optimizer after all buckets."""
        for i in range(len(self.buckets)):
            self.reduce_bucket(i)
        if self.use_stream and self.world > 1:
            torch.cuda.current_stream(self.flat.device).wait_stream(self.stream)
        for h in self.handles:
            h.wait()
        self.handles.clear()
        self.done = [False] * len(self.buckets)
        self._backwards = 0


class GradBoundary(Function):
    """Identity on one or more tensors in forward; in backward -- which autograd runs once the gradients of ALL of them are
    known -- signals that every gradient produced AFTER this point of the forward (i.e. earlier in backward) is final, and
    launches the given buckets' all-reduce.  Several tensors: the two encoder memories of MultimodalTransformer, behind
    which the mixer and the decoder lie; a memory the step did not use simply has no gradient."""

    @staticmethod
    def forward(ctx, reducer, bucket_ids, *xs):
        ctx.reducer, ctx.bucket_ids = reducer, bucket_ids
        ctx.set_materialize_grads(False)
        outs = tuple(x.view_as(x) for x in xs)
        return outs[0] if len(outs) == 1 else outs

    @staticmethod
    def backward(ctx, *gs):
        if ctx.reducer is not None:
            ctx.reducer._backwards += 1
            for i in ctx.bucket_ids:
                ctx.reducer.reduce_bucket(i)
        else:
            WgradStream.flush()        # single GPU: the decoder's collected weight gradients start now, under the encoder's backward pass
        return (None, None) + tuple(gs)


def shard_indices(n_samples: int, rank: int, world: int, epoch: int = 0, shuffle: bool = True, seed: int = 0) -> List[int]:
    """DistributedSampler semantics (what Lightning injects): shared-seed shuffle per epoch,
    pad by wrap-around to a multiple of world, rank r takes indices r::world."""
    if shuffle:
        g = torch.Generator()
        g.manual_seed(seed + epoch)
        idx = torch.randperm(n_samples, generator=g).tolist()
    else:
        idx = list(range(n_samples))
    total = (n_samples + world - 1) // world * world
    idx += idx[: total - n_samples]
    return idx[rank:total:world]


class ShardedLoader:
    """What Lightning's DDP does to the reference's DataLoaders: a DistributedSampler over the SAMPLES
    (shard_indices: shared-seed shuffle per epoch, wrap-around padding, rank r takes r::world) followed by batching and the
    collate function (preprocessing.py:85-144).  `dataset` is any sequence of samples; call set_epoch(e) before each epoch
    (Trainer.fit does).  With world == 1 it is a plain (optionally shuffled) batched loader."""

    def __init__(self, dataset, batch_size: int, collate_fn, rank: Optional[int] = None, world: Optional[int] = None, shuffle: bool = True,
                 seed: int = 0):
        init = dist.is_available() and dist.is_initialized()
        self.dataset, self.batch_size, self.collate_fn, self.shuffle, self.seed = dataset, batch_size, collate_fn, shuffle, seed
        self.rank = rank if rank is not None else (dist.get_rank() if init else 0)
        self.world = world if world is not None else (dist.get_world_size() if init else 1)
        self.epoch = 0

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch

    def indices(self) -> List[int]:
        return shard_indices(len(self.dataset), self.rank, self.world, self.epoch, self.shuffle, self.seed)

    def __len__(self) -> int:
        return (len(self.indices()) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        idx = self.indices()
        for i in range(0, len(idx), self.batch_size):
            yield self.collate_fn([self.dataset[j] for j in idx[i:i + self.batch_size]])
