"""N>1 path on CPU: world_size-2 gloo processes exercise the flat-bucket gradient reducer, the backward-triggered
bucket boundary and the DistributedSampler-equivalent sharding (no HIP kernels involved: pure host logic)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from omr_a2s_multimodal_transformer_amd.ddp import GradBoundary, GradReducer, shard_indices
from omr_a2s_multimodal_transformer_amd.params import FlatParams


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class Tiny(nn.Module):
    def __init__(self):
        super().__init__()
        self.enc = nn.Linear(5, 7)
        self.dec = nn.Linear(7, 3)
        self.conv = nn.Conv2d(2, 4, 3)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    m = Tiny()
    flat = FlatParams(list(m.named_parameters()), torch.device("cpu"), torch.float32)
    # parameters are views of the flat buffer and keep their logical shapes; conv weights are stored channels-last
    assert m.conv.weight.shape == (4, 2, 3, 3) and m.conv.weight.omr_phys.shape == (4, 3, 3, 2)
    assert m.conv.weight.data_ptr() == m.conv.weight.omr_phys.data_ptr()
    enc_b, dec_b = flat.slice_of(["enc.weight", "enc.bias"]), flat.slice_of(["dec.weight", "dec.bias", "conv.weight", "conv.bias"])
    red = GradReducer(flat, None, [enc_b, dec_b])
    assert red.grad_scale == 1.0 / world

    # 1) plain finish(): every bucket summed over ranks
    flat.grad.fill_(float(rank + 1))
    red.finish()
    assert torch.allclose(flat.grad, torch.full_like(flat.grad, float(sum(range(1, world + 1)))))
    assert torch.equal(m.enc.weight.grad, flat.grad[: 35].view(7, 5))

    # 2) boundary fired from backward reduces ONLY its bucket; finish() does the rest exactly once
    flat.grad.fill_(float(rank + 1))
    x = torch.ones(2, requires_grad=True)
    y = GradBoundary.apply(red, (1,), x * 2.0)
    y.sum().backward()
    for h in red.handles:
        h.wait()
    total = float(sum(range(1, world + 1)))
    assert torch.allclose(flat.grad[dec_b[0]:dec_b[1]], torch.full((dec_b[1] - dec_b[0],), total))
    assert torch.allclose(flat.grad[enc_b[0]:enc_b[1]], torch.full((enc_b[1] - enc_b[0],), float(rank + 1)))
    red.finish()
    assert torch.allclose(flat.grad, torch.full_like(flat.grad, total))

    # 3) an "unused" bucket (all zeros on every rank) reduces to zeros: no negotiation, no hang
    flat.zero_grad()
    flat.grad[dec_b[0]:dec_b[1]].fill_(1.0)
    red.finish()
    assert float(flat.grad[enc_b[0]:enc_b[1]].abs().sum()) == 0.0
    dist.barrier()
    dist.destroy_process_group()
    q.put(rank)


def test_grad_reducer_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(q.get() for _ in range(world)) == [0, 1]


def test_shard_indices_matches_distributed_sampler():
    from torch.utils.data.distributed import DistributedSampler
    data = list(range(23))
    for world in (2, 8):
        for epoch in (0, 3):
            got = [shard_indices(len(data), r, world, epoch=epoch, shuffle=True, seed=0) for r in range(world)]
            for r in range(world):
                s = DistributedSampler(data, num_replicas=world, rank=r, shuffle=True, seed=0)
                s.set_epoch(epoch)
                assert list(s) == got[r]
            assert sorted(set(sum(got, []))) == data
