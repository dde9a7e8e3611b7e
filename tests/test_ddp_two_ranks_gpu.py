"""The data-parallel step with TWO real ranks on the GPU.  The one-GPU box cannot host two RCCL ranks (RCCL refuses duplicate
devices), so the ranks share cuda:0 and the collectives run over gloo on CUDA tensors -- everything else is the production
path: flat-buffer buckets, the all-reduce issued from the memory boundary in mid-backward on the communication stream, the
side-stream weight gradients and dropout words, rank-0 broadcast, 1/world folded into Adam, and (multimodal) the modality
decision from the shared generator.  Checked: the ranks end with bit-identical master parameters / Adam moments although their
data, dropout seeds and Python `random` streams differ, the step changes the parameters, the loss is finite.
(The RCCL transport itself: tests/test_rccl_single_rank_gpu.py and the driver's scaling run.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, kind, q):
    import random
    from omr_a2s_multimodal_transformer_amd import synthetic as syn
    from omr_a2s_multimodal_transformer_amd.config import ModelConfig
    from omr_a2s_multimodal_transformer_amd.model import MultimodalTransformer, Transformer
    from omr_a2s_multimodal_transformer_amd.runtime import seed_dropout
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    V, T = 60, 20
    w2i, i2w = syn.make_vocab(V)
    cfg = ModelConfig(num_layers=2, compute_dtype="bf16")
    torch.manual_seed(100 + rank)                       # ranks start from DIFFERENT parameters: the reducer broadcasts rank 0's
    random.seed(1000 + 17 * rank)                       # ... and draw different Python streams
    if kind == "multimodal":
        m = MultimodalTransformer(64, 160, 195, 96, T, w2i, i2w, mixer_type="concat", teacher_forcing_prob=0.2, teacher_forcing_modality_prob=0.6,
                                  config=cfg)
        xi, xli, y_in, y_out = syn.synthetic_unimodal_batch(3, 64, 160, T, V, w2i["<sos>"], w2i["<eos>"], seed=10 + rank)
        xa, xla, _, _ = syn.synthetic_unimodal_batch(3, 195, 96, T, V, w2i["<sos>"], w2i["<eos>"], seed=20 + rank, pad_value=0.0)
        batch = (xi.to(dev), xli, xa.to(dev), xla, y_in, y_out.to(dev))
    else:
        m = Transformer(64, 160, T, w2i, i2w, teacher_forcing_prob=0.2, config=cfg)
        x, xl, y_in, y_out = syn.synthetic_unimodal_batch(3, 64, 160, T, V, w2i["<sos>"], w2i["<eos>"], seed=10 + rank)
        batch = (x.to(dev), xl, y_in, y_out.to(dev))
    m.flatten_parameters(device=dev)
    m.train()
    seed_dropout(55, rank)
    opt = m.configure_optimizers()
    red = m.attach_reducer()
    assert red.world == world and red.use_stream
    start = m._flat.master.clone()
    losses, touched = [], []
    for i in range(6):
        opt.zero_grad()
        loss = m.training_step(batch, i)
        loss.backward()
        red.finish()
        opt.step(grad_scale=red.grad_scale)
        losses.append(float(loss))
        touched.append(getattr(m, "_touched", None))
    torch.cuda.synchronize()
    f = m._flat
    ok_finite = all(l == l and abs(l) < 1e4 for l in losses)
    moved = not torch.equal(start, f.master)
    # identical replicas: max |difference| of master / moments across the ranks
    diffs = []
    for t in (f.master, f.exp_avg, f.exp_avg_sq):
        lo, hi = t.clone(), t.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        diffs.append(float((hi - lo).abs().max()))
    q.put((rank, ok_finite, moved, diffs, dict(opt.steps), touched))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["unimodal", "multimodal"])
def test_two_ranks_sharing_the_gpu_stay_identical(kind):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kind, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=300) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, ok_finite, moved, diffs, steps, touched in got:
        assert ok_finite and moved, (rank, ok_finite, moved)
        assert diffs == [0.0, 0.0, 0.0], (rank, diffs)
    assert got[0][4] == got[1][4] and got[0][5] == got[1][5]          # same per-sub-module step counts, same modality decisions
    if kind == "multimodal":
        assert len(set(got[0][5])) >= 2, got[0][5]                    # at least one dropped-modality step and one mixed step
