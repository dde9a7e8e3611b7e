"""The stream choreography of the data-parallel step on a real GPU, with RCCL in the loop: a one-rank "nccl" process group
(all-reduce = identity) and a reducer told it has two ranks, so every line of the world > 1 path runs -- bucket all-reduce on
the communication stream behind an event, issued from the memory boundary in the middle of backward while weight-gradient
GEMMs are still in flight on the side stream, finish(), Adam with grad_scale.  The result must equal the single-process step
with the same 1/2 gradient scale.  (The multi-rank arithmetic is covered by the gloo world-2 tests on the CPU.)"""
import random

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu

from omr_a2s_multimodal_transformer_amd import synthetic as syn  # noqa: E402
from omr_a2s_multimodal_transformer_amd.config import ModelConfig  # noqa: E402
from test_model_gpu import DEV, NO_DROP, make_transformer  # noqa: E402


@pytest.fixture()
def one_rank_group():
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29731", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        yield
    finally:
        dist.destroy_process_group()


def test_overlapped_bucket_all_reduce_equals_the_plain_step(one_rank_group):
    V = 50
    x = xl = y_in = y_out = None
    results = []
    for with_reducer in (False, True):
        m, w2i = make_transformer(V, ModelConfig(num_layers=2, **NO_DROP), 47, hw=(64, 96), max_seq=16)
        m.train()
        m.teacher_forcing_prob = 0.0
        if x is None:
            x, xl, y_in, y_out = (t.to(DEV) for t in syn.synthetic_unimodal_batch(3, 64, 96, 12, V, w2i["<sos>"], w2i["<eos>"], seed=8))
        opt = m.configure_optimizers()
        red = None
        if with_reducer:
            red = m.attach_reducer()
            assert red.world == 1
            red.world = 2                      # pretend: the collective still runs over the one real rank
        for step in range(2):
            random.seed(step)
            opt.zero_grad()
            m.compute_loss(m(x, xl, y_in), y_out).backward()
            if red is not None:
                assert red.done[1] and not red.done[0]          # decoder bucket left during backward, encoder bucket not yet
                red.finish()
                if step == 0:
                    first_grad = m._flat.grad.clone()
                opt.step(grad_scale=red.grad_scale)
            else:
                if step == 0:
                    first_grad = m._flat.grad.clone()
                opt.step(grad_scale=0.5)
        torch.cuda.synchronize()
        results.append((m._flat.master.clone(), first_grad, m._flat.grad.clone()))
    (p0, f0, g0), (p1, f1, g1) = results
    assert torch.isfinite(p1).all() and g1.abs().max() > 0
    # First step (identical parameters on both sides): every slice to fp32-atomics noise -- the forward pass and the
    # data-gradient chain are deterministic (InstanceNorm statistics are a fixed-order reduction), only the weight-gradient
    # sums use atomics.  Second step: the parameters already differ by that noise times Adam's sign-like update, and a
    # ReLU mask that flips on a 4x12 map moves whole upstream tensors by ~1/sqrt(N) (tests/test_dropout_parity_gpu.py),
    # so that comparison only guards against a lost or torn bucket (an O(1) error).
    for n, (o, c) in m._flat.offsets.items():
        r = f0[o:o + c]
        if r.abs().max() == 0:
            continue
        assert ((f1[o:o + c] - r).norm() / r.norm()).item() < 1e-5, n
        assert ((g1[o:o + c] - g0[o:o + c]).norm() / g0[o:o + c].norm()).item() < 5e-2, n
    assert ((p1 - p0).abs().max()).item() < 5e-4      # two Adam steps of lr 1e-4: parameters can differ by at most ~2 lr
