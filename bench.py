#!/usr/bin/env python3
"""Headline benchmark: training samples/s of the encoder->decoder step (forward + backward + Adam) on synthetic
score-image batches, BASELINE.json config C2: 6-layer d_model=256 image encoder + **bekern decoder, bs=32 per GPU,
bf16 compute / fp32 master, 256x2048 images, T=512, V=6997, dropout ON.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

Rank 0 prints ONE JSON line with the driver's contract plus `roofline` (dominant kernel, live HIP-event timing
against the MI355X peak) and `cpu_baseline` (the CPU oracle timed on this node's host cores on a bounded sample).
"""
import argparse
import json
import os
import random
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
MFMA_BF16_PEAK_TF = 2500.0   # dense bf16 MFMA


def fwd_flops_per_sample(H, W, T, S, d, L, V):
    """encoder 122908*H*W + decoder L*(16Td^2 + 4Sd^2 + 4T^2d + 4TSd) + head 2TdV."""
    return 122908.0 * H * W + L * (16.0 * T * d * d + 4.0 * S * d * d + 4.0 * T * T * d + 4.0 * T * S * d) + 2.0 * T * d * V


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (weak scaling)")
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--seq", type=int, default=512)
    ap.add_argument("--layers", type=int, default=6)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    import torch.distributed as dist
    from omr_a2s_multimodal_transformer_amd import synthetic as syn
    from omr_a2s_multimodal_transformer_amd.config import ModelConfig
    from omr_a2s_multimodal_transformer_amd.model import Transformer
    from omr_a2s_multimodal_transformer_amd.runtime import seed_dropout

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    V, H, W, T, B = syn.GRANDSTAFF_VOCAB, args.height, args.width, args.seq, args.batch
    cfg = ModelConfig(d_model=256, nhead=4, ff_dim=256, num_layers=args.layers, compute_dtype=args.dtype)
    # len(w2i) is the vocabulary size in the reference (model.py:101-103); ids as in grandstaff/vocabs/ar_w2i_kern.json
    w2i = {("<PAD>" if i == 0 else "<eos>" if i == syn.GRANDSTAFF_EOS else "<sos>" if i == syn.GRANDSTAFF_SOS else f"t{i}"): i for i in range(V)}
    i2w = {v: k for k, v in w2i.items()}

    torch.manual_seed(0)       # identical random-init weights on every rank (torch default init distributions)
    random.seed(1234)          # Python RNG drives dropout placement / teacher-forcing noise: same stream on all ranks
    model = Transformer(H, W, T, w2i, i2w, attn_window=-1, teacher_forcing_prob=0.2, config=cfg)
    model.flatten_parameters(device=dev)
    model.train()
    seed_dropout(1234, rank)
    reducer = model.attach_reducer() if world > 1 else None
    opt = model.configure_optimizers()

    x, xl, y_in, y_out = syn.synthetic_unimodal_batch(B, H, W, T, V, syn.GRANDSTAFF_SOS, syn.GRANDSTAFF_EOS, seed=1234 + rank)
    x, xl, y_out = x.to(dev), xl.to(dev), y_out.to(dev)
    y_in = y_in.pin_memory()    # token noise is host logic in the reference (model.py:152-160); stays on the host
    batch = (x, xl, y_in, y_out)

    def step(i):
        opt.zero_grad()
        loss = model.training_step(batch, i)
        loss.backward()
        if reducer is not None:
            reducer.finish()
            opt.step(grad_scale=reducer.grad_scale)
        else:
            opt.step()
        return loss

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    final_loss = float(loss.detach())

    S = ((H + 15) // 16) * ((W + 7) // 8)
    step_flops = 3.0 * fwd_flops_per_sample(H, W, T, S, 256, args.layers, V)
    samples_per_s = B * world * args.steps / elapsed

    out = {
        "metric": "training samples/sec (score+audio pairs)", "value": round(samples_per_s, 3), "unit": "samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000.0 * elapsed / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"C2: image encoder (CNN) + {args.layers}-layer d_model=256 kern decoder, fwd+bwd+Adam, dropout on",
                   "per_gpu_batch": B, "global_batch": B * world, "image": f"{H}x{W}x1", "seq_len": T, "memory_tokens": S, "vocab": V,
                   "parallelism": f"dp{world}"},
        "final_loss": round(final_loss, 4),
        "step_tflops_algorithmic": round(step_flops * samples_per_s / 1e12, 2),
        "mfma_frac_whole_step": round(step_flops * samples_per_s / world / 1e12 / MFMA_BF16_PEAK_TF, 4),
    }

    if rank == 0 and not args.no_roofline:
        out["roofline"] = roofline_dominant_kernel(B, H, W, args.dtype)
        out["decode"] = decode_rate(model, x[:1])
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(H, W, T, V, args.layers)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def decode_rate(model, x1, steps=96):
    """Greedy decode rate (bs=1, KV cache, host argmax readback per token like model.py:187) on one benchmark image.
    Random-init weights never emit <eos> reliably, so a fixed number of steps is timed."""
    from omr_a2s_multimodal_transformer_amd import kernels as K
    model.eval()
    with torch.no_grad():
        mem = model.encode(x1)
        st = model.decoder.init_decode(mem)
        tok = torch.full((1, 1), model.w2i["<sos>"], dtype=torch.int64, device=mem.device)
        for i in range(steps + 8):
            if i == 8:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            idx, _ = K.argmax(model.decoder.decode_step(tok, st).contiguous())
            tok = idx.view(1, 1)
            _ = int(idx.item())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        # batched greedy: the same step over 32 same-sized images, one host readback per 8 steps
        Bd = 32
        memb = mem.expand(Bd, -1, -1).contiguous()
        stb = model.decoder.init_decode(memb)
        tokb = torch.full((Bd, 1), model.w2i["<sos>"], dtype=torch.int64, device=mem.device)
        for i in range(steps + 8):
            if i == 8:
                torch.cuda.synchronize()
                t1 = time.perf_counter()
            idx, _ = K.argmax(model.decoder.decode_step(tokb, stb).contiguous())
            tokb = idx.view(Bd, 1)
            if i % 8 == 7:
                _ = idx.cpu()
        torch.cuda.synchronize()
        dtb = time.perf_counter() - t1
    model.train()
    return {"tokens_per_s": round(steps / dt, 1), "steps": steps, "memory_tokens": int(mem.shape[1]), "kv_cache": True,
            "batched_tokens_per_s": round(Bd * steps / dtb, 1), "batch": Bd}


def roofline_dominant_kernel(B, H, W, dtype):
    """Dominant kernel of the step (profiles/): conv3x3_mfma on conv_blocks.1.conv2 (32->32 channels at full
    resolution) -- the largest single contraction of the encoder.  Timed live with HIP events on the launch stream.
    Algorithmic bytes per launch (SURVEY.md section 8d accounting: read the input once, write the output once, weights
    negligible) = B*H*W*(Cin + Cout)*sizeof; algorithmic flops = 2*9*Cin*Cout*B*H*W."""
    from omr_a2s_multimodal_transformer_amd import kernels as K
    dt = torch.bfloat16 if dtype == "bf16" else torch.float32
    cin = cout = 32
    x = torch.rand((B, H, W, cin), device="cuda").to(dt)
    w = (torch.rand((cout, 3, 3, cin), device="cuda") - 0.5).to(dt)
    bias = torch.zeros(cout, device="cuda")
    for _ in range(2):
        K.conv3x3(x, w, bias, relu=True)
    torch.cuda.synchronize()
    n = 5
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        K.conv3x3(x, w, bias, relu=True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    nbytes = float(B) * H * W * (cin + cout) * x.element_size()
    flops = 2.0 * 9 * cin * cout * B * H * W
    gbs = nbytes / (ms * 1e-3) / 1e9
    traffic = None   # HBM bytes per launch from the committed rocprofv3 PMC passes of this same launch (profiles/)
    try:
        with open(os.path.join(ROOT, "profiles", "r01_dominant_kernel_pmc.json")) as f:
            if (B, H, W, dtype) == (32, 256, 2048, "bf16"):
                traffic = json.load(f)["hbm_bytes_per_launch"]
    except Exception:
        pass
    return {"kernel": "conv3x3_mfma_kernel (conv_blocks.1.conv2: 32->32 ch @ full resolution)", "bound": "hbm",
            "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": traffic,
            "avg_launch_ms": round(ms, 4), "algorithmic_bytes_per_launch": nbytes,
            "mfma_tflops": round(flops / (ms * 1e-3) / 1e12, 1), "mfma_frac": round(flops / (ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TF, 4)}


def usable_cores():
    """Host cores this process may really use: scheduler affinity capped by the cgroup CPU quota (the GPU box shows
    256 logical CPUs but grants 16: running torch with 256 threads there is a 20x slowdown, not a baseline)."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(H, W, T, V, layers):
    """The CPU oracle (oracle/ref_cpu.py, fp32, plain torch ops = the reference's arithmetic) timed on this node's host
    cores: forward + backward + Adam on a bounded sample (B=2 at the benchmark shapes; one timed step after a small
    thread-pool warm-up)."""
    from oracle import ref_cpu as R
    from omr_a2s_multimodal_transformer_amd import synthetic as syn
    cores = usable_cores()
    torch.set_num_threads(cores)
    Bc = 2
    sd = syn.seeded_state_dict(syn.transformer_shapes(V, 256, 256, layers), 0, mode="torch_default")
    for v in sd.values():
        v.requires_grad_(True)
    ps = list(sd.values())
    m = [torch.zeros_like(p) for p in ps]
    v2 = [torch.zeros_like(p) for p in ps]
    cfg = R.OracleCfg(num_layers=layers)

    def one(step, x, xl, y_in, y_out, h, w):
        for p in ps:
            p.grad = None
        loss = R.ce_loss(R.transformer_forward(sd, x, xl, y_in, cfg, h, w), y_out)
        loss.backward()
        with torch.no_grad():
            R.adam_step(ps, [p.grad for p in ps], m, v2, step)

    print(f"[bench] cpu_baseline: warm-up on {cores} threads", file=sys.stderr, flush=True)
    one(1, *syn.synthetic_unimodal_batch(1, 32, 128, 16, V, syn.GRANDSTAFF_SOS, syn.GRANDSTAFF_EOS, seed=2), 32, 128)
    batch = syn.synthetic_unimodal_batch(Bc, H, W, T, V, syn.GRANDSTAFF_SOS, syn.GRANDSTAFF_EOS, seed=1)
    print("[bench] cpu_baseline: timed step", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    one(2, *batch, H, W)
    dt = time.perf_counter() - t0
    print(f"[bench] cpu_baseline: {dt:.1f} s", file=sys.stderr, flush=True)
    return {"value": round(Bc / dt, 4), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"1 timed step of B={Bc} at the benchmark shapes ({H}x{W}, T={T}, L={layers}, V={V}), fp32, dropout off"}


if __name__ == "__main__":
    main()
