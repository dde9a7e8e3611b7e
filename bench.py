#!/usr/bin/env python3
"""Headline benchmark: training samples/s of the encoder->decoder step (forward + backward + Adam) on synthetic batches.

    python bench.py --gpus N --steps K --warmup W [--config c2]     (N>1: launched by torch.distributed.run, one rank per GPU)

Default workload = BASELINE.json configs[1] (C2): 6-layer d_model=256 image encoder + kern decoder, bs=32 per GPU, bf16 compute /
fp32 master, 256x2048 images, T=512, V=6997, dropout ON, teacher-forcing noise ON.  Other BASELINE configs (one JSON line each
is kept under profiles/): c1 tiny (d=128, L=2, 128x1024, T=128, fp32), c3 audio-only 195x512 log-STFT, c3mel audio-only
80x1024 mel, c4 multimodal (image 256x2048 + audio 195x512, `concat` mixer, per-GPU batch 8 = global 64 on 8 GPUs).

Rank 0 prints ONE JSON line with the driver's contract plus `roofline` (dominant kernel, live HIP-event timing against the
MI355X peak) and `cpu_baseline` (the CPU oracle timed on this node's host cores on a bounded sample).
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

# name: (modalities, per-GPU batch, image HxW, audio HxW, T, layers, d_model, dtype)
CONFIGS = {
    "c1": dict(kind="image", batch=16, img=(128, 1024), aud=None, seq=128, layers=2, d=128, dtype="fp32",
               what="C1: tiny 2-layer d_model=128 image-only OMR transformer, 16 samples"),
    "c2": dict(kind="image", batch=32, img=(256, 2048), aud=None, seq=512, layers=6, d=256, dtype="bf16",
               what="C2: image encoder (CNN) + 6-layer d_model=256 kern decoder"),
    "c3": dict(kind="audio", batch=32, img=None, aud=(195, 512), seq=512, layers=6, d=256, dtype="bf16",
               what="C3: audio-only branch, 195-bin log-STFT spectrogram encoder + 6-layer decoder"),
    "c3mel": dict(kind="audio", batch=32, img=None, aud=(80, 1024), seq=512, layers=6, d=256, dtype="bf16",
                  what="C3 (mel variant): audio-only branch, T x 80 mel-spectrogram encoder + 6-layer decoder"),
    "c4": dict(kind="multimodal", batch=8, img=(256, 2048), aud=(195, 512), seq=512, layers=6, d=256, dtype="bf16",
               what="C4: full multimodal (image + audio dual encoder, concat mixer, shared cross-attn decoder), global batch 64 on 8 GPUs"),
}


def tokens_of(hw):
    return ((hw[0] + 15) // 16) * ((hw[1] + 7) // 8)


def fwd_flops_per_sample(c, V):
    """encoder 122908*H*W per encoder + decoder L*(16Td^2 + 4Sd^2 + 4T^2d + 4TSd) + head 2TdV."""
    enc = sum(122908.0 * hw[0] * hw[1] for hw in (c["img"], c["aud"]) if hw)
    S = sum(tokens_of(hw) for hw in (c["img"], c["aud"]) if hw)
    T, d, L = c["seq"], c["d"], c["layers"]
    return enc + L * (16.0 * T * d * d + 4.0 * S * d * d + 4.0 * T * T * d + 4.0 * T * S * d) + 2.0 * T * d * V, S


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch override (weak scaling)")
    ap.add_argument("--layers", type=int, default=0)
    ap.add_argument("--dtype", default="", choices=["", "bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-side-stream", action="store_true",
                    help="profiling aid: weight gradients on the main stream (kernel durations in a trace are then not stretched by overlap)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="rehearse the multi-rank launch on the CPU: gloo group, no GPU work, prints the JSON contract with value 0")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))       # BEFORE anything touches the GPU: the ranks are child processes, never an exec
    if args.dry_launch:
        return dry_launch(args)
    c = dict(CONFIGS[args.config])
    if args.batch:
        c["batch"] = args.batch
    if args.layers:
        c["layers"] = args.layers
    if args.dtype:
        c["dtype"] = args.dtype

    import torch.distributed as dist
    from omr_a2s_multimodal_transformer_amd import synthetic as syn
    from omr_a2s_multimodal_transformer_amd.config import ModelConfig
    from omr_a2s_multimodal_transformer_amd.model import MultimodalTransformer, Transformer
    from omr_a2s_multimodal_transformer_amd.runtime import seed_dropout

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                 f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    V, T, B = syn.GRANDSTAFF_VOCAB, c["seq"], c["batch"]
    cfg = ModelConfig(d_model=c["d"], nhead=4, ff_dim=c["d"], num_layers=c["layers"], compute_dtype=c["dtype"])
    # len(w2i) is the vocabulary size in the reference (model.py:101-103); ids as in grandstaff/vocabs/ar_w2i_kern.json
    w2i = {("<PAD>" if i == 0 else "<eos>" if i == syn.GRANDSTAFF_EOS else "<sos>" if i == syn.GRANDSTAFF_SOS else f"t{i}"): i for i in range(V)}
    i2w = {v: k for k, v in w2i.items()}

    torch.manual_seed(0)       # identical random-init weights on every rank (torch default init distributions)
    random.seed(1234)          # Python RNG drives dropout placement / modality drop / teacher-forcing noise: same stream on all ranks
    sos, eos = syn.GRANDSTAFF_SOS, syn.GRANDSTAFF_EOS
    if c["kind"] == "multimodal":
        model = MultimodalTransformer(c["img"][0], c["img"][1], c["aud"][0], c["aud"][1], T, w2i, i2w, mixer_type="concat", attn_window=-1,
                                      teacher_forcing_prob=0.2, teacher_forcing_modality_prob=0.2, config=cfg)      # train.py:76-95
        xi, xli, y_in, y_out = syn.synthetic_unimodal_batch(B, c["img"][0], c["img"][1], T, V, sos, eos, seed=1234 + rank)
        xa, xla, _, _ = syn.synthetic_unimodal_batch(B, c["aud"][0], c["aud"][1], T, V, sos, eos, seed=4321 + rank, pad_value=0.0)
        batch = (xi.to(dev), xli.to(dev), xa.to(dev), xla.to(dev), y_in.pin_memory(), y_out.to(dev))
        x1 = None
    else:
        hw = c["img"] if c["kind"] == "image" else c["aud"]
        model = Transformer(hw[0], hw[1], T, w2i, i2w, attn_window=-1, teacher_forcing_prob=0.2, config=cfg)         # train.py:97-105
        x, xl, y_in, y_out = syn.synthetic_unimodal_batch(B, hw[0], hw[1], T, V, sos, eos, seed=1234 + rank,
                                                          pad_value=1.0 if c["kind"] == "image" else 0.0)
        # token noise is host logic in the reference (model.py:152-160): y_in stays on the host, pinned
        batch = (x.to(dev), xl.to(dev), y_in.pin_memory(), y_out.to(dev))
        x1 = batch[0][:1]
    model.flatten_parameters(device=dev)
    model.train()
    if args.no_side_stream:
        from omr_a2s_multimodal_transformer_amd.runtime import WgradStream
        WgradStream.enabled = False
    seed_dropout(1234, rank)
    reducer = model.attach_reducer() if world > 1 else None
    opt = model.configure_optimizers()

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

    fwd_flops, S = fwd_flops_per_sample(c, V)
    step_flops = 3.0 * fwd_flops
    samples_per_s = B * world * args.steps / elapsed
    peak_tf = MFMA_BF16_PEAK_TF if c["dtype"] == "bf16" else 157.3

    out = {
        "metric": "training samples/sec", "value": round(samples_per_s, 3), "unit": "samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000.0 * elapsed / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": c["dtype"], "data": "synthetic",
        "config": {"workload": c["what"] + ", fwd+bwd+Adam, dropout on, teacher-forcing noise on; one sample = " +
                   {"image": "one score image", "audio": "one spectrogram", "multimodal": "one score image + one spectrogram"}[c["kind"]] +
                   " + its token sequence", "name": args.config, "per_gpu_batch": B, "global_batch": B * world,
                   "image": None if not c["img"] else f"{c['img'][0]}x{c['img'][1]}x1", "audio": None if not c["aud"] else f"{c['aud'][0]}x{c['aud'][1]}x1",
                   "seq_len": T, "memory_tokens": S, "layers": c["layers"], "d_model": c["d"], "vocab": V, "parallelism": f"dp{world}"},
        "final_loss": round(final_loss, 4),
        "step_tflops_algorithmic": round(step_flops * samples_per_s / 1e12, 2),
        "mfma_frac_whole_step": round(step_flops * samples_per_s / world / 1e12 / peak_tf, 4),
    }

    if rank == 0 and not args.no_roofline and args.config == "c2":
        out["roofline"] = roofline_dominant_kernel(B, c["img"][0], c["img"][1], c["dtype"])
        out["roofline"]["others"] = roofline_others(B, c["img"][0], c["img"][1], T, c["d"], 4)
    if rank == 0 and not args.no_roofline and x1 is not None:
        out["decode"] = decode_rate(model, x1)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(c, V)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) through torch.distributed.run as a CHILD
    process with the same arguments, relay its output (rank 0 prints the JSON line) and return its exit code.  Lightning
    starts the reference's ranks the same way (train.py:140-154: Trainer(devices=..., strategy=ddp) spawns them)."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores() // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def dry_launch(args):
    """CPU rehearsal of the launch contract (tests/test_bench_launch_cpu.py): the ranks form a gloo group, agree on the world
    size, run the barrier + MAX-over-ranks timing protocol around an empty region and rank 0 prints the JSON line."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        dist.barrier()
    t0 = time.perf_counter()
    elapsed = time.perf_counter() - t0
    seen = torch.ones(1, dtype=torch.int64)
    if world > 1:
        dist.barrier()
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(seen)
        elapsed = float(t.item())
    if rank == 0:
        print(json.dumps({"metric": "training samples/sec", "value": 0.0, "unit": "samples/s", "n_gpus": world, "ranks_seen": int(seen.item()),
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": 0.0, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "none", "data": "none (dry launch)", "config": {"workload": "dry launch", "parallelism": f"dp{world}"}}),
              flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


def decode_rate(model, x1, steps=96):
    """Greedy decode rate (bs=1, KV cache, host argmax readback per token like model.py:187) on one benchmark input.
    Random-init weights never emit <eos> reliably, so a fixed number of steps is timed."""
    model.eval()
    chunk = 16                                   # tokens per host call / readback (Transformer._greedy's default)
    steps = max(chunk, min(steps, model.max_seq_len - chunk) // chunk * chunk)

    def run(mem, B):
        st = model.decoder.init_decode(mem)
        tok = torch.full((B, 1), model.w2i["<sos>"], dtype=torch.int64, device=mem.device)
        toks, _ = model.decoder.decode_tokens(tok, st, chunk)          # warm-up chunk
        tok = toks[-1].view(B, 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps // chunk):
            toks, _ = model.decoder.decode_tokens(tok, st, chunk)
            _ = toks.cpu()                                              # the host needs the tokens to look for <eos>
            tok = toks[-1].view(B, 1)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    with torch.no_grad():
        mem = model.encode(x1)
        dt = run(mem, 1)
        Bd = 32                                   # batched greedy: 32 same-sized inputs in lock-step
        memb = mem.expand(Bd, -1, -1).contiguous()
        dtb = run(memb, Bd)
        # BASELINE configs[4] "fp8 MFMA weights" (an extension): the same decode with every matrix of a position as e4m3 rows
        # (dequantised on load by the row kernel; half the weight bytes per position)
        model.decoder.fp8_weights = True
        dt8, dtb8 = run(mem, 1), run(memb, Bd)
        model.decoder.fp8_weights = False
    model.train()
    return {"tokens_per_s": round(steps / dt, 1), "steps": steps, "memory_tokens": int(mem.shape[1]), "kv_cache": True, "tokens_per_host_call": chunk,
            "batched_tokens_per_s": round(Bd * steps / dtb, 1), "batch": Bd,
            "fp8_weights_tokens_per_s": round(steps / dt8, 1), "fp8_weights_batched_tokens_per_s": round(Bd * steps / dtb8, 1)}


def roofline_dominant_kernel(B, H, W, dtype):
    """Dominant kernel of the step (profiles/): conv3x3_mfma on conv_blocks.1.conv2 (32->32 channels at full resolution) --
    the largest single contraction of the encoder -- IN THE INSTANTIATION THE TRAINING STEP RUNS: bias + ReLU + fused
    InstanceNorm statistics of the output (template EPI = 1; with the block's MixDropout on this conv, one step in three, the
    EPI = 3 variant, timed as well; the bias/ReLU-only EPI = 0 variant serves the masked data gradient).  Timed live with HIP
    events on the launch stream.  Algorithmic bytes per launch (SURVEY.md section
    8d accounting: read the input once, write the output once, weights and statistics negligible) = B*H*W*(Cin + Cout)*sizeof;
    algorithmic flops = 2*9*Cin*Cout*B*H*W."""
    from omr_a2s_multimodal_transformer_amd import kernels as K
    dt = torch.bfloat16 if dtype == "bf16" else torch.float32
    cin = cout = 32
    x = torch.rand((B, H, W, cin), device="cuda").to(dt)
    w = (torch.rand((cout, 3, 3, cin), device="cuda") - 0.5).to(dt)
    bias = torch.zeros(cout, device="cuda")
    ws, slots = K.conv_stat_ws(B, H, W, cout, x.device)

    def timed(drop):
        def launch():
            return K.conv3x3(x, w, bias, relu=True, drop=drop, stat_mode=1, stat_ws=ws, stat_slots=slots)
        for _ in range(3):
            launch()
        torch.cuda.synchronize()
        n = 16
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            launch()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    ms = timed(None)                          # what the step launches in 2 steps out of 3 (the block's MixDropout sits on another conv)
    ms_drop = timed((0.5, 1234, False))       # ... and with the elementwise MixDropout fused in (1 step out of 6; channel mode is cheaper)
    nbytes = float(B) * H * W * (cin + cout) * x.element_size()
    flops = 2.0 * 9 * cin * cout * B * H * W
    gbs = nbytes / (ms * 1e-3) / 1e9
    traffic = None   # HBM bytes per launch from the committed rocprofv3 PMC passes of this same launch (profiles/)
    try:
        with open(os.path.join(ROOT, "profiles", "r03_dominant_kernel_pmc.json")) as f:
            if (B, H, W, dtype) == (32, 256, 2048, "bf16"):
                traffic = json.load(f)["hbm_bytes_per_launch"]
    except Exception:
        pass
    return {"kernel": "conv3x3_mfma_kernel<EPI=1> (conv_blocks.1.conv2 forward: 32->32 ch @ full resolution, bias+ReLU+InstanceNorm statistics)",
            "avg_launch_ms_with_fused_elementwise_dropout": round(ms_drop, 4),
            "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": traffic,
            "avg_launch_ms": round(ms, 4), "algorithmic_bytes_per_launch": nbytes,
            "mfma_tflops": round(flops / (ms * 1e-3) / 1e12, 1), "mfma_frac": round(flops / (ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TF, 4)}


def _timed(fn, n=12, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def roofline_others(B, H, W, T, d, nhead):
    """The other large kernels of a C2 step by time (profiles/r03_c2_families_*.txt), each against the roofline that bounds it,
    timed live at the step's own shapes (HIP events on the launch stream; the library launches on torch's current stream):
    cross-attention forward / backward (one decoder layer's: B x 4 heads x 512 queries x 4096 memory keys, dropout + key bias as
    in training; MFMA-eligible work 4*T*S*d*B flops forward, 2.5x that backward), InstanceNorm backward apply on the largest map
    (32 channels at full resolution: 2 reads + 1 write; the step itself no longer runs it on that map), the one-pass backward of the
    32-channel full-resolution conv that replaced it (3 reads + 1 write), the 64-channel conv weight gradient (conv_blocks.2.conv2:
    18*64*64 flops per pixel; also 2 tensor reads)."""
    from omr_a2s_multimodal_transformer_amd import kernels as K
    dev, bf = torch.device("cuda"), torch.bfloat16
    S = tokens_of((H, W))
    res = []
    q = torch.randn(B, T, d, device=dev, dtype=bf)
    kv = torch.randn(B, S, 2 * d, device=dev, dtype=bf)
    k, v = kv[..., :d], kv[..., d:]
    kb = torch.zeros(B, S, device=dev)
    words = K.attn_dropout_words(B, nhead, T, S, 0.1, 7, dev)
    kw = dict(key_bias=kb, dropout_p=0.1, seed=7, drop_words=words)
    o, lse = K.attn_fwd(q, k, v, nhead, **kw)
    do, dq, dkv = torch.randn_like(o), torch.empty_like(q), torch.empty_like(kv)
    fl = 4.0 * B * T * S * d
    ms = _timed(lambda: K.attn_fwd(q, k, v, nhead, **kw))
    res.append({"kernel": "attn_fwd_kernel (cross-attention of one decoder layer, dropout 0.1 + key bias)", "bound": "mfma", "avg_launch_ms": round(ms, 4),
                "algorithmic_flops": fl, "achieved": round(fl / ms / 1e9, 1), "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s", "frac": round(fl / ms / 1e9 / MFMA_BF16_PEAK_TF, 4)})
    ms = _timed(lambda: K.attn_bwd(q, k, v, o, do, lse, dq, dkv[..., :d], dkv[..., d:], nhead, **kw))
    res.append({"kernel": "attn_bwd_dq (forms delta) + attn_bwd_dkv (the same layer's backward)", "bound": "mfma", "avg_launch_ms": round(ms, 4),
                "algorithmic_flops": 2.5 * fl, "achieved": round(2.5 * fl / ms / 1e9, 1), "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                "frac": round(2.5 * fl / ms / 1e9 / MFMA_BF16_PEAK_TF, 4)})
    del q, kv, o, do, dq, dkv, words
    C = 32
    x = torch.rand((B, H, W, C), device=dev).to(bf)
    g = torch.rand((B, H, W, C), device=dev).to(bf)
    mean, rstd = K.instnorm_stats(x)
    ws, slots = K.conv_stat_ws(B, H, W, C, dev)
    ws.zero_()
    ms = _timed(lambda: K.instnorm_bwd_apply(g, x, mean, rstd, ws, slots, relu_mask=True, relu_scale=1.0))
    nb = 3.0 * B * H * W * C * 2
    res.append({"kernel": "instnorm_bwd_apply_kernel (conv_blocks.1: 32 channels at full resolution)", "bound": "hbm", "avg_launch_ms": round(ms, 4),
                "algorithmic_bytes": nb, "achieved": round(nb / ms / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(nb / ms / 1e6 / HBM_PEAK_GBS, 4)})
    # the one-pass backward of conv_blocks.1.conv2 (32 -> 32 channels at full resolution) with the InstanceNorm backward applied on
    # load: reads dL/dxhat, the norm's input and the conv's input once, writes the data gradient once (weight / bias gradients are 37 KB)
    y = torch.rand((B, H, W, C), device=dev).sub_(0.4).clamp_min_(0).to(bf)
    wf = K.conv3x3_weight_flip((torch.rand((C, 3, 3, C), device=dev) - 0.5).to(bf))
    dw = torch.zeros((C, 3, 3, C), device=dev)
    db = torch.zeros(C, device=dev)
    K.instnorm_reduce_sums(ws, slots, B, C)
    ms = _timed(lambda: K.conv3x3_bwd_fused(g, x, wf, dw, db, True, 1.0, norm=(y, mean, rstd, ws, slots, True, 1.0)))
    nb = 4.0 * B * H * W * C * 2
    res.append({"kernel": "conv_bwd_fused_kernel<32,32,apply> (conv_blocks.1.conv2: data + weight + bias gradient, InstanceNorm backward on load)",
                "bound": "hbm", "avg_launch_ms": round(ms, 4), "algorithmic_bytes": nb, "algorithmic_flops": 2 * 18.0 * C * C * B * H * W,
                "achieved": round(nb / ms / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(nb / ms / 1e6 / HBM_PEAK_GBS, 4)})
    del x, g, y
    C, h2, w2 = 64, H // 2, W // 2
    x = torch.rand((B, h2, w2, C), device=dev).to(bf)
    dy = torch.rand((B, h2, w2, C), device=dev).to(bf)
    dw = torch.zeros((C, 3, 3, C), device=dev)
    db = torch.zeros(C, device=dev)
    ms = _timed(lambda: K.conv3x3_wgrad(x, dy, dw, db=db))
    fl = 18.0 * C * C * B * h2 * w2
    res.append({"kernel": "wgrad_dma_kernel<64,64> (conv_blocks.2.conv2 weight gradient: 64 -> 64 channels at 128x1024)", "bound": "mfma", "avg_launch_ms": round(ms, 4),
                "algorithmic_flops": fl, "algorithmic_bytes": 2.0 * B * h2 * w2 * C * 2, "achieved": round(fl / ms / 1e9, 1), "peak": MFMA_BF16_PEAK_TF,
                "unit": "TFLOP/s", "frac": round(fl / ms / 1e9 / MFMA_BF16_PEAK_TF, 4)})
    return res


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


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(c, V):
    """The CPU oracle (oracle/ref_cpu.py, fp32, plain torch ops = the reference's arithmetic) timed on this node's host cores as
    SURVEY.md section 8d prescribes: forward + backward + Adam at the benchmark shapes, B = 2, dropout ON (every site draws a
    fresh Bernoulli mask with torch's CPU RNG, like the reference's nn.Dropout / nn.Dropout2d), 1 warm-up + 3 timed steps."""
    from oracle import ref_cpu as R
    from omr_a2s_multimodal_transformer_amd import synthetic as syn
    cores = usable_cores()
    torch.set_num_threads(cores)
    Bc, T, L, d = 2, c["seq"], c["layers"], c["d"]
    multimodal = c["kind"] == "multimodal"
    shapes = syn.multimodal_shapes(V, "concat", d, d, L) if multimodal else syn.transformer_shapes(V, d, d, L)
    sd = syn.seeded_state_dict(shapes, 0, mode="torch_default")
    for v in sd.values():
        v.requires_grad_(True)
    ps = list(sd.values())
    m = [torch.zeros_like(p) for p in ps]
    v2 = [torch.zeros_like(p) for p in ps]
    cfg = R.OracleCfg(d_model=d, ff_dim=d, num_layers=L)

    def bernoulli_mask(site, kind, p, shape, channel):
        if channel:
            shape = tuple(shape[:2]) + (1,) * (len(shape) - 2)
        return torch.empty(shape).bernoulli_(1.0 - p).div_(1.0 - p)

    sos, eos = syn.GRANDSTAFF_SOS, syn.GRANDSTAFF_EOS
    if multimodal:
        xi, xli, y_in, y_out = syn.synthetic_unimodal_batch(Bc, c["img"][0], c["img"][1], T, V, sos, eos, seed=1)
        xa, xla, _, _ = syn.synthetic_unimodal_batch(Bc, c["aud"][0], c["aud"][1], T, V, sos, eos, seed=2, pad_value=0.0)
        fwd = lambda plan: R.multimodal_forward(sd, xi, xli, xa, xla, y_in, cfg, "concat", c["img"], c["aud"], "both", drop=plan)
    else:
        hw = c["img"] if c["kind"] == "image" else c["aud"]
        x, xl, y_in, y_out = syn.synthetic_unimodal_batch(Bc, hw[0], hw[1], T, V, sos, eos, seed=1, pad_value=1.0 if c["kind"] == "image" else 0.0)
        fwd = lambda plan: R.transformer_forward(sd, x, xl, y_in, cfg, hw[0], hw[1], drop=plan)

    def one(step):
        for p in ps:
            p.grad = None
        loss = R.ce_loss(fwd(R.DropPlan(bernoulli_mask)), y_out)
        loss.backward()
        with torch.no_grad():
            R.adam_step(ps, [p.grad for p in ps], m, v2, step)

    random.seed(99)
    print(f"[bench] cpu_baseline: warm-up step on {cores} threads", file=sys.stderr, flush=True)
    one(1)
    times = []
    for s in range(3):
        t0 = time.perf_counter()
        one(2 + s)
        times.append(time.perf_counter() - t0)
        print(f"[bench] cpu_baseline: step {s} {times[-1]:.1f} s", file=sys.stderr, flush=True)
    dt = sum(times) / len(times)
    return {"value": round(Bc / dt, 4), "unit": "samples/s", "cores": cores, "cpu": cpu_model_name(), "kind": "port",
            "sample": f"1 warm-up + 3 timed steps (mean) of B={Bc} at the benchmark shapes, fp32, dropout on, torch threads = {cores}",
            "step_seconds": [round(t, 2) for t in times]}


if __name__ == "__main__":
    main()
