"""Greedy-decode rate of the native executor at the benchmark's shapes (C2: S = 4096 memory tokens, L = 6, bf16):
python tools/decode_bench.py [B] [chunk] [tokens]   (profile with rocprofv3 --kernel-trace --stats -- python3 tools/decode_bench.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omr_a2s_multimodal_transformer_amd import synthetic as syn
from omr_a2s_multimodal_transformer_amd.config import ModelConfig
from omr_a2s_multimodal_transformer_amd.model import Transformer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 16
tokens = int(sys.argv[3]) if len(sys.argv) > 3 else 256
V = syn.GRANDSTAFF_VOCAB
w2i, i2w = syn.make_vocab(V)
torch.manual_seed(0)
m = Transformer(256, 2048, 512, w2i, i2w, config=ModelConfig(num_layers=6, compute_dtype="bf16")).eval()
m.flatten_parameters()
with torch.no_grad():
    mem = m.encode(torch.rand(1, 1, 256, 2048).cuda()).expand(B, -1, -1).contiguous()
    st = m.decoder.init_decode(mem)
    tok = torch.full((B, 1), w2i["<sos>"], dtype=torch.int64, device="cuda")
    toks, _ = m.decoder.decode_tokens(tok, st, chunk)
    tok = toks[-1].view(B, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(tokens // chunk):
        toks, _ = m.decoder.decode_tokens(tok, st, chunk)
        _ = toks.cpu()
        tok = toks[-1].view(B, 1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f"B={B} chunk={chunk}: {B * (tokens // chunk * chunk) / dt:.1f} tokens/s, {1e3 * dt / (tokens // chunk * chunk):.3f} ms per position")
