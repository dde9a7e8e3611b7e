"""bs = 1 KV-cached greedy decode rate on one benchmark image (what bench.py reports under "decode"), three repeats."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from omr_a2s_multimodal_transformer_amd import synthetic as syn
from omr_a2s_multimodal_transformer_amd.config import ModelConfig
from omr_a2s_multimodal_transformer_amd.model import Transformer
V = syn.GRANDSTAFF_VOCAB
w2i, i2w = syn.make_vocab(V)
m = Transformer(256, 2048, 512, w2i, i2w, config=ModelConfig(num_layers=6, compute_dtype="bf16"))
m.flatten_parameters(); m.eval()
x = torch.rand(1, 1, 256, 2048, device="cuda")
for _ in range(3):
    print(bench.decode_rate(m, x), flush=True)
