"""Host-side cost of issuing one training step (no device sync inside): if this approaches the GPU time per step the
step is launch-bound and needs hipGraph capture."""
import os, sys, time, random
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omr_a2s_multimodal_transformer_amd import synthetic as syn
from omr_a2s_multimodal_transformer_amd.config import ModelConfig
from omr_a2s_multimodal_transformer_amd.model import Transformer

V, H, W, T, B = syn.GRANDSTAFF_VOCAB, 256, 2048, 512, 32
w2i = {("<PAD>" if i == 0 else "<eos>" if i == syn.GRANDSTAFF_EOS else "<sos>" if i == syn.GRANDSTAFF_SOS else f"t{i}"): i for i in range(V)}
i2w = {v: k for k, v in w2i.items()}
m = Transformer(H, W, T, w2i, i2w, teacher_forcing_prob=0.2, config=ModelConfig(num_layers=6, compute_dtype="bf16"))
m.flatten_parameters(); m.train()
opt = m.configure_optimizers()
x, xl, y_in, y_out = syn.synthetic_unimodal_batch(B, H, W, T, V, syn.GRANDSTAFF_SOS, syn.GRANDSTAFF_EOS, seed=1)
batch = (x.cuda(), xl.cuda(), y_in.pin_memory(), y_out.cuda())
def step(i):
    opt.zero_grad(); loss = m.training_step(batch, i); loss.backward(); opt.step()
for i in range(3): step(i)
torch.cuda.synchronize()
t_tf = time.perf_counter(); m.apply_teacher_forcing(batch[2]); t_tf = time.perf_counter() - t_tf
ts = []
for i in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(i); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append((t1 - t0, t2 - t0))
print("host issue time / total per step (ms):", [(round(a * 1e3, 1), round(b * 1e3, 1)) for a, b in ts])
print("apply_teacher_forcing host loop (ms):", round(t_tf * 1e3, 2))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(3): step(i)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
