import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omr_a2s_multimodal_transformer_amd import kernels as K
dev = "cuda:0"
M, N, Kd = 131072, 3072, 256
dy = torch.randn(M, N, device=dev, dtype=torch.bfloat16) * 0.1
x = torch.randn(M, Kd, device=dev, dtype=torch.bfloat16)
dw = torch.zeros(N, Kd, device=dev)
db = torch.zeros(N, device=dev)
def run(): K.linear_wgrad_grouped([(dy, x, dw, db, None)])
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10 * 1e3
print(f"K|V dW {N}x{Kd} over {M} rows: {t:.1f} us ({2.0*M*N*Kd/t/1e6:.0f} TF/s)")
# the decoder-layer set of one flush point: 6 layers x (in_proj 768, out 256, q 256, out 256, lin1 1024x256, lin2 256x1024), rows 16384
R = 16384
probs = []
for l in range(6):
    for n_out, n_in in ((768, 256), (256, 256), (256, 256), (256, 256), (1024, 256), (256, 1024)):
        probs.append((torch.randn(R, n_out, device=dev, dtype=torch.bfloat16), torch.randn(R, n_in, device=dev, dtype=torch.bfloat16), torch.zeros(n_out, n_in, device=dev), torch.zeros(n_out, device=dev), None))
def run2(): K.linear_wgrad_grouped(probs[:24]); K.linear_wgrad_grouped(probs[24:])
for _ in range(3): run2()
torch.cuda.synchronize()
e0.record()
for _ in range(10): run2()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10 * 1e3
fl = sum(2.0 * R * p[0].shape[1] * p[1].shape[1] for p in probs)
print(f"36 decoder-layer dW problems: {t:.1f} us ({fl/t/1e6:.0f} TF/s)")
