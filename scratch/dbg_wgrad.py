import os, sys, torch
sys.path.insert(0, "/root/repo")
from omr_a2s_multimodal_transformer_amd import kernels as K
from tools.gemm_shapes import timeit
dev = torch.device("cuda:0")
for (ci, co) in ((16, 16), (32, 32), (64, 64)):
    x = torch.randn(32, 256 if ci < 64 else 128, 2048 if ci < 64 else 1024, ci, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(32, 256 if ci < 64 else 128, 2048 if ci < 64 else 1024, co, device=dev, dtype=torch.bfloat16)
    dw = torch.zeros(co, 3, 3, ci, device=dev); db = torch.zeros(co, device=dev)
    for dbg in (0, 2, 4, 6, 32, 38, 46, 62):
        os.environ["OMR_WGRAD_DBG"] = str(dbg)
        t = timeit(lambda: K.conv3x3_wgrad(x, dy, dw, stride=(1, 1), db=db), 10)
        print(ci, co, "dbg", dbg, f"{t:8.1f} us", flush=True)
