"""What this box's HBM delivers to a plain streaming kernel: device-to-device copy and read-modify-write of a tensor
the size of the dominant conv's activations (B=32, 256x256, 32 channels, bf16 = 134 MB), so roofline.frac of the
HBM-bound kernels can be read against the practical ceiling as well as the 8 TB/s data-sheet peak."""
import torch

def t(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

for mb in (134, 1074):
    n = mb * 1000 * 1000 // 2
    x = torch.randn(n, device="cuda", dtype=torch.bfloat16)
    y = torch.empty_like(x)
    s = t(lambda: y.copy_(x))
    print(f"copy            {mb:5d} MB in + {mb:5d} MB out: {s * 1e6:8.1f} us  {2 * n * 2 / s / 1e12:5.2f} TB/s")
    s = t(lambda: x.mul_(1.0001))
    print(f"scale in place  {mb:5d} MB in + {mb:5d} MB out: {s * 1e6:8.1f} us  {2 * n * 2 / s / 1e12:5.2f} TB/s")
    s = t(lambda: x.sum())
    print(f"read only (sum) {mb:5d} MB in               : {s * 1e6:8.1f} us  {n * 2 / s / 1e12:5.2f} TB/s")
    del x, y
