"""GPU audio front end: the reference's log-STFT spectrogram
(src/data/preprocessing.py:17-30) for signals already at 22 050 Hz.

The windowed DFT is ONE fp32 MFMA GEMM: the zero-padded signal is viewed as overlapping frames (row stride = hop, no
framing copy) and multiplied with the Hann-weighted [cos | -sin] basis of the 195 kept bins; a small kernel pair then does
magnitude, amplitude_to_db(ref=max, top_db=80) and the reference's /80 + 1.  librosa is not installed here, so parity is
against oracle/ref_cpu.log_stft (librosa's documented algorithm) -- "parity unpinned" with respect to librosa itself.

The resampling step in front of it (preprocessing.py:19) is `resample_poly`: polyphase conversion as ONE fp32 GEMM too (frames
of the signal at a stride of g*down samples times a [window, g*up] matrix of filter phases), pinned against
scipy.signal.resample_poly = librosa's res_type="polyphase"; the reference's call uses librosa's default "soxr_hq" (library
absent, not restated), so against the reference itself this step is "parity unpinned" as well.
"""
from __future__ import annotations

import math
from functools import lru_cache

import torch

from . import kernels as K
from ._lib import cur_stream, lib, ptr

N_FFT, HOP, SR, FMAX = 2048, 512, 22050, 2093.0
NUM_FREQ_BINS = sum(1 for k in range(N_FFT // 2 + 1) if k * SR / N_FFT <= FMAX)      # 195 (preprocessing.py:13)


@lru_cache(maxsize=4)
def _basis(device: str) -> torch.Tensor:
    """[2*bins, n_fft] fp32: rows 0..bins-1 = hann[n] cos(2 pi k n / N), rows bins.. = -hann[n] sin(2 pi k n / N) (float64 host math)."""
    n = torch.arange(N_FFT, dtype=torch.float64)
    win = 0.5 - 0.5 * torch.cos(2.0 * math.pi * n / N_FFT)
    k = torch.arange(NUM_FREQ_BINS, dtype=torch.float64).unsqueeze(1)
    ang = 2.0 * math.pi * k * n.unsqueeze(0) / N_FFT
    return torch.cat([torch.cos(ang) * win, -torch.sin(ang) * win], dim=0).to(torch.float32).to(device)


def log_stft(y: torch.Tensor) -> torch.Tensor:
    """y: fp32 waveform [n] on the GPU (22 050 Hz) -> normalised log-spectrogram [1, 195, 1 + n // 512] (preprocess_audio layout)."""
    assert y.is_cuda and y.dtype == torch.float32 and y.dim() == 1
    n = y.numel()
    frames = 1 + n // HOP
    padded = torch.zeros(n + N_FFT + HOP, dtype=torch.float32, device=y.device)        # centre padding (+ slack so every frame row is in bounds)
    padded[N_FFT // 2:N_FFT // 2 + n] = y
    a = torch.as_strided(padded, (frames, N_FFT), (HOP, 1))                            # overlapping frames, no copy
    spec = torch.empty((frames, 2 * NUM_FREQ_BINS), dtype=torch.float32, device=y.device)
    K.gemm(a, _basis(str(y.device)), out=spec)                                         # [frames, 2048] x [390, 2048]^T, exact fp32 MFMA chain
    out = torch.empty((NUM_FREQ_BINS, frames), dtype=torch.float32, device=y.device)
    ws = torch.empty(1, dtype=torch.int32, device=y.device)
    lib().call("omr_log_stft_post", ptr(spec), frames, NUM_FREQ_BINS, ptr(ws), ptr(out), cur_stream())
    return out.unsqueeze(0)


@lru_cache(maxsize=8)
def _resample_plan(up: int, down: int, device: str):
    """Filter phases of scipy.signal.resample_poly(window=("kaiser", 5.0)) as a GEMM operand.  Output n = q*G + p (G = g*up
    outputs per row, g the smallest count that makes the row stride g*down a multiple of 4 samples = 16 bytes) is
    sum_i' x[q*g*down + i'] h[(p + r0)*down - i'*up]: the matrix H[p][i' - lo] does not depend on q.
    -> (H [G, K] fp32 on the device, lo = first i' with a non-zero tap, g)."""
    import numpy as np
    max_rate = max(up, down)
    half_len = 10 * max_rate
    ntaps = 2 * half_len + 1
    c = 1.0 / max_rate
    m = np.arange(ntaps) - half_len
    h = c * np.sinc(c * m) * np.kaiser(ntaps, 5.0)                   # scipy.signal.firwin(ntaps, c, window=("kaiser", 5.0)) ...
    h = h / h.sum() * up                                             # ... unit DC gain, times up (resample_poly)
    n_pre_pad = down - half_len % down
    h = np.concatenate([np.zeros(n_pre_pad), h])
    r0 = (half_len + n_pre_pad) // down
    g = 1
    while (g * down) % 4:
        g += 1
    G = g * up
    lo = -((len(h) - 1 - r0 * down) // up)                           # smallest i' any phase touches: (p + r0)*down - i'*up <= len(h) - 1 at p = 0
    hi = ((G - 1 + r0) * down) // up                                 # largest: index >= 0 at p = G - 1
    lo -= lo % 4                                                     # frame rows start 16-byte aligned
    K_ = (hi - lo + 1 + 7) // 8 * 8
    H = np.zeros((G, K_))
    for p in range(G):
        for ip in range(lo, hi + 1):
            j = (p + r0) * down - ip * up
            if 0 <= j < len(h):
                H[p, ip - lo] = h[j]
    return torch.from_numpy(H.astype(np.float32)).to(device), lo, g


def resample_poly(y: torch.Tensor, orig_sr: int, target_sr: int = SR) -> torch.Tensor:
    """y: fp32 waveform [n] on the GPU at orig_sr -> [ceil(n * up / down)] at target_sr (preprocessing.py:19 with librosa's
    res_type="polyphase", i.e. scipy.signal.resample_poly; see the module docstring for what that pins)."""
    assert y.is_cuda and y.dtype == torch.float32 and y.dim() == 1
    gcd = math.gcd(int(orig_sr), int(target_sr))
    up, down = int(target_sr) // gcd, int(orig_sr) // gcd
    if up == down:
        return y.clone()
    H, lo, g = _resample_plan(up, down, str(y.device))
    G, K_ = H.shape
    n = y.numel()
    n_out = -(-n * up // down)
    rows = -(-n_out // G)
    stride = g * down
    front = -lo                                                      # zeros in front so that frame q starts at sample q*stride + lo
    padded = torch.zeros(front + (rows - 1) * stride + K_ + 8, dtype=torch.float32, device=y.device)
    padded[front:front + n] = y
    frames = torch.as_strided(padded, (rows, K_), (stride, 1))       # overlapping windows, no copy
    out = torch.empty((rows, G), dtype=torch.float32, device=y.device)
    K.gemm(frames, H, out=out)                                       # [rows, K] x [G, K]^T, exact fp32 MFMA chain
    return out.reshape(-1)[:n_out].contiguous()
