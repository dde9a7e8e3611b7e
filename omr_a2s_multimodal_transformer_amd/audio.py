"""GPU audio front end: the reference's log-STFT spectrogram
(src/data/preprocessing.py:17-30) for signals already at 22 050 Hz.

The windowed DFT is ONE fp32 MFMA GEMM: the zero-padded signal is viewed as overlapping frames (row stride = hop, no
framing copy) and multiplied with the Hann-weighted [cos | -sin] basis of the 195 kept bins; a small kernel pair then does
magnitude, amplitude_to_db(ref=max, top_db=80) and the reference's /80 + 1.  librosa is not installed here, so parity is
against oracle/ref_cpu.log_stft (librosa's documented algorithm) -- "parity unpinned" with respect to librosa itself; the
resampling step is not covered.
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
