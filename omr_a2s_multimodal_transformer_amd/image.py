"""Score-image half of the input pipeline on the GPU (reference src/data/preprocessing.py:44-75):
``preprocess_image`` = PIL ``convert("L")`` -> ``resize((int(H * w / h), H))`` (Pillow's default BICUBIC, antialiased) ->
``ToTensor``, and ``pad_batch_inputs`` = right/bottom padding of a ragged list to the batch maximum.  The decoded pixels
go to the GPU as bytes (one H2D of h*w*channels uint8 instead of H*W fp32), both resampling passes and the /255 run as HIP
kernels (csrc/image.hip) and the result lands directly in its slot of the padded batch tensor, in the compute dtype.
Bit-exact with Pillow on the uint8 image, hence equal floats (tests/test_image_gpu.py, oracle.ref_cpu.pil_*).
There is no CPU path: tensors must be CUDA tensors."""
from __future__ import annotations

import ctypes
from functools import lru_cache
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from ._lib import cur_stream, dtype_code, lib, ptr

Tensor = torch.Tensor


@lru_cache(maxsize=4096)
def _tables_host(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray, int]:
    ks = lib().query("omr_resample_ksize", in_size, out_size)
    if ks <= 0:
        raise RuntimeError(f"omr_resample_ksize({in_size}, {out_size}) failed")
    bounds = np.empty((out_size, 2), dtype=np.int32)
    coefs = np.empty((out_size, ks), dtype=np.int32)
    rc = lib().query("omr_resample_coeffs", in_size, out_size, ctypes.c_void_p(bounds.ctypes.data), ctypes.c_void_p(coefs.ctypes.data))
    if rc != ks:
        raise RuntimeError(f"omr_resample_coeffs({in_size}, {out_size}) failed: {rc}")
    return bounds, coefs, ks


@lru_cache(maxsize=1024)
def _tables(in_size: int, out_size: int, device_index: int):
    bounds, coefs, ks = _tables_host(in_size, out_size)
    dev = torch.device("cuda", device_index)
    return torch.from_numpy(bounds).to(dev), torch.from_numpy(coefs).to(dev), ks


def as_hwc_uint8(raw) -> Tensor:
    """PIL image (modes L / RGB / RGBA) or uint8 array/tensor [h, w] / [h, w, c] -> CPU uint8 tensor [h, w, c]."""
    if isinstance(raw, torch.Tensor):
        t = raw
    elif isinstance(raw, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(raw))
    else:                                           # PIL.Image without importing PIL here
        mode = getattr(raw, "mode", None)
        if mode not in ("L", "RGB", "RGBA"):
            raise ValueError(f"image mode {mode!r}: the GPU front end converts L, RGB and RGBA (preprocessing.py:45)")
        t = torch.from_numpy(np.asarray(raw).copy())
    if t.dtype != torch.uint8 or t.dim() not in (2, 3):
        raise ValueError("expected uint8 pixels [h, w] or [h, w, c]")
    if t.dim() == 2:
        t = t.unsqueeze(-1)
    if t.shape[2] not in (1, 3, 4):
        raise ValueError("channels must be 1 (L), 3 (RGB) or 4 (RGBA)")
    return t.contiguous()


def resized_width(h: int, w: int, img_height: Optional[int]) -> Tuple[int, int]:
    """(height, width) after preprocess_image: width = int(img_height * w / h) (preprocessing.py:47)."""
    if img_height is None:
        return h, w
    return img_height, int(img_height * w / h)


def preprocess_image_into(pixels: Tensor, img_height: Optional[int], out: Tensor) -> Tuple[int, int]:
    """pixels: CUDA uint8 [h, w, c]; out: CUDA fp32/bf16 2-D view [>=H, >=W] (unit inner stride) whose top-left H x W
    receives the image.  Returns (H, W)."""
    if not (pixels.is_cuda and out.is_cuda):
        raise RuntimeError("omr_a2s_multimodal_transformer_amd.image runs on the GPU only (HIP kernels); there is no CPU fallback")
    h, w, c = pixels.shape
    H, W = resized_width(h, w, img_height)
    if H <= 0 or W <= 0:
        raise ValueError(f"resized image would be {H}x{W}")
    assert out.dim() == 2 and out.stride(1) == 1 and out.shape[0] >= H and out.shape[1] >= W
    dev = pixels.device
    dev_index = dev.index if dev.index is not None else torch.cuda.current_device()
    bh = ch = bv = cv = None
    ksh = ksv = 0
    if W != w:
        bh, ch, ksh = _tables(w, W, dev_index)
    if H != h:
        bv, cv, ksv = _tables(h, H, dev_index)
    if W != w or c != 1:
        tmp = torch.empty((h, W), dtype=torch.uint8, device=dev)
        lib().call("omr_image_gray_hpass", ptr(pixels), h, w, c, pixels.stride(0), ptr(bh), ptr(ch), ksh, W, ptr(tmp), cur_stream())
    else:
        tmp = pixels.view(h, w)
    lib().call("omr_image_vpass_to_float", ptr(tmp), h, W, ptr(bv), ptr(cv), ksv, H, dtype_code(out.dtype), ptr(out), out.stride(0), cur_stream())
    return H, W


def preprocess_image(raw, img_height: Optional[int] = None, dtype: torch.dtype = torch.float32, device=None) -> Tensor:
    """preprocessing.py:44-52 -> [1, H, W] on the GPU."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    px = as_hwc_uint8(raw)
    H, W = resized_width(px.shape[0], px.shape[1], img_height)
    out = torch.empty((1, H, W), dtype=dtype, device=device)
    preprocess_image_into(px.to(device, non_blocking=True), img_height, out[0])
    return out


def pad_batch_inputs(x: Sequence[Tensor], pad_value: float = 0.0, dtype: torch.dtype = torch.float32) -> Tensor:
    """preprocessing.py:55-75 on CUDA samples [1, h, w] -> [B, 1, max h, max w], right/bottom padded."""
    H = max(int(s.shape[1]) for s in x)
    W = max(int(s.shape[2]) for s in x)
    out = torch.full((len(x), 1, H, W), pad_value, dtype=dtype, device=x[0].device)
    for i, s in enumerate(x):
        out[i, :, : s.shape[1], : s.shape[2]] = s
    return out


def image_batch(raws: Sequence, img_height: Optional[int], pad_value: float = 1.0, dtype: torch.dtype = torch.float32, device=None) -> Tuple[Tensor, Tensor]:
    """preprocess_image + pad_batch_inputs(pad_value=1.0: white background, preprocessing.py:104-109) in one go: every image is
    resampled straight into its slot of the padded batch.  Returns (x [B, 1, Hmax, Wmax], widths int32 [B])."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    px = [as_hwc_uint8(r) for r in raws]
    sizes = [resized_width(p.shape[0], p.shape[1], img_height) for p in px]
    Hm, Wm = max(s[0] for s in sizes), max(s[1] for s in sizes)
    out = torch.full((len(px), 1, Hm, Wm), pad_value, dtype=dtype, device=device)
    for i, p in enumerate(px):
        preprocess_image_into(p.pin_memory().to(device, non_blocking=True), img_height, out[i, 0])
    return out, torch.tensor([s[1] for s in sizes], dtype=torch.int32)
