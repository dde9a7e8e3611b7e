"""ctypes binding of libomr_hip.so (the C ABI declared in include/omr_hip.h).

The prototypes are parsed from the header itself, so the header stays the single source of truth.
There is NO fallback: if the shared library is missing or a call fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Tuple

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(_HERE)
HEADER = os.path.join(REPO_ROOT, "include", "omr_hip.h")
LIB_PATH = os.environ.get("OMR_HIP_LIB") or os.path.join(_HERE, "libomr_hip.so")   # OMR_HIP_LIB: another build of the same ABI (kernel experiments)

F32, BF16 = 0, 1
_DTYPE_CODE = {torch.float32: F32, torch.bfloat16: BF16}

_ERRORS = {-1: "invalid argument", -2: "kernel launch failed", -3: "unsupported configuration"}


def dtype_code(dt: torch.dtype) -> int:
    try:
        return _DTYPE_CODE[dt]
    except KeyError:
        raise TypeError(f"libomr_hip supports float32 and bfloat16, got {dt}") from None


def parse_header(path: str = HEADER) -> Dict[str, Tuple[str, List[str]]]:
    """{function name: (return type, [argument C types])} for every prototype in the header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    protos: Dict[str, Tuple[str, List[str]]] = {}
    for m in re.finditer(r"\b(int|long)\s+(omr_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        types: List[str] = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                types.append(a if "*" in a else a.rsplit(" ", 1)[0])
        protos[name] = (ret, types)
    return protos


def _ctype(t: str):
    if "*" in t:
        return ctypes.c_void_p
    return {"int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "double": ctypes.c_double,
            "unsigned long long": ctypes.c_ulonglong}[t]


class _Lib:
    def __init__(self) -> None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the product path.")
        self.cdll = ctypes.CDLL(LIB_PATH)
        self.protos = parse_header()
        self.fns = {}                      # name -> bound foreign function (one dict lookup per launch on the hot path)
        for name, (ret, types) in self.protos.items():
            fn = getattr(self.cdll, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = ctypes.c_long if ret == "long" else ctypes.c_int
            fn.argtypes = [_ctype(t) for t in types]
            self.fns[name] = fn

    def call(self, name: str, *args):
        rc = self.fns[name](*args)
        if rc != 0:
            raise RuntimeError(f"libomr_hip: {name} failed: {_ERRORS.get(rc, rc)}")

    def query(self, name: str, *args) -> int:
        return int(self.fns[name](*args))


_LIB = None


def lib() -> _Lib:
    global _LIB
    if _LIB is None:
        _LIB = _Lib()
    return _LIB


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Tensors must be contiguous in the layout the kernel expects."""
    if t is None:
        return None
    return t.data_ptr()          # a plain int: ctypes converts it for a void* parameter without an intermediate object


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def cur_stream():
    """hipStream_t of torch's current stream on the current device, as an int (0 = the default stream -> NULL)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device()) or None
    return torch.cuda.current_stream().cuda_stream or None


def require_cuda(*tensors) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("omr_a2s_multimodal_transformer_amd ops run on the GPU only (HIP kernels); got a CPU tensor. "
                               "There is no CPU fallback.")
