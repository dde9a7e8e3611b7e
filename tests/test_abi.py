"""CPU-only: the C-ABI library loads and exports every symbol include/omr_hip.h declares (no compute calls)."""
import ctypes
import os

from omr_a2s_multimodal_transformer_amd import _lib


def test_header_parses_and_library_exports_every_symbol():
    protos = _lib.parse_header()
    assert len(protos) >= 25 and "omr_gemm" in protos and "omr_attn_bwd" in protos
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    cdll = ctypes.CDLL(_lib.LIB_PATH)
    for name in protos:
        assert hasattr(cdll, name), f"libomr_hip.so lacks {name}"
    assert _lib.lib().query("omr_abi_version") >= 1


def test_argument_types_follow_the_header():
    protos = _lib.parse_header()
    ret, types = protos["omr_adam"]
    assert ret == "int" and types[0] == "float* p" and "unsigned long long" not in types
    assert protos["omr_instnorm_workspace_bytes"][0] == "long"
    assert "unsigned long long" in protos["omr_dropout"][1]
