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
    assert _lib.lib().query("omr_abi_version") >= 2


def test_every_exported_entry_point_is_declared_in_the_header():
    """The header is the single source of truth: nothing `extern "C"` may exist in the library without a prototype (and a
    citation) in include/omr_hip.h, and the struct mirrors used over ctypes have the C layout's size."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if " T omr_" in line}
    protos = set(_lib.parse_header())
    assert exported == protos, (sorted(exported - protos), sorted(protos - exported))
    from omr_a2s_multimodal_transformer_amd.decoder import _DecodeDesc
    from omr_a2s_multimodal_transformer_amd.kernels import _DecodeLinearArgs, _DwProblem
    assert ctypes.sizeof(_DecodeDesc) == 12 * 4 + 11 * 8 + 4 * 8 and ctypes.sizeof(_DwProblem) == 80
    assert ctypes.sizeof(_DecodeLinearArgs) == 12 * 4 + 2 * 4 + 24 * 8        # + w8, w8_scale (round 3)


def test_argument_types_follow_the_header():
    protos = _lib.parse_header()
    ret, types = protos["omr_adam"]
    assert ret == "int" and types[0] == "float* p" and "unsigned long long" not in types
    assert protos["omr_instnorm_workspace_bytes"][0] == "long"
    assert "unsigned long long" in protos["omr_dropout"][1]
