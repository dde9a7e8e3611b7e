"""omr_a2s_multimodal_transformer_amd -- MI355X-native (gfx950) build of the encoder->decoder training / greedy-decode
hot path of mariaalfaroc/omr_a2s_multimodal_transformer.  Python host code (this package) drives hand-written HIP
kernels through the C ABI in include/omr_hip.h; there is no CPU or eager-PyTorch fallback."""
from .config import C1_TINY, C2_IMAGE, REFERENCE_DEFAULT, ModelConfig  # noqa: F401

__all__ = ["ModelConfig", "C1_TINY", "C2_IMAGE", "REFERENCE_DEFAULT"]


def __getattr__(name):
    # heavy modules are imported on first use so `import omr_a2s_multimodal_transformer_amd` stays cheap
    import importlib
    if name in ("Transformer", "MultimodalTransformer", "CrossAttention", "PositionalEncoding2D"):
        return getattr(importlib.import_module(".model", __name__), name)
    if name in ("Encoder", "ConvBlock", "DSCBlock", "DepthSepConv2D", "MixDropout", "HEIGHT_REDUCTION", "WIDTH_REDUCTION"):
        return getattr(importlib.import_module(".encoder", __name__), name)
    if name in ("Decoder", "PositionalEncoding1D"):
        return getattr(importlib.import_module(".decoder", __name__), name)
    raise AttributeError(name)
