"""Model configuration (HF-style to_dict/from_dict/save/load surface).

The reference hard-codes its hyper-parameters (decoder.py:61-68: embedding_dim=256,
ff_dim=256, nhead=4, num_transformer_layers=8, dropout_p=0.1; encoder.py:251 dropout=0.5;
model.py:95-96 num_channels=256).  Defaults here equal those values; BASELINE.json's
"2-layer d_model=128" and "6-layer d_model=256" configs are reached by overriding them.
"""
from __future__ import annotations

import json
from dataclasses import asdict, dataclass, fields
from typing import Any, Dict


@dataclass
class ModelConfig:
    d_model: int = 256          # decoder.py:61 embedding_dim / encoder.py:267 last DSCBlock width
    nhead: int = 4              # decoder.py:66
    ff_dim: int = 256           # decoder.py:64
    num_layers: int = 8         # decoder.py:67
    dropout: float = 0.1        # decoder.py:65, model.py:29 (PE dropouts)
    encoder_dropout: float = 0.5  # encoder.py:251
    compute_dtype: str = "fp32"   # "fp32" (parity mode) or "bf16" (fp32 master weights + Adam)
    fp8_decode: bool = False      # BASELINE config 5 (extension): greedy / beam decode with fp8 (e4m3) MFMA weights in the decoder

    def to_dict(self) -> Dict[str, Any]:
        return asdict(self)

    @classmethod
    def from_dict(cls, d: Dict[str, Any]) -> "ModelConfig":
        names = {f.name for f in fields(cls)}
        return cls(**{k: v for k, v in d.items() if k in names})

    def save_pretrained(self, path: str) -> None:
        with open(path, "w") as f:
            json.dump(self.to_dict(), f, indent=2)

    @classmethod
    def from_pretrained(cls, path: str) -> "ModelConfig":
        with open(path) as f:
            return cls.from_dict(json.load(f))


# BASELINE.json configs (SURVEY.md section 8 "Configs")
C1_TINY = ModelConfig(d_model=128, nhead=4, ff_dim=128, num_layers=2)
C2_IMAGE = ModelConfig(d_model=256, nhead=4, ff_dim=256, num_layers=6, compute_dtype="bf16")
REFERENCE_DEFAULT = ModelConfig()
