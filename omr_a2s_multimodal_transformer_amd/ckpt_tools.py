"""Checkpoint tools: split a MultimodalTransformer checkpoint into the two unimodal
Transformer checkpoints it contains, like the reference's src/utils/split_multimodal_ckpt.py:8-117 does -- same output
file names, same hyper-parameter and state-dict key rewrites, same ModelCheckpoint-callback bookkeeping -- written against
the Lightning ``.ckpt`` layout ({"state_dict", "hyper_parameters", "callbacks", ...}) that lightning_shim.py reads and writes.

Files are only ever opened with ``torch.load(..., weights_only=True)``: nothing stored in a checkpoint is executed.
"""
from __future__ import annotations

import copy
import os
from typing import Dict, Tuple

import torch

_DROP_HP = ("mixer_type", "teacher_forcing_modality_prob")


def _other(modality: str) -> str:
    if modality not in ("image", "audio"):
        raise ValueError(f"Unknown modality: {modality}")
    return "audio" if modality == "image" else "image"


def unimodal_state_dict(state_dict: Dict[str, torch.Tensor], modality: str) -> Dict[str, torch.Tensor]:
    """Keep `<modality>_encoder.*` / `<modality>_pos_2d.*` (prefix stripped) and `decoder.*`; drop the other modality's
    encoder / positional encoding and the `cross_attn.*` mixer (split_multimodal_ckpt.py:45-72)."""
    other = _other(modality)
    kept = {k: v for k, v in state_dict.items()
            if not (k.startswith(f"{other}_encoder") or k.startswith(f"{other}_pos_2d") or k.startswith("cross_attn"))}
    # key ORDER as the reference leaves it (it pops each prefixed key and re-inserts the renamed one at the end,
    # split_multimodal_ckpt.py:56-62): untouched keys first, renamed keys after them, each group in its original order
    out: Dict[str, torch.Tensor] = {k: v for k, v in kept.items() if not k.startswith(f"{modality}_")}
    for k, v in kept.items():
        if k.startswith(f"{modality}_"):
            out[k.replace(f"{modality}_", "", 1)] = v
    return out


def unimodal_hyper_parameters(hp: dict, modality: str) -> dict:
    """max_<img|audio>_{height,width} -> max_input_{height,width}; mixer options removed (split_multimodal_ckpt.py:18-43)."""
    _other(modality)
    hp = dict(hp)
    for k in _DROP_HP:
        hp.pop(k, None)
    src = "img" if modality == "image" else "audio"
    hp["max_input_height"] = hp[f"max_{src}_height"]
    hp["max_input_width"] = hp[f"max_{src}_width"]
    for m in ("img", "audio"):
        hp.pop(f"max_{m}_height", None)
        hp.pop(f"max_{m}_width", None)
    return hp


def _retarget_callbacks(ckpt: dict, suffix: str) -> None:
    """Point the ModelCheckpoint callback state at the new file (split_multimodal_ckpt.py:9-16)."""
    for key, st in (ckpt.get("callbacks") or {}).items():
        if "ModelCheckpoint" not in str(key) or "best_model_path" not in st:
            continue
        root, ext = os.path.splitext(st["best_model_path"])
        new_path = root + f"_only_{suffix}" + ext
        st["best_model_path"] = new_path
        st["kth_best_model_path"] = new_path
        st["best_k_models"] = {new_path: st.get("best_model_score")}
        break


def split_both_ckpt_in_two(ckpt_path: str) -> Tuple[str, str]:
    """Writes `<name>_only_image_distorted<ext>` and `<name>_only_audio<ext>` next to ckpt_path and returns both paths."""
    ckpt = torch.load(ckpt_path, map_location="cpu", weights_only=True)
    root, ext = os.path.splitext(ckpt_path)
    paths = []
    for modality, suffix in (("image", "image_distorted"), ("audio", "audio")):
        one = copy.deepcopy(ckpt)
        _retarget_callbacks(one, suffix)
        one["hyper_parameters"] = unimodal_hyper_parameters(one["hyper_parameters"], modality)
        one["state_dict"] = unimodal_state_dict(one["state_dict"], modality)
        out = root + f"_only_{suffix}" + ext
        torch.save(one, out)
        paths.append(out)
    return paths[0], paths[1]


if __name__ == "__main__":
    import sys
    img, aud = split_both_ckpt_in_two(sys.argv[1])
    print(f"Image model saved at: {img}\nAudio model saved at: {aud}")
