"""Late (prediction-level) fusion of the image and audio models' outputs (reference
src/multimodal/smith_waterman/smith_waterman.py:13-173 and test.py:136-160): Smith-Waterman
local alignment of the two predicted token sequences (wrapped in start / end sentinels) and the fixed policy -- agree: keep;
disagree: the more probable token; one side missing: take the other.

The alignment itself is a host function of libomr_hip.so (`omr_sw_align`, integer dynamic programme over token ids) that
restates `swalign.LocalAlignment`; `swalign` is absent from the reference tree and from this image, so the alignment is
**parity unpinned**.  What follows the alignment is the reference's own Python and is reproduced as it is written,
including `preprocess_prob`, which inserts the sentinel / gap probabilities at `position + insertions so far` (so the
probabilities behind the first gap are shifted by one -- kept, because the fused output depends on it)."""
from __future__ import annotations

import ctypes
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from ._lib import lib

START, END, GAP = "¡", "!", "-"          # the reference's sentinel / gap symbols (smith_waterman.py:5,31-32)


def sw_align(ref: Sequence[int], query: Sequence[int], match: int = 2, mismatch: int = -1, gap_penalty: int = -1,
             gap_extension_penalty: int = -1) -> Tuple[str, int, int, int]:
    """-> (ops over {'m','i','d'}, r_pos, q_pos, score) of the best local alignment (swalign.LocalAlignment.align)."""
    r = np.ascontiguousarray(ref, dtype=np.int32)
    q = np.ascontiguousarray(query, dtype=np.int32)
    ops = ctypes.create_string_buffer(len(r) + len(q) + 1)
    r_pos, q_pos, score = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    n = lib().query("omr_sw_align", ctypes.c_void_p(r.ctypes.data), len(r), ctypes.c_void_p(q.ctypes.data), len(q), match, mismatch,
                    gap_penalty, gap_extension_penalty, ctypes.cast(ops, ctypes.c_void_p), ctypes.cast(ctypes.pointer(r_pos), ctypes.c_void_p),
                    ctypes.cast(ctypes.pointer(q_pos), ctypes.c_void_p), ctypes.cast(ctypes.pointer(score), ctypes.c_void_p))
    if n < 0:
        raise RuntimeError(f"omr_sw_align failed: {n}")
    return ops.raw[:n].decode("ascii"), r_pos.value, q_pos.value, score.value


def swalign_preprocess(r: List[str], q: List[str]):
    """smith_waterman.py:13-33 without the detour through single characters: tokens -> ids shared by both sequences, wrapped
    in the start / end sentinels.  Returns (ref ids, query ids, id -> token)."""
    vocab = sorted(set(r + q))
    w2i = {w: i + 2 for i, w in enumerate(vocab)}
    i2w: Dict[int, str] = {i: w for w, i in w2i.items()}
    i2w[0], i2w[1] = START, END
    return [0] + [w2i[t] for t in r] + [1], [0] + [w2i[t] for t in q] + [1], i2w


def dump(ref: Sequence, query: Sequence, ops: str, r_pos: int, q_pos: int) -> Tuple[list, str, list]:
    """smith_waterman.py:36-94: the aligned region as three equally long sequences (query, matches, reference); matches is
    '|' (equal), '.' (different) or ' ' (one side has the gap symbol)."""
    i, j = r_pos, q_pos
    qs, ms, rs = [], "", []
    for op in ops:
        if op == "m":
            qs.append(query[j]); rs.append(ref[i])
            ms += "|" if query[j] == ref[i] else "."
            i += 1; j += 1
        elif op == "d":
            qs.append(GAP); rs.append(ref[i]); ms += " "
            i += 1
        elif op == "i":
            qs.append(query[j]); rs.append(GAP); ms += " "
            j += 1
    return qs, ms, rs


def preprocess_prob(s: Sequence, prob: List[float], sentinels=(START, END)) -> List[float]:
    """smith_waterman.py:97-117, as written (insert position = index + number of insertions so far)."""
    new_prob = list(prob)
    count = 0
    for idx, v in enumerate(s):
        if v in sentinels:
            new_prob.insert(idx + count, 1)
            count += 1
        elif v == GAP:
            new_prob.insert(idx + count, 0)
            count += 1
    return new_prob


def get_alignment(q: Sequence, m: str, r: Sequence, q_prob: List[float], r_prob: List[float]) -> list:
    """smith_waterman.py:120-159: the fusion policy, column by column."""
    out = []
    for qv, mv, rv, qp, rp in zip(q, m, r, q_prob, r_prob):
        if mv == "|":
            out.append(qv)
        elif mv == ".":
            out.append(qv if qp >= rp else rv)
        elif mv == " ":
            out.append(qv if rv == GAP else rv)
    return out


def fuse(r: List[str], r_prob: List[float], q: List[str], q_prob: List[float], match: int = 2, mismatch: int = -1,
         gap_penalty: int = -1) -> List[str]:
    """test.py:141-160 for one sample: r = image model's tokens, q = audio model's tokens (with their probabilities)."""
    rid, qid, i2w = swalign_preprocess(r, q)
    ops, r_pos, q_pos, _ = sw_align(rid, qid, match, mismatch, gap_penalty)
    rs = [i2w[t] for t in rid]
    qs = [i2w[t] for t in qid]
    qa, ma, ra = dump(rs, qs, ops, r_pos, q_pos)
    fused = get_alignment(qa, ma, ra, preprocess_prob(qa, q_prob), preprocess_prob(ra, r_prob))
    return [t for t in fused if t not in (START, END)]          # undo_swalign_preprocess (smith_waterman.py:162-173)
