"""Sym-ER / Seq-ER edit-distance metrics (reference: src/utils/metrics.py:15-88).  MV2H (music21 + pyMV2H,
off by default in the reference, metrics.py:18) is out of scope."""
from __future__ import annotations

from typing import Dict, List, Sequence


def edit_distance(a: Sequence, b: Sequence) -> int:
    """Levenshtein distance with unit costs, two-row dynamic programme over the shorter sequence."""
    if len(a) > len(b):
        a, b = b, a
    prev = list(range(len(a) + 1))
    for i, bi in enumerate(b, start=1):
        cur = [i]
        for j, aj in enumerate(a, start=1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (aj != bi)))
        prev = cur
    return prev[-1]


def ed_counts(y_true: List[List[str]], y_pred: List[List[str]]) -> List[int]:
    """[sum of edit distances, sum of truth lengths, sequences with any error, sequences]: the four integers sym-er / seq-er
    are ratios of.  They ADD over shards of the evaluation set, so a data-parallel evaluation all-reduces them and every rank
    gets exactly the single-process metrics (metrics.py:76-88)."""
    ed_acc = length_acc = wrong = 0
    for t, h in zip(y_true, y_pred):
        ed = edit_distance(t, h)
        ed_acc += ed
        length_acc += len(t)
        wrong += ed > 0
    return [ed_acc, length_acc, wrong, len(y_pred)]


def metrics_from_counts(counts: Sequence[int]) -> Dict[str, float]:
    ed_acc, length_acc, wrong, n = (int(c) for c in counts)
    return {"sym-er": 100.0 * ed_acc / length_acc, "seq-er": 100.0 * wrong / n}


def compute_ed_metrics(y_true: List[List[str]], y_pred: List[List[str]]) -> Dict[str, float]:
    """sym-er = 100 * sum(edit distance) / sum(len(truth)); seq-er = 100 * (#sequences with any error) / #sequences."""
    return metrics_from_counts(ed_counts(y_true, y_pred))


def compute_metrics_sharded(y_true: List[List[str]], y_pred: List[List[str]], process_group=None) -> Dict[str, float]:
    """compute_metrics over an evaluation set that is sharded across the ranks of a torch.distributed group: the local counts
    are summed with ONE all-reduce of four int64 (RCCL on the GPU, gloo on the CPU).  Without an initialised group it is
    compute_metrics."""
    import torch
    import torch.distributed as dist
    counts = ed_counts(y_true, y_pred)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1:
        dev = "cuda" if dist.get_backend(process_group) == "nccl" else "cpu"
        t = torch.tensor(counts, dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=process_group)
        counts = t.tolist()
    return metrics_from_counts(counts)


def compute_metrics(y_true: List[List[str]], y_pred: List[List[str]], compute_mv2h: bool = False) -> Dict[str, float]:
    if compute_mv2h:
        raise NotImplementedError("MV2H needs music21/pyMV2H, which are outside this build's scope (SURVEY.md section 2, row 9)")
    return compute_ed_metrics(y_true=y_true, y_pred=y_pred)
