"""Late fusion (late_fusion.py + the host function omr_sw_align of libomr_hip.so; no GPU involved).  swalign is absent, so
the alignment is parity-unpinned with respect to the package: the C++ dynamic programme is checked against the oracle's
plain-Python restatement (bit-exact: integer work) and against properties every Smith-Waterman alignment has; the fusion
policy is checked on hand-built cases, including the reference's probability-shift quirk."""
import random

import pytest

from omr_a2s_multimodal_transformer_amd import late_fusion as LF
from oracle import ref_cpu as R


def rand_pair(rng, n, vocab, edits):
    a = [rng.randrange(vocab) + 2 for _ in range(n)]
    b = list(a)
    for _ in range(edits):
        k = rng.randrange(3)
        pos = rng.randrange(len(b) + 1)
        if k == 0 and b:
            b[min(pos, len(b) - 1)] = rng.randrange(vocab) + 2
        elif k == 1 and len(b) > 1:
            del b[min(pos, len(b) - 1)]
        else:
            b.insert(pos, rng.randrange(vocab) + 2)
    return [0] + a + [1], [0] + b + [1]


def test_host_alignment_equals_the_oracle_restatement():
    rng = random.Random(4)
    for case in range(60):
        r, q = rand_pair(rng, rng.randrange(1, 40), rng.choice([3, 10, 60]), rng.randrange(0, 8))
        pen = rng.choice([(2, -1, -1), (2, -1, -2), (1, -1, -1), (3, -2, -4)])
        assert LF.sw_align(r, q, *pen) == R.sw_align(r, q, *pen), (case, r, q, pen)


def test_alignment_properties():
    rng = random.Random(5)
    for _ in range(40):
        r, q = rand_pair(rng, rng.randrange(2, 50), 20, rng.randrange(0, 6))
        ops, r_pos, q_pos, score = LF.sw_align(r, q)
        assert set(ops) <= set("mid")
        assert r_pos + ops.count("m") + ops.count("d") <= len(r) and q_pos + ops.count("m") + ops.count("i") <= len(q)
        i, j, s, run = r_pos, q_pos, 0, None          # the path's own score equals the reported optimum
        for o in ops:
            if o == "m":
                s += 2 if r[i] == q[j] else -1
                i += 1; j += 1
            else:
                s += -1
                i, j = (i + 1, j) if o == "d" else (i, j + 1)
        assert s == score and score >= 2
    same = [0, 5, 6, 7, 8, 1]
    assert LF.sw_align(same, same) == ("m" * 6, 0, 0, 12)


def test_fusion_policy_and_reference_probability_shift():
    # agreement everywhere: the fused sequence is the common one
    toks = ["*clefG2", "4c", "8d", "=1"]
    assert LF.fuse(toks, [0.9] * 4, toks, [0.8] * 4) == toks
    # one substitution: the more probable token wins, ties go to the query (audio) side (qv_prob >= rv_prob)
    r, q = ["a", "b", "c", "d"], ["a", "x", "c", "d"]
    assert LF.fuse(r, [0.9, 0.2, 0.9, 0.9], q, [0.9, 0.7, 0.9, 0.9]) == ["a", "x", "c", "d"]
    assert LF.fuse(r, [0.9, 0.7, 0.9, 0.9], q, [0.9, 0.2, 0.9, 0.9]) == ["a", "b", "c", "d"]
    assert LF.fuse(r, [0.9, 0.5, 0.9, 0.9], q, [0.9, 0.5, 0.9, 0.9]) == ["a", "x", "c", "d"]
    # a token missing on one side is taken from the other
    assert LF.fuse(["a", "b", "c", "d", "e"], [0.9] * 5, ["a", "b", "d", "e"], [0.9] * 4) == ["a", "b", "c", "d", "e"]
    # smith_waterman.py:97-117 as written: after a gap the probabilities are shifted, so a LATER disagreement is decided by
    # the neighbours' probabilities -- reproduced, not repaired
    qa = [LF.START, "a", LF.GAP, "c", "x", LF.END]
    assert LF.preprocess_prob(qa, [0.1, 0.2, 0.3]) == [1, 0.1, 0.2, 0, 0.3, 1]      # the 0 of the gap lands one place late
    assert LF.preprocess_prob([LF.START, "a", "b", LF.END], [0.4, 0.6]) == [1, 0.4, 0.6, 1]


def test_bad_arguments_are_refused():
    with pytest.raises(RuntimeError):
        LF.sw_align([], [1, 2])
