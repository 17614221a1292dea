"""Oracle pinning, integer path: the numpy restatement of the delayed pattern against vectors
produced by the reference's Pattern.build_pattern_sequence / revert_pattern_sequence
(tests/golden/pattern.npz, made by oracle/gen_golden.py)."""
import os

import numpy as np
import pytest

from oracle import pattern as P

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "pattern.npz"))
CASES = sorted({k.split("_", 1)[1] for k in GOLD.files if k.startswith("z_")})


@pytest.mark.parametrize("case", CASES)
def test_shift_matches_reference(case):
    z = GOLD[f"z_{case}"]
    want = GOLD[f"shift_{case}"]
    assert np.array_equal(P.delayed_shift(z, 2048), want)
    assert np.array_equal(P.build_sequence_from_layout(z, 2048), want)


@pytest.mark.parametrize("case", CASES)
def test_revert_matches_reference(case):
    z = GOLD[f"z_{case}"]
    T = z.shape[2]
    sh = GOLD[f"shift_{case}"]
    assert np.array_equal(P.delayed_revert(sh, T, 2048), GOLD[f"revert_{case}"])
    assert np.array_equal(P.revert_sequence_from_layout(sh, T, 2048), GOLD[f"revert_{case}"])
    cut = np.ascontiguousarray(sh[:, :, : T + 1])
    assert np.array_equal(P.delayed_revert(cut, T, 2048), GOLD[f"revertcut_{case}"])
    assert np.array_equal(P.revert_sequence_from_layout(cut, T, 2048), GOLD[f"revertcut_{case}"])


@pytest.mark.parametrize("K,T", [(4, 1), (4, 9), (4, 800), (4, 1000), (2, 5), (8, 3)])
def test_round_trip_and_docstring_example(K, T):
    rs = np.random.RandomState(T * 10 + K)
    z = rs.randint(0, 2048, size=(3, K, T)).astype(np.int64)
    sh = P.delayed_shift(z, 2048)
    assert sh.shape == (3, K, T + K)
    assert np.array_equal(P.delayed_revert(sh, T, 2048), z)          # revert(shift(z)) == z
    # first column is all special, codebook q starts at column q+1 (codebooks_patterns.py:307-316)
    assert (sh[:, :, 0] == 2048).all()
    for q in range(K):
        assert (sh[:, q, : q + 1] == 2048).all() and np.array_equal(sh[:, q, q + 1: q + 1 + T], z[:, q])


def test_empty_prompt_shift():
    z = np.zeros((1, 4, 0), dtype=np.int64)
    assert np.array_equal(P.delayed_shift(z, 7), np.full((1, 4, 4), 7))


@pytest.mark.parametrize("N,K", [(4, 4), (5, 4), (654, 4), (10, 3)])
def test_unshift_is_inverse_of_generation_layout(N, K):
    """A span emitted step by step in delayed order un-shifts to the frames it encodes."""
    rs = np.random.RandomState(N)
    frames = rs.randint(0, 2048, size=(K, N - K)).astype(np.int64)
    span = np.full((N, K), 2048, dtype=np.int64)
    for j in range(K):
        span[j: j + N - K, j] = frames[j]
    assert np.array_equal(P.unshift_span(span), frames)
