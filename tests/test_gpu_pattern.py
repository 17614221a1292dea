"""-m gpu: the HIP pattern kernels, through the C ABI, are bit-exact against the oracle's closed
forms, the reference-made golden vectors, and round-trip at BASELINE sizes."""
import os

import numpy as np
import pytest
import torch

from oracle import pattern as P

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "pattern.npz"))
CASES = sorted({k.split("_", 1)[1] for k in GOLD.files if k.startswith("z_")})


@pytest.fixture(scope="module")
def ops():
    from voicecraft_amd import engine
    assert torch.cuda.is_available()
    return engine


@pytest.mark.parametrize("case", CASES)
def test_golden_vectors(ops, case):
    z = torch.from_numpy(GOLD[f"z_{case}"]).cuda()
    T = z.shape[2]
    sh = ops.pattern_shift(z, 2048)
    assert np.array_equal(sh.cpu().numpy(), GOLD[f"shift_{case}"])
    assert np.array_equal(ops.pattern_revert(sh, T, 2048).cpu().numpy(), GOLD[f"revert_{case}"])
    cut = sh[:, :, : T + 1].contiguous()
    assert np.array_equal(ops.pattern_revert(cut, T, 2048).cpu().numpy(), GOLD[f"revertcut_{case}"])


@pytest.mark.parametrize("B,K,T", [(1, 4, 0), (1, 4, 1), (1, 4, 150), (1, 4, 800), (1, 4, 1000), (64, 4, 800), (3, 8, 77), (2, 1, 5)])
def test_against_oracle_and_round_trip(ops, B, K, T):
    rs = np.random.RandomState(B * 1000 + T)
    z = rs.randint(0, 2052, size=(B, K, T)).astype(np.int64)
    sh = ops.pattern_shift(torch.from_numpy(z).cuda(), 2048)
    assert np.array_equal(sh.cpu().numpy(), P.delayed_shift(z, 2048))
    if T > 0:
        back = ops.pattern_revert(sh, T, 2048)
        assert np.array_equal(back.cpu().numpy(), z)


@pytest.mark.parametrize("N,K", [(4, 4), (5, 4), (654, 4), (1004, 4), (11, 3)])
def test_unshift(ops, N, K):
    rs = np.random.RandomState(N)
    span = rs.randint(0, 2052, size=(N, K)).astype(np.int64)
    out = ops.pattern_unshift(torch.from_numpy(span).cuda())
    assert np.array_equal(out.cpu().numpy(), P.unshift_span(span).reshape(K, N - K))


def test_bad_arguments_raise(ops):
    with pytest.raises(AssertionError):
        ops.pattern_unshift(torch.zeros((2, 4), dtype=torch.int64).cuda())   # N < K
