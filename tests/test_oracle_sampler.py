"""Oracle pinning: top-k / top-p filter and the seeded categorical draw against outputs of the
reference's top_k_top_p_filtering / topk_sampling (tests/golden/sampler.npz)."""
import os

import numpy as np
import torch

from oracle.voicecraft_oracle import draw, filter_top_k_top_p

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "sampler.npz"))


def test_filters_match_reference():
    lg = torch.from_numpy(GOLD["logits"])
    for name, (k, p) in {"k5": (5, 1.0), "k40": (40, 1.0), "p08": (0, 0.8), "k20p06": (20, 0.6), "none": (-100, 1.0)}.items():
        got = filter_top_k_top_p(lg.clone(), top_k=k, top_p=p).numpy()
        assert np.array_equal(got, GOLD[f"filt_{name}"]), name


def test_known_answers():
    # ties at the k-th value survive; the first token past top_p survives
    got = filter_top_k_top_p(torch.tensor([[1., 3., 3., 2., 0.]]), top_k=2).numpy()
    assert np.array_equal(got, GOLD["kat_topk"])
    assert np.array_equal(got, np.array([[-np.inf, 3, 3, -np.inf, -np.inf]], dtype=np.float32))
    got = filter_top_k_top_p(torch.log(torch.tensor([[.5, .3, .15, .05]])), top_p=0.8).numpy()
    assert np.array_equal(got, GOLD["kat_topp"])
    assert np.isinf(got[0, 3]) and not np.isinf(got[0, :3]).any()


def test_seeded_draws_match_reference():
    lg = torch.from_numpy(GOLD["logits"])
    torch.manual_seed(5)
    got = torch.stack([draw(lg.clone(), top_k=10, top_p=0.9, temperature=0.7) for _ in range(8)]).numpy()
    assert np.array_equal(got, GOLD["draws_seed5"])
