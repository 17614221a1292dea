"""-m gpu: the device sampler as a DISTRIBUTION.

`vc_debug_sample` draws n tokens from one logits row through the product code path (temperature,
top-k threshold search, top-p mass search, Philox inverse-CDF draw: vc_tokens.hip filter_draw).  The
expected distribution is softmax of the ORACLE's filter (oracle.filter_top_k_top_p, pinned against the
reference's top_k_top_p_filtering in tests/test_oracle_sampler.py) - not a re-statement inside the test.
Checked: (a) no token outside the oracle's support is ever drawn, every token of the support that should
have appeared did; (b) Pearson chi-square of the counts against the expected probabilities stays below
the 1 - 1e-6 quantile (bins with expectation < 8 pooled), which a sampler with a mis-applied temperature,
a biased CDF walk or wrong filter thresholds fails by orders of magnitude (verified below with a
deliberately wrong expectation).
"""
import numpy as np
import pytest
import torch
from scipy import stats

from oracle.voicecraft_oracle import filter_top_k_top_p

pytestmark = pytest.mark.gpu

N_DRAWS = 60000
V = 2052


def make_row(seed, scale):
    rs = np.random.RandomState(seed)
    lg = (rs.standard_normal(V) * scale).astype(np.float32)
    lg[5] = lg[9]                                           # an exact tie
    return torch.from_numpy(lg)


def expected_probs(row, top_k, top_p, temperature):
    lg = row.clone().unsqueeze(0)
    if temperature != 1.0:
        lg = lg / temperature
    lg = filter_top_k_top_p(lg, top_k=top_k, top_p=top_p)
    return torch.softmax(lg.double(), dim=-1)[0].numpy(), lg[0]


def top_p_boundary_is_clear(row, top_k, top_p, temperature, eps=2e-4):
    """True when no cumulative mass of the sorted row lies within eps of top_p: then the kept set does not
    depend on the summation order and the oracle's support is THE support."""
    if top_p >= 1.0:
        return True
    lg = row.clone().unsqueeze(0) / temperature
    lg = filter_top_k_top_p(lg, top_k=top_k, top_p=1.0)
    cum = torch.cumsum(torch.softmax(torch.sort(lg[0], descending=True)[0].double(), -1), -1)
    return bool((torch.abs(cum - top_p) > eps).all())


def chi_square(counts, probs, n):
    exp = probs * n
    big = exp >= 8.0
    obs_b, exp_b = counts[big].astype(np.float64), exp[big]
    rest_obs, rest_exp = counts[~big].sum(), exp[~big].sum()
    if rest_exp > 0:
        obs_b, exp_b = np.append(obs_b, rest_obs), np.append(exp_b, rest_exp)
    stat = float(((obs_b - exp_b) ** 2 / exp_b).sum())
    dof = len(exp_b) - 1
    return stat, dof


CASES = [
    # (top_k, top_p, temperature, logits scale)
    (40, 1.0, 1.0, 1.0),          # the benchmark's setting
    (0, 0.8, 1.0, 2.0),           # nucleus only
    (40, 0.9, 0.7, 1.0),          # all three knobs
    (-100, 1.0, 1.0, 1.5),        # the reference default: no filter at all
    (8, 0.6, 1.3, 2.5),
    (1, 1.0, 1.0, 1.0),           # greedy
]


@pytest.mark.parametrize("top_k,top_p,temperature,scale", CASES)
def test_device_sampler_distribution(top_k, top_p, temperature, scale):
    from voicecraft_amd.engine import debug_sample
    seed = 11
    row = make_row(seed, scale)
    while not top_p_boundary_is_clear(row, top_k, top_p, temperature):
        seed += 1
        row = make_row(seed, scale)
    probs, filt = expected_probs(row, top_k, top_p, temperature)
    toks = debug_sample(row.cuda(), N_DRAWS, top_k=top_k, top_p=top_p, temperature=temperature, seed=2024).cpu().numpy()
    assert toks.min() >= 0 and toks.max() < V
    counts = np.bincount(toks, minlength=V)
    support = np.isfinite(filt.numpy())
    assert counts[~support].sum() == 0, f"{counts[~support].sum()} draws outside the oracle's support"
    if top_k > 0 and top_p >= 1.0:
        assert support.sum() >= min(top_k, V)               # ties at the k-th value survive (voicecraft.py:38-44)
    # every token whose expected count is >= 30 must have shown up (P(miss) < 1e-13)
    assert (counts[probs * N_DRAWS >= 30] > 0).all()
    if support.sum() == 1:
        assert counts[support][0] == N_DRAWS
        return
    stat, dof = chi_square(counts, probs, N_DRAWS)
    limit = stats.chi2.ppf(1 - 1e-6, dof)
    assert stat < limit, f"chi-square {stat:.1f} over {dof} dof exceeds {limit:.1f}"
    # the test has teeth: the same counts against the distribution of a slightly wrong temperature fail
    wrong, _ = expected_probs(row, top_k, top_p, temperature * 1.15)
    if np.isfinite(wrong).all() and (wrong > 0).sum() == (probs > 0).sum():
        stat_w, dof_w = chi_square(counts, wrong, N_DRAWS)
        assert stat_w > stats.chi2.ppf(1 - 1e-6, dof_w), "a 15 % temperature error went unnoticed"


def test_draws_are_seeded_and_streams_differ():
    from voicecraft_amd.engine import debug_sample
    row = make_row(3, 1.0).cuda()
    a = debug_sample(row, 4096, top_k=40, seed=1).cpu().numpy()
    b = debug_sample(row, 4096, top_k=40, seed=1).cpu().numpy()
    c = debug_sample(row, 4096, top_k=40, seed=2).cpu().numpy()
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert len(np.unique(a)) > 20                            # not one value repeated


def test_top_k_fallback_when_one_lane_owns_many_of_the_largest():
    """The fast top-k path keeps 5 candidates per lane (lane = index mod 64) and proves its threshold with one count
    over the row; when a lane owns more than 5 of the k largest logits the proof fails and the full search runs.
    Rows built to force that (6 and 9 of the top values on one lane, with ties) must still give exactly the
    reference's kept set and distribution."""
    from voicecraft_amd.engine import debug_sample
    for n_same_lane, seed in ((6, 1), (9, 2)):
        rs = np.random.RandomState(seed)
        lg = (rs.standard_normal(V) * 0.5).astype(np.float32)
        idx = 7 + 64 * np.arange(n_same_lane)                 # all on lane 7
        lg[idx] = 4.0 + 0.1 * np.arange(n_same_lane)
        lg[idx[1]] = lg[idx[0]]                               # a tie among the largest
        row = torch.from_numpy(lg)
        for top_k in (8, 40):
            probs, filt = expected_probs(row, top_k, 1.0, 1.0)
            toks = debug_sample(row.cuda(), N_DRAWS, top_k=top_k, seed=77 + top_k).cpu().numpy()
            counts = np.bincount(toks, minlength=V)
            support = np.isfinite(filt.numpy())
            assert counts[~support].sum() == 0 and (counts[probs * N_DRAWS >= 30] > 0).all()
            stat, dof = chi_square(counts, probs, N_DRAWS)
            assert stat < stats.chi2.ppf(1 - 1e-6, dof), (n_same_lane, top_k, stat, dof)
