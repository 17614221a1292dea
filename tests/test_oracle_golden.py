"""Oracle pinning, model path: the torch-CPU restatement of inference_tts / inference_tts_batch /
inference reproduces the reference's outputs (tokens bit-exact, head logits to rounding) on every
golden case, including seeded top-k/top-p sampling (same ATen ops, same RNG consumption)."""
import numpy as np
import pytest
import torch

from _util import MODEL_CASES, load_golden, run_oracle_case
from oracle.gen_golden import FULL_STEPS, STRIDE


@pytest.mark.parametrize("name", sorted(MODEL_CASES))
def test_oracle_matches_reference(name):
    g = load_golden(name)
    trace = []
    res, gen = run_oracle_case(name, trace=trace)
    assert np.array_equal(res.numpy(), g["res"]), "token ids differ from the reference"
    if gen is not None:
        assert np.array_equal(gen.numpy(), g["gen"])
    assert len(trace) == int(g["n_steps"])
    lg = torch.stack([t["logits"][0] for t in trace]).numpy()
    np.testing.assert_allclose(lg[:FULL_STEPS], g["logits_full"][: len(lg[:FULL_STEPS])], rtol=0, atol=2e-6)
    np.testing.assert_allclose(lg[:, :, ::STRIDE], g["logits_sub"], rtol=0, atol=2e-6)


def test_length_identities():
    # T_gen = 10*Lx - T_prompt when the terminator is muted; res = prompt + gen
    g = load_golden("tts_greedy")
    Lx, T = g["x"].shape[1], g["y"].shape[1]
    assert g["gen"].shape[2] == 10 * Lx - T
    assert np.array_equal(g["res"][0, :, :T], g["y"][0].T)
    assert int(g["n_steps"]) == g["gen"].shape[2] + 4
    # prompt already past the cap: generation ends at once, nothing is produced
    g = load_golden("tts_early_stop")
    assert g["gen"].shape[2] == 0 and int(g["n_steps"]) == 4


def test_kv_cache_equals_no_cache_and_batch_equals_single():
    a, b, c = load_golden("tts_greedy"), load_golden("tts_greedy_nokv"), load_golden("tts_batch_greedy")
    assert np.array_equal(a["res"], b["res"])
    assert np.array_equal(a["res"], c["res"])          # §8c-3
