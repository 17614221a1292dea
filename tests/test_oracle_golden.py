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


SAMPLED = sorted(n for n, s in MODEL_CASES.items() if "tseed" in s)


@pytest.mark.parametrize("name", SAMPLED)
def test_oracle_replays_recorded_reference_draws(name):
    """The reference's raw topk_sampling outputs (golden `draws`) pushed through the oracle's state machine
    with NO seeding: overrides, terminator / arg-max / cap conditions, silence bookkeeping and the
    best-of-N keep rule alone must reproduce the reference's result (the same replay the HIP engine is
    held to in tests/test_gpu_model.py)."""
    from _util import build_case
    from oracle.voicecraft_oracle import VoiceCraftOracle
    g = load_golden(name)
    spec, args, sd, x, x_lens, y = build_case(name)
    orc = VoiceCraftOracle(args, sd)
    kn = dict(spec["knobs"])
    torch.manual_seed(0)            # a different stream on purpose: the draws come from the fixture
    if spec["mode"] == "tts":
        res = orc.inference_tts(x, x_lens, y, forced_draws=g["draws"], **kn)[0]
    elif spec["mode"] == "tts_batch":
        res = orc.inference_tts_batch(x, x_lens, y, forced_draws=g["draws"], **kn)[0]
    else:
        res = orc.inference(x, x_lens, y, torch.tensor([spec["spans"]], dtype=torch.int64), forced_draws=g["draws"], **kn)
    assert np.array_equal(res.numpy(), g["res"])


def test_unmuted_terminator_cases_end_before_the_cap():
    """The round-2 fixtures really exercise the early-termination branches (voicecraft.py:1024, :1041-1045)."""
    for name in ("tts_eos_guard", "tts_eos_greedy", "tts_sampled_eos", "tts_batch4_sampled", "tts_oldscheme_eog"):
        g = load_golden(name)
        Lx, T = g["x"].shape[1], g["y"].shape[1]
        assert 0 < g["gen"].shape[2] < 10 * Lx - T, name
    g = load_golden("tts_eos_guard")          # terminator is the arg-max from the start: ends when the guard releases
    assert g["gen"].shape[2] == 11 and int(g["n_steps"]) == 15
    for name in ("tts_silence_sr1", "tts_silence_sr2", "tts_silence_sr3"):      # the penalty breaks the silence runs
        g = load_golden(name)
        cb0 = g["gen"][0, 0]
        sil = 1388 if name.endswith("3") else 131
        runs = np.diff(np.flatnonzero(np.concatenate(([1], (cb0 != sil).astype(int), [1])))) - 1
        assert runs.max() >= 3 and (cb0 == sil).sum() >= 10 and (cb0 != sil).sum() >= 2, name


def test_one_pass_trajectory_logits_equal_the_incremental_path():
    """oracle.tts_logits_for_trajectory (one full causal pass) against the step-by-step cached loop on a
    golden case: pins the shortcut the full-size long-context GPU tests rely on."""
    from _util import build_case
    from oracle.voicecraft_oracle import VoiceCraftOracle
    spec, args, sd, x, x_lens, y = build_case("tts_greedy_hd128")
    orc = VoiceCraftOracle(args, sd)
    trace = []
    orc.inference_tts(x, x_lens, y, trace=trace, **spec["knobs"])
    want = torch.stack([t["logits"][0] for t in trace]).numpy()
    toks = torch.stack([t["tokens"] for t in trace]).numpy()
    steps = [0, 1, 5, len(trace) // 2, len(trace) - 1]
    got = orc.tts_logits_for_trajectory(x, y, toks, steps=steps).numpy()
    np.testing.assert_allclose(got, want[steps], rtol=0, atol=2e-5)


def test_edit_one_pass_evaluation_equals_the_cached_editing_loop():
    """oracle.edit_logits_for_trajectory (one causal pass over [x ; rearranged prompt ; forced tokens], used to check
    full-size editing contexts in seconds) against the step-by-step cached loop of `inference` on the same forced
    trajectory - the loop itself is pinned to the reference by the editing goldens above."""
    import torch
    from oracle.voicecraft_oracle import VoiceCraftOracle
    from voicecraft_amd import synth
    a = synth.make_args("tiny")
    sd = synth.make_state_dict(a, seed=5)
    x, xl, y = synth.random_prompt(a, 7, 40, seed=3)
    mi = torch.tensor([[[12, 20]]], dtype=torch.int64)
    orc = VoiceCraftOracle(a, sd)
    trace = []
    orc.inference(x, xl, y, mi, top_k=1, trace=trace)
    toks = torch.stack([t["tokens"] for t in trace]).numpy()
    want = torch.stack([t["logits"][0] for t in trace])
    got = orc.edit_logits_for_trajectory(x, y, mi, toks, steps=list(range(len(trace))))
    assert got.shape == want.shape
    assert float((got - want).abs().max()) <= 2e-4 * float(want.abs().max().clamp(max=1e3)) + 1e-4
