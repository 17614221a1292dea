"""-m gpu: the HIP engine against the oracle / reference-made golden vectors, through the C ABI.

fp32 (exact) mode : greedy token ids bit-equal to the reference's on every golden case
                    (TTS, best-of-N, editing with 1/2/3 spans, both special-token schemes) and
                    per-step head logits within 1e-3 absolute.
bf16 mode         : teacher-forced on the reference's own trajectory; per-step head logits within
                    2e-2 relative L2 (SURVEY.md §8c tolerance) and top-1 agreement wherever the
                    oracle's top-1/top-2 margin exceeds 4x the observed max error.
"""
import numpy as np
import pytest
import torch

from _util import MODEL_CASES, build_case, load_golden, run_oracle_case

pytestmark = pytest.mark.gpu

GREEDY = [n for n, s in MODEL_CASES.items() if s["knobs"]["top_k"] == 1 and s["knobs"]["kvcache"] == 1]


def rel_l2(got, want):
    """Per-step relative L2 error of the head logits.  The synthetic checkpoints mute the terminator
    with a -1e4 bias; those columns would swamp the norm, so they are left out."""
    live = np.abs(want) < 1e3
    num = np.sqrt((((got - want) * live) ** 2).reshape(len(want), -1).sum(1))
    den = np.sqrt(((want * live) ** 2).reshape(len(want), -1).sum(1))
    return num / den


def assert_free_running_frames(gen, a, n_frames):
    """What the state machine guarantees about generated frames with a muted terminator (synthetic weights):
    the length set by the reference's cap, ids inside the vocabulary, never the terminator (muted / forced only
    on the dropped tail columns) and never `eog` when `eos` ends the utterance (voicecraft.py:1091-1093); on
    codebooks >= 1 never `empty` (:1021-1023).  `empty` on codebook 0 and the pad id are ordinary (if unlikely)
    samples for random weights - the reference does not mask them either."""
    V = a.audio_vocab_size + a.n_special
    assert gen.shape == (1, a.n_codebooks, n_frames), gen.shape
    assert gen.min() >= 0 and gen.max() < V
    term = a.eos if a.eos > 0 else a.eog
    assert not (gen == term).any()
    if a.eos > 0:
        assert not (gen == a.eog).any()
    assert not (gen[:, 1:] == a.empty_token).any()


def make_engine(name, dtype, **kw):
    from voicecraft_amd.engine import VoiceCraftEngine
    spec, args, sd, x, x_lens, y = build_case(name)
    eng = VoiceCraftEngine(args, sd, device="cuda:0", dtype=dtype, max_seqs=4, max_positions=512, **kw)
    return eng, spec, x.cuda(), x_lens.cuda(), y.cuda()


def engine_run(eng, spec, x, x_lens, y, forced=None, logit_steps=0, seed=1, forced_mode="tokens"):
    kn = dict(spec["knobs"])
    if spec["mode"] == "tts":
        out = eng.inference_tts(x, x_lens, y, **kn, _forced=forced, _logit_steps=logit_steps, _seed=seed, _forced_mode=forced_mode)
        return (out[0], out[2]) if logit_steps else (out[0], None)
    if spec["mode"] == "tts_batch":
        out = eng.inference_tts_batch(x, x_lens, y, **kn, _seed=seed, _forced=forced, _forced_mode=forced_mode)
        return out[0], None
    mi = torch.tensor([spec["spans"]], dtype=torch.int64)
    out = eng.inference(x, x_lens, y, mi, **kn, _forced=forced, _logit_steps=logit_steps, _seed=seed, _forced_mode=forced_mode)
    return (out[0], out[1]) if logit_steps else (out, None)


SAMPLED = sorted(n for n, s_ in MODEL_CASES.items() if "tseed" in s_)


@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("name", SAMPLED)
def test_fp32_replay_of_reference_draws_equals_reference(name, graph):
    """Sampled reference runs (top-k / top-p / temperature, UN-muted terminator, best-of-N with four
    different samples): the reference's recorded raw draws are fed to the device state machine
    (forced_mode = draws), which must turn them into exactly the reference's result - the K-1 `empty`
    overrides, the sampled / arg-max / length-cap terminator, the min-length guard, the silence
    bookkeeping, the staggered EOG tail, the keep-LAST rule of inference_tts_batch and the drop of the
    other samples (models/voicecraft.py:1018-1067, :1269-1325, :718-787)."""
    g = load_golden(name)
    eng, spec, x, x_lens, y = make_engine(name, "fp32", use_graph=graph)
    res, _ = engine_run(eng, spec, x, x_lens, y, forced=g["draws"], forced_mode="draws", seed=99)
    assert list(res.shape) == list(g["res"].shape), (res.shape, g["res"].shape)
    assert np.array_equal(res.cpu().numpy(), g["res"])


def test_best_of_n_keeps_the_last_terminating_sample():
    """Hand-made draws for inference_tts_batch(batch_size=4): samples 1 and 2 both emit the terminator at
    step 12 (sample 0 and 3 do not): the reference keeps the LAST of them (`keep = b` overwritten in order,
    models/voicecraft.py:1296-1302).  Checked against the oracle replaying the same draws, fp32."""
    from oracle.voicecraft_oracle import VoiceCraftOracle
    name = "tts_batch4_sampled"
    eng, spec, x, x_lens, y = make_engine(name, "fp32", use_graph=True)
    _, args, sd, xc, xlc, yc = build_case(name)
    K, B, n = 4, 4, 24
    rs = np.random.RandomState(5)
    draws = rs.randint(0, 2048, size=(n, B, K)).astype(np.int64)
    draws[12, 1, 0] = draws[12, 2, 0] = int(args.eos)
    kn = dict(spec["knobs"])
    want = VoiceCraftOracle(args, sd).inference_tts_batch(xc, xlc, yc, forced_draws=draws, **kn)[0].numpy()
    got = eng.inference_tts_batch(x, x_lens, y, **kn, _forced=draws, _forced_mode="draws", _seed=1)[0].cpu().numpy()
    assert got.shape == want.shape and np.array_equal(got, want)
    T = yc.shape[1]
    assert got.shape[2] == T + 12 and np.array_equal(got[0, 0, T:], draws[:12, 2, 0])     # sample 2's trajectory


@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("name", GREEDY)
def test_fp32_greedy_tokens_equal_reference(name, graph):
    g = load_golden(name)
    eng, spec, x, x_lens, y = make_engine(name, "fp32", use_graph=graph)
    res, _ = engine_run(eng, spec, x, x_lens, y)
    assert res.shape == tuple(g["res"].shape) or list(res.shape) == list(g["res"].shape), (res.shape, g["res"].shape)
    assert np.array_equal(res.cpu().numpy(), g["res"]), "generated token ids differ from the reference"


@pytest.mark.parametrize("name", ["tts_greedy", "tts_greedy_hd128", "edit_2span", "tts_oldscheme"])
def test_fp32_logits_close_to_oracle(name):
    trace = []
    run_oracle_case(name, trace=trace)
    want = torch.stack([t["logits"][0] for t in trace]).numpy()
    forced = torch.stack([t["tokens"] for t in trace]).numpy()
    eng, spec, x, x_lens, y = make_engine(name, "fp32", use_graph=False)
    _, lg = engine_run(eng, spec, x, x_lens, y, forced=forced, logit_steps=len(trace))
    got = lg.cpu().numpy()
    err = np.abs(got - want).max(axis=(1, 2))
    assert err.max() <= 1e-3, f"fp32 logits max|d| per step: {err}"


@pytest.mark.parametrize("name", ["tts_greedy", "tts_greedy_hd128", "tts_sampled", "edit_2span", "edit_3span_edges", "edit_oldscheme"])
def test_bf16_teacher_forced_logits(name):
    trace = []
    res_o, _ = run_oracle_case(name, trace=trace)
    want = torch.stack([t["logits"][0] for t in trace]).numpy()            # [steps,K,V] fp32 oracle
    forced = torch.stack([t["tokens"] for t in trace]).numpy()
    eng, spec, x, x_lens, y = make_engine(name, "bf16", use_graph=True)
    res, lg = engine_run(eng, spec, x, x_lens, y, forced=forced, logit_steps=len(trace))
    # teacher forcing replays the reference trajectory, so the assembled output must be identical
    assert np.array_equal(res.cpu().numpy(), res_o.numpy())
    got = lg.cpu().numpy()
    rel = rel_l2(got, want)
    assert rel.max() <= 2e-2, f"bf16 relative L2 error per step: max {rel.max():.4f}"
    max_err = np.abs(got - want).max()
    srt = np.sort(want, axis=-1)
    margin = srt[..., -1] - srt[..., -2]
    clear = margin > 4 * max_err
    assert (np.argmax(got, -1)[clear] == np.argmax(want, -1)[clear]).all()


def test_sampling_is_seeded_and_in_range():
    eng, spec, x, x_lens, y = make_engine("tts_sampled", "bf16", use_graph=True)
    kn = dict(spec["knobs"])
    a = eng.inference_tts(x, x_lens, y, **kn, _seed=7)[1].cpu().numpy()
    b = eng.inference_tts(x, x_lens, y, **kn, _seed=7)[1].cpu().numpy()
    c = eng.inference_tts(x, x_lens, y, **kn, _seed=8)[1].cpu().numpy()
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert_free_running_frames(a, eng.args, 10 * x.shape[1] - y.shape[1])


@pytest.mark.parametrize("top_k,top_p,temperature", [(3, 1.0, 1.0), (0, 0.35, 1.0), (8, 0.6, 0.7)])
def test_sampled_tokens_stay_inside_the_filter_support(top_k, top_p, temperature):
    """The device sampler (top-k threshold search, top-p mass search, inverse-CDF draw) against the
    reference's filter semantics (models/voicecraft.py:26-68) applied to the engine's OWN logits of the
    same step: every free-running codebook-0 token must lie in the set top_k_top_p_filtering keeps."""
    eng, spec, x, x_lens, y = make_engine("tts_sampled", "fp32", use_graph=True)
    n = 40
    res, gen, lg = eng.inference_tts(x, x_lens, y, top_k=top_k, top_p=top_p, temperature=temperature,
                                     stop_repetition=-1, _seed=123, _logit_steps=n)
    lg = lg.cpu()
    T = y.shape[1]
    toks = res[0, 0, T:].cpu().numpy()                  # codebook 0 is not delayed: frame T+s comes from step s
    steps = min(n, len(toks) - 1)
    assert steps >= 20
    args = eng.args
    eos = int(args.eos)
    term = eos if eos > 0 else int(args.eog)            # eog_inference (voicecraft.py:1016)
    for s_ in range(steps):
        logits = lg[s_, 0].clone()
        if eos > 0:
            logits[int(args.eog)] = -10000.0            # :1091-1093 (eog is never generated when eos ends the utterance)
        if s_ <= int(args.encodec_sr) // 5:
            logits[term] = -10000.0                     # :1024 (cur_num_gen == step while codebook 0 runs free)
        logits = logits / temperature
        if top_k > 0:
            kth = torch.topk(logits, min(max(top_k, 1), logits.numel()))[0][-1]
            logits[logits < kth] = -float("inf")
        if top_p < 1.0:
            sl, si = torch.sort(logits, descending=True)
            cum = torch.cumsum(torch.softmax(sl, dim=-1), dim=-1)
            rm = cum > top_p * (1.0 + 1e-4)      # the device sums exp() in another order: a hair of slack at the boundary
            rm[1:] = rm[:-1].clone(); rm[0] = False
            logits[si[rm]] = -float("inf")
        keep = set(torch.nonzero(torch.isfinite(logits)).flatten().tolist())
        assert int(toks[s_]) in keep, (s_, int(toks[s_]), len(keep))


def test_graph_equals_eager_bf16():
    eng, spec, x, x_lens, y = make_engine("tts_sampled", "bf16", use_graph=True)
    kn = dict(spec["knobs"])
    a = eng.inference_tts(x, x_lens, y, **kn, _seed=3)[0].cpu().numpy()
    eng.use_graph = False
    b = eng.inference_tts(x, x_lens, y, **kn, _seed=3)[0].cpu().numpy()
    assert np.array_equal(a, b)


def test_multi_utterance_equals_single_fp32():
    """vc_tts_multi: every row of a ragged batch equals its own single-utterance run."""
    from voicecraft_amd import synth
    eng, spec, x, x_lens, y = make_engine("tts_greedy", "fp32", use_graph=True)
    args = eng.args
    prompts = [synth.random_prompt(args, Lx, T, seed=s) for Lx, T, s in [(6, 21, 11), (4, 9, 21), (8, 33, 22)]]
    outs = eng.inference_tts_multi([p[0][0] for p in prompts], [p[2][0] for p in prompts], top_k=1, stop_repetition=3)
    for (xx, xl, yy), (res, gen) in zip(prompts, outs):
        single = eng.inference_tts(xx.cuda(), xl.cuda(), yy.cuda(), top_k=1, stop_repetition=3)[0]
        assert np.array_equal(res.cpu().numpy(), single.cpu().numpy())
    assert np.array_equal(outs[0][0].cpu().numpy(), load_golden("tts_greedy")["res"])


@pytest.mark.parametrize("B,dtype", [(8, "fp32"), (12, "fp32"), (16, "fp32"), (12, "bf16"), (24, "fp32")])
def test_wide_batch_decode_equals_per_utterance_oracle(B, dtype):
    """8 / 12 / 16 / 24 / 40 utterances decoded together on a 16-head model (more than 16 = more than one MFMA row
    tile: the decode pass then runs on the block GEMM, the heads 16 rows at a time): the batched-decode kernel variants
    (per-row LayerNorm launch + plain prologue with 16 X slots, attention with 4 and 2 splits and the
    matching merge prologues, multi-row attention grid) against one oracle run per utterance.
    fp32: greedy tokens bit-equal.  bf16: shapes, token range and the first generated token (codebook 0 of the
    first new frame: it comes from the shared prefill path) equal the single-utterance run; later frames may legitimately differ,
    the batched attention sums its splits in another order and the synthetic logits are flat."""
    from oracle.voicecraft_oracle import VoiceCraftOracle
    from voicecraft_amd import synth
    from voicecraft_amd.engine import VoiceCraftEngine
    a = synth.make_args("tiny_h16")
    sd = synth.make_state_dict(a, seed=4)
    prompts = [synth.random_prompt(a, 4 + (u % 5), 9 + 3 * (u % 7), seed=300 + u) for u in range(B)]
    eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype=dtype, max_seqs=B, max_positions=256)
    outs = eng.inference_tts_multi([p[0][0] for p in prompts], [p[2][0] for p in prompts], top_k=1, stop_repetition=3)
    orc = VoiceCraftOracle(a, sd) if dtype == "fp32" else None
    for u, ((xx, xl, yy), (res, gen)) in enumerate(zip(prompts, outs)):
        got = res.cpu().numpy()
        if B > 16 and u % 3:            # wide batches: every third utterance against the (slow) CPU oracle
            continue
        if orc is not None:
            want = orc.inference_tts(xx, xl, yy, top_k=1, stop_repetition=3)[0].numpy()
            assert got.shape == want.shape and np.array_equal(got, want)
        else:
            want = eng.inference_tts(xx.cuda(), xl.cuda(), yy.cuda(), top_k=1, stop_repetition=3)[0].cpu().numpy()
            T = yy.shape[1]
            assert got.shape == want.shape and got.min() >= 0 and got.max() < a.audio_vocab_size + a.n_special
            assert np.array_equal(got[:, :, :T], want[:, :, :T]) and np.array_equal(got[:, 0, T], want[:, 0, T])


def test_full_size_giga830M_logits_against_oracle():
    """BASELINE-size parity (d=2048, 16 layers, 16 heads of 128): the kernel variants only this shape
    reaches - 16 k-tiles per wave, 12-channel QKV tiles on a 512-workgroup grid, split-K 2 and 4 slabs,
    8 attention splits, a 2-pass prefill of 191 rows - against the CPU oracle on the first decode steps.
    fp32 mode: head logits within 1e-3 absolute and the same arg-max; bf16 mode: relative L2 <= 2e-2."""
    from oracle.voicecraft_oracle import VoiceCraftOracle
    from voicecraft_amd import synth
    from voicecraft_amd.engine import VoiceCraftEngine
    a = synth.make_args("giga830M")
    sd = synth.make_state_dict(a, seed=0, fast=True)
    x, xl, y = synth.random_prompt(a, 40, 150, seed=5)
    n = 6
    trace = []
    torch.set_num_threads(min(16, torch.get_num_threads() or 1) or 1)
    VoiceCraftOracle(a, sd).inference_tts(x, xl, y, top_k=1, stop_repetition=3, trace=trace, max_steps=n)
    trace = [t for t in trace if "tokens" in t and "logits" in t]     # the cut-off step carries no tokens
    assert len(trace) >= 4
    want = torch.stack([t["logits"][0] for t in trace]).numpy()
    forced = torch.stack([t["tokens"] for t in trace]).numpy()
    for dtype in ("fp32", "bf16"):
        eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype=dtype, max_seqs=1, max_positions=512)
        out = eng.inference_tts(x.cuda(), xl.cuda(), y.cuda(), top_k=1, stop_repetition=3, _forced=forced, _logit_steps=len(trace))
        got = out[2].cpu().numpy()
        if dtype == "fp32":
            live = np.abs(want) < 1e3
            assert np.abs((got - want) * live).max() <= 1e-3
            assert np.array_equal((got * live).argmax(-1), (want * live).argmax(-1))
        else:
            assert rel_l2(got, want).max() <= 2e-2
        del eng
        torch.cuda.empty_cache()


def test_full_size_giga830M_batch8_equals_single_fp32():
    """BASELINE-size batched decode (8 utterances, d=2048): per-row LayerNorm launch + 16-slot plain prologue,
    4-split attention and its (4 x 2) merge prologue, two weight chunks per workgroup (fp32: K/16 = 128
    k-tiles) - every utterance must reproduce its own single-utterance run token for token (fp32)."""
    from voicecraft_amd import synth
    from voicecraft_amd.engine import VoiceCraftEngine
    a = synth.make_args("giga830M")
    sd = synth.make_state_dict(a, seed=0, fast=True)
    eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype="fp32", max_seqs=8, max_positions=256)
    prompts = [synth.random_prompt(a, 4 + (u % 3), 10 + 4 * (u % 5), seed=700 + u) for u in range(8)]
    outs = eng.inference_tts_multi([p[0][0] for p in prompts], [p[2][0] for p in prompts], top_k=1, stop_repetition=3)
    for (xx, xl, yy), (res, gen) in zip(prompts, outs):
        single = eng.inference_tts(xx.cuda(), xl.cuda(), yy.cuda(), top_k=1, stop_repetition=3)[0]
        assert res.shape == single.shape and np.array_equal(res.cpu().numpy(), single.cpu().numpy())


def test_full_size_giga830M_two_span_edit_equals_oracle_fp32():
    """BASELINE-size editing: rearranged prompt with two masked spans, the 3-row span switch (per-row
    LayerNorm launch + 3-row plain prologue, 8-split attention over 3 rows) - greedy tokens equal the
    CPU oracle's, fp32 mode."""
    from oracle.voicecraft_oracle import VoiceCraftOracle
    from voicecraft_amd import synth
    from voicecraft_amd.engine import VoiceCraftEngine
    a = synth.make_args("giga830M")
    sd = synth.make_state_dict(a, seed=0, fast=True)
    x, xl, y = synth.random_prompt(a, 5, 30, seed=41)
    mi = torch.tensor([[[5, 10], [18, 24]]], dtype=torch.int64)
    torch.set_num_threads(min(16, torch.get_num_threads() or 1) or 1)
    want = VoiceCraftOracle(a, sd).inference(x, xl, y, mi, top_k=1, stop_repetition=3)
    want = (want[0] if isinstance(want, tuple) else want).numpy()
    eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype="fp32", max_seqs=1, max_positions=256)
    got = eng.inference(x.cuda(), xl.cuda(), y.cuda(), mi, top_k=1, stop_repetition=3).cpu().numpy()
    assert got.shape == want.shape and np.array_equal(got, want)


def test_input_validation_mirrors_reference_asserts():
    eng, spec, x, x_lens, y = make_engine("tts_greedy", "bf16")
    with pytest.raises(AssertionError):
        eng.inference_tts(x[0], x_lens, y)                      # x.ndim != 2   (voicecraft.py:939)
    with pytest.raises(AssertionError):
        eng.inference_tts(x, x_lens, y[:, :, :3])               # wrong K       (voicecraft.py:945)
    with pytest.raises(AssertionError):
        eng.inference(x, x_lens, y, torch.tensor([[1, 2]]))     # mask_interval shape (voicecraft.py:607)
    bad = y.clone(); bad[0, 0, 0] = 5000
    with pytest.raises(AssertionError):
        eng.inference_tts(x, x_lens, bad)                       # token id outside the vocabulary
    with pytest.raises(IndexError):
        eng.inference(x, x_lens, y, torch.tensor([[[0, 3]]]))   # span at frame 0: the reference raises too


@pytest.mark.parametrize("Lx,T", [(30, 180), (25, 103), (40, 260)])
def test_fp32_multi_pass_prefill_equals_oracle(Lx, T):
    """Prompts longer than one 128-row prefill pass (and one that ends exactly on a pass boundary:
    25 + 103 = 128 rows): greedy tokens must still equal the CPU oracle's, fp32 mode."""
    from oracle.voicecraft_oracle import VoiceCraftOracle
    from voicecraft_amd import synth
    from voicecraft_amd.engine import VoiceCraftEngine
    a = synth.make_args("tiny")
    sd = synth.make_state_dict(a, seed=9)
    x, xl, y = synth.random_prompt(a, Lx, T, seed=100 + Lx)
    want = VoiceCraftOracle(a, sd).inference_tts(x, xl, y, top_k=1, stop_repetition=3)[0].numpy()
    eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype="fp32", max_seqs=1, max_positions=1024)
    got = eng.inference_tts(x.cuda(), xl.cuda(), y.cuda(), top_k=1, stop_repetition=3)[0].cpu().numpy()
    assert got.shape == want.shape and np.array_equal(got, want)


def test_bf16_long_edit_prefill_teacher_forced():
    """Editing with a 300-frame utterance (prefill of ~330 rows = 3 passes), bf16, teacher-forced logits."""
    from oracle.voicecraft_oracle import VoiceCraftOracle
    from voicecraft_amd import synth
    from voicecraft_amd.engine import VoiceCraftEngine
    a = synth.make_args("tiny128")
    sd = synth.make_state_dict(a, seed=10)
    x, xl, y = synth.random_prompt(a, 36, 300, seed=77)
    mi = torch.tensor([[[100, 140], [220, 230]]], dtype=torch.int64)
    trace = []
    want_res = VoiceCraftOracle(a, sd).inference(x, xl, y, mi, top_k=1, trace=trace)
    want = torch.stack([t["logits"][0] for t in trace]).numpy()
    forced = torch.stack([t["tokens"] for t in trace]).numpy()
    eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype="bf16", max_seqs=1, max_positions=1024)
    res, lg = eng.inference(x.cuda(), xl.cuda(), y.cuda(), mi, top_k=1, _forced=forced, _logit_steps=len(trace))
    assert np.array_equal(res.cpu().numpy(), want_res.numpy())
    rel = rel_l2(lg.cpu().numpy(), want)
    assert rel.max() <= 2e-2, rel.max()


def test_bf16_long_context_at_the_benchmark_shape_tiny128():
    """The benchmarked sequence geometry (Lx 80, 150 prompt frames -> 650 generated, 654 graph-replayed
    steps, final context 884 positions) on the 2-layer hd-128 model: bf16 logits of EVERY step, teacher-forced
    on the oracle's own trajectory, within 2e-2 relative L2 - named check points 0 / 100 / 300 / 500 / 653."""
    from oracle.voicecraft_oracle import VoiceCraftOracle
    from voicecraft_amd import synth
    from voicecraft_amd.engine import VoiceCraftEngine
    a = synth.make_args("tiny128")
    sd = synth.make_state_dict(a, seed=12)
    x, xl, y = synth.random_prompt(a, 80, 150, seed=2)
    trace = []
    res_o, gen_o = VoiceCraftOracle(a, sd).inference_tts(x, xl, y, top_k=1, stop_repetition=3, trace=trace)
    assert gen_o.shape[2] == 650 and len(trace) == 654
    want = torch.stack([t["logits"][0] for t in trace]).numpy()
    forced = torch.stack([t["tokens"] for t in trace]).numpy()
    eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype="bf16", max_seqs=1, max_positions=1024, use_graph=True)
    res, gen, lg = eng.inference_tts(x.cuda(), xl.cuda(), y.cuda(), top_k=1, stop_repetition=3, _forced=forced, _logit_steps=654)
    assert np.array_equal(res.cpu().numpy(), res_o.numpy())
    rel = rel_l2(lg.cpu().numpy(), want)
    assert rel.max() <= 2e-2, {s_: float(rel[s_]) for s_ in (0, 100, 300, 500, 653)}
    # free-running, sampled, as bench.py runs it: exactly 650 frames of plain codes
    gen = eng.inference_tts(x.cuda(), xl.cuda(), y.cuda(), top_k=40, stop_repetition=3, _seed=5)[1].cpu().numpy()
    assert_free_running_frames(gen, a, 650)


def test_full_size_giga830M_long_context_bench_shape():
    """BASELINE config 3 itself: giga830M, bf16, Lx 80, 150 -> 650 frames, hipGraph replay, 654 steps to a
    context of 884 positions (16 heads x 128, 8 attention splits over up to 111 positions each).  A random
    token trajectory is forced; the oracle evaluates it in ONE causal pass (tts_logits_for_trajectory) and the
    engine's logits at steps 0 / 100 / 300 / 500 / 653 must agree within 2e-2 relative L2.  Then the
    free-running top-k 40 run of bench.py: exactly 650 frames, every id a plain code."""
    from oracle.voicecraft_oracle import VoiceCraftOracle
    from voicecraft_amd import synth
    from voicecraft_amd.engine import VoiceCraftEngine
    a = synth.make_args("giga830M")
    sd = synth.make_state_dict(a, seed=0, fast=True)
    x, xl, y = synth.random_prompt(a, 80, 150, seed=1)
    K, n = 4, 654
    rs = np.random.RandomState(3)
    toks = rs.randint(0, 2048, size=(n, K)).astype(np.int64)
    for j in range(K):                                   # the staggered end of the span (steps 650..653)
        toks[650 + j, :j] = a.empty_token
        toks[650 + j, j] = a.eos
    steps = [0, 100, 300, 500, 653]
    torch.set_num_threads(min(16, torch.get_num_threads() or 1) or 1)
    want = VoiceCraftOracle(a, sd).tts_logits_for_trajectory(x, y, toks, steps=steps).numpy()
    eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype="bf16", max_seqs=1, max_positions=1024, use_graph=True)
    c0 = eng.launch_counts()
    res, gen, lg = eng.inference_tts(x.cuda(), xl.cuda(), y.cuda(), top_k=40, stop_repetition=3, _forced=toks, _logit_steps=n)
    c = {k: eng.launch_counts()[k] - c0[k] for k in c0}
    assert gen.shape[2] == 650 and eng.last_steps >= 654
    # round 5: at this width the one-row step's FFN down-projection finishes its row (row_gemm_fr1_k, once per layer and captured
    # step) unless the form is preset off (VC_FR_ONE=0)
    assert c["row_gemm_fr1"] >= a.num_decoder_layers or "|r1=0," in eng.options(), c
    got = lg.cpu().numpy()[steps]
    rel = rel_l2(got, want)
    assert rel.max() <= 2e-2, dict(zip(steps, rel.tolist()))
    gen = eng.inference_tts(x.cuda(), xl.cuda(), y.cuda(), top_k=40, stop_repetition=3, _seed=11)[1].cpu().numpy()
    assert_free_running_frames(gen, a, 650)


@pytest.mark.parametrize("B", [8, 12, 24])
def test_bf16_batched_decode_logits_per_sequence(B):
    """bf16 batched decode (per-row LayerNorm launch, 16-slot plain prologue, 4- and 2-split attention merges):
    every sequence of a ragged batch is teacher-forced on ITS OWN oracle trajectory and its per-step head
    logits must stay within 2e-2 relative L2 of the fp32 oracle."""
    from oracle.voicecraft_oracle import VoiceCraftOracle
    from voicecraft_amd import synth
    from voicecraft_amd.engine import VoiceCraftEngine
    a = synth.make_args("tiny_h16")
    sd = synth.make_state_dict(a, seed=4)
    prompts = [synth.random_prompt(a, 4 + (u % 5), 9 + 3 * (u % 7), seed=300 + u) for u in range(B)]
    orc = VoiceCraftOracle(a, sd)
    traces, want_res = [], []
    for (xx, xl, yy) in prompts:
        tr = []
        want_res.append(orc.inference_tts(xx, xl, yy, top_k=1, stop_repetition=3, trace=tr)[0].numpy())
        traces.append(tr)
    torch.set_num_threads(min(16, torch.get_num_threads() or 1) or 1)
    n = max(len(t) for t in traces)
    forced = np.zeros((n, B, 4), dtype=np.int64)
    for b, tr in enumerate(traces):
        forced[: len(tr), b] = torch.stack([t["tokens"] for t in tr]).numpy()
    eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype="bf16", max_seqs=B, max_positions=256)
    outs, lg = eng.inference_tts_multi([p[0][0] for p in prompts], [p[2][0] for p in prompts], top_k=1, stop_repetition=3,
                                       _forced=forced, _logit_steps=n)
    lg = lg.cpu().numpy()
    worst = 0.0
    for b, tr in enumerate(traces):
        assert np.array_equal(outs[b][0].cpu().numpy(), want_res[b])
        want = torch.stack([t["logits"][0] for t in tr]).numpy()
        worst = max(worst, float(rel_l2(lg[: len(tr), b], want).max()))
    assert worst <= 2e-2, worst


@pytest.mark.parametrize("dtype", ["fp32"])
def test_long_tts_sentences_with_shared_prefix_equal_independent_calls(dtype):
    """SURVEY §8f-2, gradio_app.py:231-236, :249-313: every sentence of a Long-TTS request is synthesised on the text
    [transcript of the voice prompt ; sentence] with the same audio prompt.  inference_tts_long decodes them together and
    computes the K/V of the shared transcript prefix once (vc_tts_multi shared_text_prefix): every sentence must be
    token-identical (fp32, greedy) to an INDEPENDENT oracle call on its own full text, with and without the reuse."""
    from oracle.voicecraft_oracle import VoiceCraftOracle
    from voicecraft_amd import synth
    from voicecraft_amd.engine import VoiceCraftEngine
    a = synth.make_args("tiny")
    sd = synth.make_state_dict(a, seed=21)
    rs = np.random.RandomState(8)
    x_prompt = torch.from_numpy(rs.randint(0, 100, size=(9,)).astype(np.int64))
    sents = [torch.from_numpy(rs.randint(0, 100, size=(n,)).astype(np.int64)) for n in (5, 11, 3, 8, 6)]
    _, _, y = synth.random_prompt(a, 1, 37, seed=9)
    eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype=dtype, max_seqs=4, max_positions=512)   # 5 sentences -> chunks of 4 + 1
    orc = VoiceCraftOracle(a, sd)
    want = []
    for sv in sents:
        x = torch.cat([x_prompt, sv]).unsqueeze(0)
        want.append(orc.inference_tts(x, torch.tensor([x.shape[1]]), y, top_k=1, stop_repetition=3)[0].numpy())
    for reuse in (True, False):
        outs = eng.inference_tts_long(x_prompt, sents, y, top_k=1, stop_repetition=3, reuse_prefix=reuse)
        assert len(outs) == len(sents)
        for (res, gen), w in zip(outs, want):
            assert res.shape == w.shape and np.array_equal(res.cpu().numpy(), w), f"reuse={reuse}"
    # a later plain call must not see a stale shared prefix
    x = torch.cat([x_prompt, sents[1]]).unsqueeze(0)
    res = eng.inference_tts(x.cuda(), torch.tensor([x.shape[1]]).cuda(), y.cuda(), top_k=1, stop_repetition=3)[0]
    assert np.array_equal(res.cpu().numpy(), want[1])


def test_bf16_residual_stream_with_a_large_common_offset():
    """Trained pre-LN stacks carry channels-wide offsets and outlier channels in the residual stream.  The decode GEMMs
    fold the LayerNorm into the weights and multiply the row itself, so the row is CENTRED before it is rounded to
    bf16 (vc_gemm.hip, LN prologue): with |mean| / sigma ~ 10 and a few 10x outlier channels the teacher-forced bf16
    logits must still sit within the 2e-2 bar (un-centred rounding amplifies the error by |mean| / sigma)."""
    from oracle.voicecraft_oracle import VoiceCraftOracle
    from voicecraft_amd import synth
    from voicecraft_amd.engine import VoiceCraftEngine
    a = synth.make_args("tiny128")
    sd = synth.make_state_dict(a, seed=14)
    rs = np.random.RandomState(5)
    outl = torch.from_numpy(rs.choice(a.d_model, size=6, replace=False))
    sd["text_embedding.word_embeddings.weight"] += 20.0                 # sigma of a text row ~ 1, of an audio row (4 tables) ~ 2
    sd["text_embedding.word_embeddings.weight"][:, outl] *= 10.0
    for k in range(a.n_codebooks):
        sd[f"audio_embedding.{k}.word_embeddings.weight"] += 5.0
        sd[f"audio_embedding.{k}.word_embeddings.weight"][:, outl] *= 10.0
    x, xl, y = synth.random_prompt(a, 9, 30, seed=8)
    trace = []
    res_o, _ = VoiceCraftOracle(a, sd).inference_tts(x, xl, y, top_k=1, stop_repetition=3, trace=trace)
    want = torch.stack([t["logits"][0] for t in trace]).numpy()
    forced = torch.stack([t["tokens"] for t in trace]).numpy()
    eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype="bf16", max_seqs=1, max_positions=256)
    res, gen, lg = eng.inference_tts(x.cuda(), xl.cuda(), y.cuda(), top_k=1, stop_repetition=3, _forced=forced, _logit_steps=len(trace))
    rel = rel_l2(lg.cpu().numpy(), want)
    assert rel.max() <= 2e-2, float(rel.max())


def test_shared_prefix_is_verified_and_does_not_leak_into_later_calls():
    """vc_tts_multi(shared_text_prefix): texts that differ inside the claimed prefix are refused (device check while the
    prompts are built), and the device word that redirects positions below the prefix to sequence 0's cache is cleared
    on that error path too: a teacher-forced evaluation pass (vc_eval_forward, B > 1) right after the failed call gives
    the same loss as on a fresh engine."""
    from _util import build_forward_case
    from voicecraft_amd import synth
    from voicecraft_amd.engine import VoiceCraftEngine
    spec, args, sd, batch = build_forward_case("fwd_b3_ragged")
    cu = {k: v.cuda() for k, v in batch.items()}
    fresh = VoiceCraftEngine(args, sd, device="cuda:0", dtype="fp32", max_seqs=4, max_positions=512)
    want = float(fresh.forward(cu, spec["spans"])["loss"])
    del fresh
    eng = VoiceCraftEngine(args, sd, device="cuda:0", dtype="fp32", max_seqs=4, max_positions=512)
    xs = [torch.tensor([1, 2, 3, 4, 5, 6]), torch.tensor([1, 2, 9, 4, 7, 8, 9])]          # differ at index 2 < 4
    _, _, y = synth.random_prompt(args, 1, 12, seed=2)
    with pytest.raises(AssertionError, match="shared_text_prefix"):
        eng.inference_tts_multi(xs, [y[0], y[0]], top_k=1, _shared_text_prefix=4)
    assert float(eng.forward(cu, spec["spans"])["loss"]) == want
    # and the same texts with an honest prefix run
    outs = eng.inference_tts_multi(xs, [y[0], y[0]], top_k=1, _shared_text_prefix=2)
    assert len(outs) == 2
