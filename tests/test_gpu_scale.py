"""-m gpu: parity AT THE BENCHMARKED SHAPES of giga830M, for the kernel forms those shapes really launch.

The tiny presets never reach the 128-channel side-by-side block GEMM (it needs >= 240 workgroups: a giga-size matrix
and a pass of >= 385 rows), the MFMA prefill attention at head_dim 128 over several passes, or the weight-stationary
wide-decode kernel at d = 2048.  Here BASELINE configs 4 and 5 (C4 = Lx 80, 800-frame utterance, span
[300,400) -> a 792-row prefill; C5 = 8 utterances per GPU, Lx 80, 150 prompt frames -> one 1 848-row prefill stream and
8-row decode steps) and a 32-sequence decode are teacher-forced on random trajectories and compared with the oracle's
ONE-PASS evaluation of the same trajectory (fp32, CPU).  Every test also reads the engine's launch census
(`launch_counts`) and asserts that the intended kernel form ran.

Tolerances: bf16 per-step relative L2 <= 2e-2; fp32 |delta| <= 1e-3 and the same arg-max per codebook.
"""
import numpy as np
import pytest
import torch

from test_gpu_model import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def giga():
    from oracle.voicecraft_oracle import VoiceCraftOracle
    from voicecraft_amd import synth
    a = synth.make_args("giga830M")
    sd = synth.make_state_dict(a, seed=0, fast=True)
    torch.set_num_threads(min(16, torch.get_num_threads() or 1) or 1)
    return a, sd, VoiceCraftOracle(a, sd)


def forced_trajectory(a, n, seed, term):
    """n steps of random plain codes whose last K steps are the staggered end of a span (voicecraft.py:1057-1066)."""
    K = a.n_codebooks
    toks = np.random.RandomState(seed).randint(0, 2048, size=(n, K)).astype(np.int64)
    for j in range(K):
        toks[n - K + j, :j] = a.empty_token
        toks[n - K + j, j] = term
    return toks


def delta(after, before):
    return {k: after[k] - before[k] for k in after}


def test_c4_editing_shape_792_row_prefill(giga):
    """BASELINE config 4: the 792-row editing prefill runs as ONE pass of 800 rows (128-channel side-by-side block GEMM
    on QKV / FFN-up / FFN-down, the 64-channel form on the out-projection, MFMA tile attention at head_dim 128); bf16
    logits of the span's first / middle / last decode step against the oracle, then the first step in fp32."""
    from voicecraft_amd import synth
    from voicecraft_amd.engine import VoiceCraftEngine
    a, sd, orc = giga
    x, xl, y = synth.random_prompt(a, 80, 800, seed=1)
    mi = torch.tensor([[[300, 400]]], dtype=torch.int64)
    n = 80                                                # the span ends on the forced terminator, before the length cap
    toks = forced_trajectory(a, n, seed=4, term=a.eog)
    steps = [0, 40, n - 1]
    want = orc.edit_logits_for_trajectory(x, y, mi, toks, steps=steps).numpy()
    eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype="bf16", max_seqs=1, max_positions=1024)
    c0 = eng.launch_counts()
    res, lg = eng.inference(x.cuda(), xl.cuda(), y.cuda(), mi, top_k=40, _forced=toks, _logit_steps=n)
    c = delta(eng.launch_counts(), c0)
    L = a.num_decoder_layers
    assert c["blk128_sbs"] == 3 * L and c["blk64"] == L and c["big256"] == 0, c     # QKV, FFN-up, FFN-down / the out-projection
    assert c["tile_attn"] == L and c["ln_rows"] == 2 * L, c
    assert res.shape == (1, a.n_codebooks, 800 - 100 + (n - a.n_codebooks))          # voicecraft.py:900
    rel = rel_l2(lg.cpu().numpy()[steps], want)
    assert rel.max() <= 2e-2, dict(zip(steps, rel.tolist()))
    del eng
    e32 = VoiceCraftEngine(a, sd, device="cuda:0", dtype="fp32", max_seqs=1, max_positions=1024)
    _, lg32 = e32.inference(x.cuda(), xl.cuda(), y.cuda(), mi, top_k=40, _forced=toks, _logit_steps=1)
    got = lg32.cpu().numpy()[0]
    live = np.abs(want[0]) < 1e3
    assert np.abs((got - want[0]) * live).max() <= 1e-3, float(np.abs((got - want[0]) * live).max())
    assert np.array_equal(np.where(live, got, -1e9).argmax(-1), np.where(live, want[0], -1e9).argmax(-1))


def test_c5_share_eight_utterances_one_prefill_stream(giga):
    """BASELINE config 5's per-GPU share: 8 x (Lx 80, 150 frames) prefilled as ONE row stream (8 x 240 = 1 920 rows in a
    single pass: QKV / FFN-up / FFN-down on the 256 x 256 LDS-DMA GEMM) and decoded 8 rows per step for 654 steps (finished-row
    form: out-projection / FFN-down on 8-channel tiles over K = 2048 / 8192, 128 KB of X in LDS); per-sequence bf16 logits at step 0 for every sequence and at
    steps 300 / 653 for sequences 0, 3 and 7 (each an 884-position one-pass oracle evaluation)."""
    from voicecraft_amd import synth
    from voicecraft_amd.engine import VoiceCraftEngine
    a, sd, orc = giga
    B, n, K = 8, 654, a.n_codebooks
    prompts = [synth.random_prompt(a, 80, 150, seed=1 + u) for u in range(B)]
    forced = np.stack([forced_trajectory(a, n, seed=50 + u, term=a.eos) for u in range(B)], axis=1)      # [n,B,K]
    eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype="bf16", max_seqs=B, max_positions=1024)
    c0 = eng.launch_counts()
    outs, lg = eng.inference_tts_multi([p[0][0] for p in prompts], [p[2][0] for p in prompts], top_k=40, stop_repetition=3,
                                       _forced=forced, _logit_steps=n)
    c = delta(eng.launch_counts(), c0)
    L = a.num_decoder_layers
    assert c["big256"] == 3 * L and c["blk128_sbs"] == L and c["tile_attn"] == L, c   # one 1 920-row pass; the out-projection stays on 128 x 128
    assert c["rows_gemm"] > 0 and c["mt2"] + c["mt4"] == 0, c                 # 8-row decode: the rows-GEMM, not the wide form
    if "|fr=0," not in eng.options():                                           # (a VC_FINISHED_ROWS=0 preset runs the slab form: still correct, other census)
        assert c["rows_gemm_fr"] + c["rows_gemm_frp"] > 0 and c["ln_rows"] == 2 * L, c             # ... in the finished-row form: the only LayerNorm launches are the prefill pass's
    lg = lg.cpu().numpy()
    worst = {}
    for u in range(B):
        assert outs[u][1].shape == (1, K, 650)
        steps = [0, 300, 653] if u in (0, 3, 7) else [0]
        want = orc.tts_logits_for_trajectory(prompts[u][0], prompts[u][2], forced[:, u], steps=steps).numpy()
        rel = rel_l2(lg[steps, u], want)
        worst[u] = float(rel.max())
    assert max(worst.values()) <= 2e-2, worst


def test_thirty_two_row_decode_on_the_weight_stationary_kernel(giga):
    """32 sequences per step at d = 2048 (the 86 k codec-tokens/s line): rows_gemm_mt_k with 16 k-tiles per wave,
    two 16-row head passes, unsplit attention - per-sequence bf16 logits at three steps."""
    from voicecraft_amd import synth
    from voicecraft_amd.engine import VoiceCraftEngine
    a, sd, orc = giga
    B, n = 32, 12
    prompts = [synth.random_prompt(a, 6 + (u % 5), 8 + (u % 7), seed=400 + u) for u in range(B)]
    forced = np.stack([forced_trajectory(a, n, seed=90 + u, term=a.eos) for u in range(B)], axis=1)
    eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype="bf16", max_seqs=B, max_positions=256)
    c0 = eng.launch_counts()
    outs, lg = eng.inference_tts_multi([p[0][0] for p in prompts], [p[2][0] for p in prompts], top_k=40, stop_repetition=3,
                                       _forced=forced, _logit_steps=n)
    c = delta(eng.launch_counts(), c0)
    assert c["mt2"] + c["mt4"] + c["wd"] > 0, c
    lg = lg.cpu().numpy()
    steps = [0, 5, n - 1]
    worst = 0.0
    for u in range(B):
        want = orc.tts_logits_for_trajectory(prompts[u][0], prompts[u][2], forced[:, u], steps=steps).numpy()
        worst = max(worst, float(rel_l2(lg[steps, u], want).max()))
        assert outs[u][1].shape[2] == n - a.n_codebooks
    assert worst <= 2e-2, worst


# ------------------------------------------------------------------------------------------ giga330M (BASELINE configs 1 and 2)
# d = 1024, 16 heads of 64, 24 layers (the shape SURVEY.md §8 assumes for the 330M checkpoint).  make_plan picks other
# launch shapes from d than at d = 2048 (FFN-down split-K 4 over 64 tiles, out-projection unsplit, k-tiles per wave 8 / 16),
# head_dim 64 spreads a cached row over 8 lanes, and the 8-sequence decode takes the finished-row form at K = 4096.
@pytest.fixture(scope="module")
def giga330():
    from oracle.voicecraft_oracle import VoiceCraftOracle
    from voicecraft_amd import synth
    a = synth.make_args("giga330M")
    sd = synth.make_state_dict(a, seed=0, fast=True)
    torch.set_num_threads(min(16, torch.get_num_threads() or 1) or 1)
    return a, sd, VoiceCraftOracle(a, sd)


def test_c2_giga330M_16s_decode_bf16_and_fp32_first_step(giga330):
    """BASELINE config 2 (giga330M, batch 1, Lx 80, 150 -> 650 frames, the bench's `--preset giga330M` workload): bf16
    teacher-forced head logits at the first / middle / last decode step (231 / 531 / 884 cached positions) against the oracle's
    one-pass evaluation, on the captured graph; then the first step in fp32 (<= 1e-3, same arg-max)."""
    from voicecraft_amd import synth
    from voicecraft_amd.engine import VoiceCraftEngine
    a, sd, orc = giga330
    x, xl, y = synth.random_prompt(a, 80, 150, seed=1)
    n = 654
    toks = forced_trajectory(a, n, seed=12, term=a.eos)
    steps = [0, 300, n - 1]
    want = orc.tts_logits_for_trajectory(x, y, toks, steps=steps).numpy()
    eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype="bf16", max_seqs=1, max_positions=1024)
    c0 = eng.launch_counts()
    res, gen, lg = eng.inference_tts(x.cuda(), xl.cuda(), y.cuda(), top_k=40, stop_repetition=3, _forced=toks, _logit_steps=n)
    c = delta(eng.launch_counts(), c0)
    L = a.num_decoder_layers
    assert gen.shape == (1, a.n_codebooks, 650)
    assert c["rows_gemm"] >= 4 * L + 2 and c["rows_attn"] >= L and c["mt2"] + c["mt4"] + c["rows_gemm_fr"] + c["rows_gemm_frp"] == 0, c   # (a captured graph launches nothing the census sees)
    assert c["blk64"] + c["blk128_sbs"] == 4 * L and c["tile_attn"] == L, c           # the 240-row prefill pass
    rel = rel_l2(lg.cpu().numpy()[steps], want)
    assert rel.max() <= 2e-2, dict(zip(steps, rel.tolist()))
    del eng
    e32 = VoiceCraftEngine(a, sd, device="cuda:0", dtype="fp32", max_seqs=1, max_positions=1024)
    _, _, lg32 = e32.inference_tts(x.cuda(), xl.cuda(), y.cuda(), top_k=40, stop_repetition=3, _forced=toks[:8], _logit_steps=1)
    got = lg32.cpu().numpy()[0]
    live = np.abs(want[0]) < 1e3
    assert np.abs((got - want[0]) * live).max() <= 1e-3, float(np.abs((got - want[0]) * live).max())
    assert np.array_equal(np.where(live, got, -1e9).argmax(-1), np.where(live, want[0], -1e9).argmax(-1))


def test_c1_giga330M_greedy_free_running_equals_the_oracle_fp32(giga330):
    """BASELINE config 1 (the reference's CPU-runnable case: giga330M, greedy, 3 s prompt -> 5 s, Lx 40, 150 -> 250 frames):
    the engine in exact mode, FREE-running, must produce the oracle's token ids for all 254 steps - graph and eager."""
    from voicecraft_amd import synth
    from voicecraft_amd.engine import VoiceCraftEngine
    a, sd, orc = giga330
    x, xl, y = synth.random_prompt(a, 40, 150, seed=1)
    want_res, want_gen = orc.inference_tts(x, xl, y, top_k=1, top_p=1.0, temperature=1.0, stop_repetition=3, kvcache=1)
    assert want_gen.shape == (1, a.n_codebooks, 250)
    eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype="fp32", max_seqs=1, max_positions=512)
    for graph in (True, False):
        eng.use_graph = graph
        res, gen = eng.inference_tts(x.cuda(), xl.cuda(), y.cuda(), top_k=1, top_p=1.0, temperature=1.0, stop_repetition=3, kvcache=1)
        assert eng.last_steps == 254
        assert np.array_equal(res.cpu().numpy(), want_res.numpy()), graph


def test_giga330M_eight_utterances_and_a_576_row_editing_prefill(giga330):
    """d = 1024 on the several-row paths: (a) 8 utterances decoded together (one 1 920-row prefill stream, then 8-row
    decode steps in the finished-row form: out-projection / FFN-down on 8-channel tiles over K = 1024 / 4096) with
    per-sequence bf16 logits; (b) an editing call whose rearranged prompt is one 576-row pass."""
    from voicecraft_amd import synth
    from voicecraft_amd.engine import VoiceCraftEngine
    a, sd, orc = giga330
    L, K = a.num_decoder_layers, a.n_codebooks
    B, n = 8, 60
    prompts = [synth.random_prompt(a, 80, 150, seed=1 + u) for u in range(B)]
    forced = np.stack([forced_trajectory(a, n, seed=70 + u, term=a.eos) for u in range(B)], axis=1)
    eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype="bf16", max_seqs=B, max_positions=1024, use_graph=False)
    c0 = eng.launch_counts()
    outs, lg = eng.inference_tts_multi([p[0][0] for p in prompts], [p[2][0] for p in prompts], top_k=40, stop_repetition=3,
                                       _forced=forced, _logit_steps=n)
    c = delta(eng.launch_counts(), c0)
    assert ("|fr=0," in eng.options() or c["rows_gemm_fr"] + c["rows_gemm_frp"] >= 2 * L * (n - 1)) and c["tile_attn"] == L, c    # eager: every decode launch is counted
    lg = lg.cpu().numpy()
    worst = {}
    for u in (0, 3, 7):
        steps = [0, 30, n - 1]
        want = orc.tts_logits_for_trajectory(prompts[u][0], prompts[u][2], forced[:, u], steps=steps).numpy()
        worst[u] = float(rel_l2(lg[steps, u], want).max())
        assert outs[u][1].shape == (1, K, n - K)
    assert max(worst.values()) <= 2e-2, worst
    del eng
    # (b) editing: Lx 60, 600-frame utterance, span [200,300): 60 + 500 + 2 (K + 1) + 2 + 1 = 573 rows -> one 576-row pass
    x, xl, y = synth.random_prompt(a, 60, 600, seed=3)
    mi = torch.tensor([[[200, 300]]], dtype=torch.int64)
    n = 40
    toks = forced_trajectory(a, n, seed=5, term=a.eog)
    steps = [0, 20, n - 1]
    want = orc.edit_logits_for_trajectory(x, y, mi, toks, steps=steps).numpy()
    eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype="bf16", max_seqs=1, max_positions=1024)
    c0 = eng.launch_counts()
    res, lg = eng.inference(x.cuda(), xl.cuda(), y.cuda(), mi, top_k=40, _forced=toks, _logit_steps=n)
    c = delta(eng.launch_counts(), c0)
    assert c["blk64"] + c["blk128_sbs"] + c["big256"] == 4 * L and c["tile_attn"] == L and c["ln_rows"] == 2 * L, c
    assert res.shape == (1, K, 600 - 100 + (n - K))
    rel = rel_l2(lg.cpu().numpy()[steps], want)
    assert rel.max() <= 2e-2, dict(zip(steps, rel.tolist()))


def test_twelve_row_decode_in_the_two_half_finished_row_form(giga):
    """12 sequences per step at d = 2048: the finished-row form beyond 8 rows - unsplit attention that normalises itself, the
    out-projection on the plain prologue, the FFN down-projection taking K = 8192 through LDS in two halves with 2 x 16 fragments
    per wave (rows_gemm_fr2_k<bf16, 16>), consumers folding two rows per wave - per-sequence bf16 logits at three steps."""
    from voicecraft_amd import synth
    from voicecraft_amd.engine import VoiceCraftEngine
    a, sd, orc = giga
    B, n = 12, 10
    prompts = [synth.random_prompt(a, 6 + (u % 5), 8 + (u % 7), seed=500 + u) for u in range(B)]
    forced = np.stack([forced_trajectory(a, n, seed=190 + u, term=a.eos) for u in range(B)], axis=1)
    eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype="bf16", max_seqs=B, max_positions=256, use_graph=False)
    c0 = eng.launch_counts()
    outs, lg = eng.inference_tts_multi([p[0][0] for p in prompts], [p[2][0] for p in prompts], top_k=40, stop_repetition=3,
                                       _forced=forced, _logit_steps=n)
    c = delta(eng.launch_counts(), c0)
    if "|fr=0," not in eng.options():
        assert c["rows_gemm_fr"] + c["rows_gemm_frp"] >= 2 * a.num_decoder_layers * (n - 1) and c["mt2"] + c["mt4"] == 0, c
    lg = lg.cpu().numpy()
    steps = [0, 4, n - 1]
    worst = 0.0
    for u in range(B):
        want = orc.tts_logits_for_trajectory(prompts[u][0], prompts[u][2], forced[:, u], steps=steps).numpy()
        worst = max(worst, float(rel_l2(lg[steps, u], want).max()))
        assert outs[u][1].shape[2] == n - a.n_codebooks
    assert worst <= 2e-2, worst


def test_sixty_four_row_decode_at_giga830M(giga):
    """64 sequences per step at d = 2048 (the widest step the engine takes; README's 64-utterance line): every layer GEMM and both
    head matrices on the wide-decode kernel, the decode attention on a 64-row grid, one LayerNorm launch per consumer - per-sequence
    bf16 logits at three steps against the oracle's one-pass evaluation of each sequence's own forced trajectory, and the launch
    census of the eager loop (every launch counted)."""
    from voicecraft_amd import synth
    from voicecraft_amd.engine import VoiceCraftEngine
    a, sd, orc = giga
    B, n, L = 64, 10, a.num_decoder_layers
    prompts = [synth.random_prompt(a, 5 + (u % 6), 7 + (u % 9), seed=900 + u) for u in range(B)]
    forced = np.stack([forced_trajectory(a, n, seed=290 + u, term=a.eos) for u in range(B)], axis=1)
    eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype="bf16", max_seqs=B, max_positions=128, use_graph=False)
    worst_by_form = {}
    for form in (1, 0):                  # X through the wave-private LDS stage (the default) / fragments straight from L2 (vc_gemm_wd.hip)
        eng.set_option("wd_stage", form)
        c0 = eng.launch_counts()
        outs, lg = eng.inference_tts_multi([p[0][0] for p in prompts], [p[2][0] for p in prompts], top_k=40, stop_repetition=3,
                                           _forced=forced, _logit_steps=n)
        c = delta(eng.launch_counts(), c0)
        worst_by_form[form] = (c, outs, lg.cpu().numpy())
    c, outs, lg = worst_by_form[1]
    assert np.array_equal(worst_by_form[0][2], lg)          # the same products summed in the same order: only the path X takes differs
    wide = c["mt2"] + c["mt4"] + c["wd"]
    assert wide >= (4 * L + 2) * (n - 1), c                 # QKV, out-projection, FFN-up, FFN-down of every layer + the two head matrices, per step
    assert c["rows_attn"] >= L * (n - 1) and c["ln_rows"] >= (2 * L + 1) * (n - 1), c
    assert c["rows_gemm_fr"] + c["rows_gemm_frp"] + c["row_gemm_fr1"] == 0, c
    steps = [0, 4, n - 1]
    worst = {}
    for u in range(B):
        want = orc.tts_logits_for_trajectory(prompts[u][0], prompts[u][2], forced[:, u], steps=steps).numpy()
        worst[u] = float(rel_l2(lg[steps, u], want).max())
        assert outs[u][1].shape[2] == n - a.n_codebooks
    assert max(worst.values()) <= 2e-2, {u: w for u, w in worst.items() if w > 1e-2}


@pytest.mark.parametrize("batch_size", [3, 4])
def test_best_of_n_at_giga830M_replays_the_oracles_draws(giga, batch_size):
    """`inference_tts_batch(batch_size = 3 | 4)` at giga830M (the mode the reference's front-ends run: gradio_app.py:506
    sample_batch_size = 3; models/voicecraft.py:1156-1171, :1296-1302): the same raw draws go through the oracle's and the
    engine's state machine - two samples emit the terminator at the same step, the LAST of them is kept, the others are dropped,
    the kept one finishes its K - 1 tail - and the kept tokens must be identical; the bf16 head logits of EVERY sample before
    that step, and of the kept one after it, stay within 2e-2 of the oracle's."""
    from voicecraft_amd import synth
    from voicecraft_amd.engine import VoiceCraftEngine
    a, sd, orc = giga
    K, V = a.n_codebooks, a.audio_vocab_size + a.n_special
    x, xl, y = synth.random_prompt(a, 24, 50, seed=21)
    n, t_end = 26, 15
    draws = np.random.RandomState(60 + batch_size).randint(0, 2048, size=(n, batch_size, K)).astype(np.int64)
    draws[t_end, 0, 0] = draws[t_end, batch_size - 2, 0] = int(a.eos)          # two samples terminate together: the later index is kept
    kn = dict(top_k=40, top_p=1.0, temperature=1.0, stop_repetition=3, kvcache=1, batch_size=batch_size)
    trace = []
    want_res, want_gen = orc.inference_tts_batch(x, xl, y, forced_draws=draws, trace=trace, **kn)
    assert want_gen.shape == (1, K, t_end)
    eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype="bf16", max_seqs=batch_size, max_positions=512)
    res, gen, lg = eng.inference_tts_batch(x.cuda(), xl.cuda(), y.cuda(), **kn, _forced=draws, _forced_mode="draws", _seed=1,
                                           _logit_steps=len(trace))
    assert np.array_equal(res.cpu().numpy(), want_res.numpy())
    assert np.array_equal(gen.cpu().numpy()[0, 0], draws[:t_end, batch_size - 2, 0])          # the kept sample's trajectory
    assert eng.last_steps == t_end + K
    lg = lg.cpu().numpy()
    keep = batch_size - 2
    worst = 0.0
    for s in (0, 7, t_end, len(trace) - 1):
        want = trace[s]["logits"].numpy()                                        # [B,K,V]
        rows = range(batch_size) if s <= t_end else [keep]
        for b in rows:
            worst = max(worst, float(rel_l2(lg[s, b][None], want[b][None]).max()))
    assert worst <= 2e-2, worst
