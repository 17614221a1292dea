"""-m gpu: the round-5 forms of the ONE-row decode step against the oracle, each with the launch census telling which form ran.

* `fr_one`: the FFN down-projection finishes its row (`row_gemm_fr1_k`: 8-channel tiles over the whole K, two k-tiles per MFMA
  fragment, residual + bias added in the epilogue) instead of leaving four split-K slabs; the next layer's QKV projection and
  heads-1 then fold the LayerNorm of ONE finished row (`rows_gemm_k<..., NP = 0>`).
* the LayerNorm prologue requests only the slabs the pass has (0 / 2 / 4; an option - `ln_trim` - through round 5, always on since).
* `attn_fast`: the decode attention takes a wave's maximum before any exponential (and exp2 in bf16 mode).
* `qkv_p8`: behind a finished row the QKV projection runs on 8-channel tiles with two k-tiles per MFMA fragment too
  (`row_gemm_fr1_k<PRO_LN, EPI_QKV>`).

Every state of the options gives the oracle's greedy tokens in the exact fp32 mode (captured graph and eager) and
teacher-forced head logits within 2e-2 in bf16; the defaults are what every other test of the suite runs on."""
import itertools

import numpy as np
import pytest
import torch

from test_gpu_model import rel_l2

pytestmark = pytest.mark.gpu


def _delta(after, before):
    return {k: after[k] - before[k] for k in after}


def _oracle_run(a, sd, x, xl, y):
    from oracle.voicecraft_oracle import VoiceCraftOracle
    orc = VoiceCraftOracle(a, sd)
    tr = []
    res = orc.inference_tts(x, xl, y, top_k=1, stop_repetition=3, trace=tr)[0].numpy()
    return res, tr


@pytest.mark.parametrize("preset", ["tiny", "tiny128", "tiny_h16"])
@pytest.mark.parametrize("use_graph", [True, False])
def test_one_row_forms_fp32_tokens_equal_the_oracle(preset, use_graph):
    from voicecraft_amd import synth
    from voicecraft_amd.engine import VoiceCraftEngine
    a = synth.make_args(preset)
    sd = synth.make_state_dict(a, seed=5)
    x, xl, y = synth.random_prompt(a, 7, 19, seed=21)
    want, tr = _oracle_run(a, sd, x, xl, y)
    eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype="fp32", max_seqs=1, max_positions=256, use_graph=use_graph)
    L, n = a.num_decoder_layers, len(tr)
    for fr_one, attn_fast, qkv_p8 in list(itertools.product((2, 0), (1, 0), (1,))) + [(2, 1, 0)]:
        eng.set_option("fr_one", fr_one)
        eng.set_option("attn_fast", attn_fast)
        eng.set_option("qkv_p8", qkv_p8)
        assert f"|r1={fr_one},{attn_fast},{qkv_p8}" in eng.options()
        c0 = eng.launch_counts()
        got = eng.inference_tts(x.cuda(), xl.cuda(), y.cuda(), top_k=1, stop_repetition=3)[0].cpu().numpy()
        c = _delta(eng.launch_counts(), c0)
        assert np.array_equal(got, want), (fr_one, attn_fast, qkv_p8)
        if fr_one:       # one finished-row producer (+ one paired QKV projection) per layer and decode step (a captured graph counts its launches once, at capture)
            per = 2 if qkv_p8 else 1
            assert c["row_gemm_fr1"] >= per * (L if use_graph else L * (n - 1)), c
            assert qkv_p8 or c["row_gemm_fr1"] < 2 * (L * 8 if use_graph else L * (n - 1)), c
        else:
            assert c["row_gemm_fr1"] == 0, c


@pytest.mark.parametrize("preset", ["tiny", "tiny128", "tiny_h16"])
def test_one_row_forms_bf16_teacher_forced_logits(preset):
    from voicecraft_amd import synth
    from voicecraft_amd.engine import VoiceCraftEngine
    a = synth.make_args(preset)
    sd = synth.make_state_dict(a, seed=6)
    x, xl, y = synth.random_prompt(a, 6, 23, seed=31)
    _, tr = _oracle_run(a, sd, x, xl, y)
    want = torch.stack([t["logits"][0] for t in tr]).numpy()
    forced = torch.stack([t["tokens"] for t in tr]).numpy()
    eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype="bf16", max_seqs=1, max_positions=256)
    got = {}
    for state in ((2, 1, 1), (0, 1, 1), (2, 1, 0), (2, 0, 1), (0, 0, 0)):
        for name, v in zip(("fr_one", "attn_fast", "qkv_p8"), state):
            eng.set_option(name, v)
        c0 = eng.launch_counts()
        _, _, lg = eng.inference_tts(x.cuda(), xl.cuda(), y.cuda(), top_k=1, stop_repetition=3, _forced=forced, _logit_steps=len(tr))
        c = _delta(eng.launch_counts(), c0)
        assert (c["row_gemm_fr1"] > 0) == bool(state[0]), (state, c)
        lg = lg.cpu().numpy()
        err = rel_l2(lg, want)
        assert err.max() <= 2e-2, (state, float(err.max()))
        got[state] = lg
    # the options change the order of sums / the exponential: roundings of the same numbers
    for st in ((0, 1, 1), (2, 1, 0), (2, 0, 1), (0, 0, 0)):
        assert np.abs(got[(2, 1, 1)] - got[st])[np.abs(got[st]) < 1e3].max() < 0.25, st


def test_one_row_finished_row_producer_writes_residual_plus_bias():
    """row_gemm_fr1_k alone (through the FFN down-projection microbenchmark of a live engine, which zeroes its inputs): with
    act = 0 and h_in = 0 the finished row is the bias, bit for bit, in every channel (the epilogue's addressing)."""
    from voicecraft_amd import synth
    from voicecraft_amd.engine import VoiceCraftEngine
    a = synth.make_args("tiny128")
    sd = synth.make_state_dict(a, seed=7)
    eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype="bf16", max_seqs=1, max_positions=256)
    eng.set_option("fr_one", 2)
    # the microbenchmark zeroes act and hA, so the finished row must equal the bias exactly
    eng.bench_kernel("ffn2", n_rows=1, iters=1)
    hB = eng.debug_read("hB", (a.d_model,))
    # (the microbenchmark's launch i runs on layer i % L: its single timed launch is layer 0's)
    b2 = sd["decoder.layers.0.linear2.bias"]
    assert torch.allclose(hB, b2, atol=0, rtol=0), float((hB - b2).abs().max())
