"""-m gpu: run-time options (vc_set_option) never change results, and the finished-row form of several-row decode steps
(2..8 rows in bf16, 2..4 in the exact fp32 mode: out-projection / FFN-down own whole rows on 8-channel tiles, QKV / FFN-up /
heads fold the LayerNorm of finished rows one wave per row - no split-K slabs, no LayerNorm launch) against the oracle,
with the launch census telling which form ran.  9..16 rows: the attention is unsplit and normalises itself, the FFN down-projection
takes K through LDS in two halves (rows_gemm_fr2_k), every consumer wave folds two rows."""
import numpy as np
import pytest
import torch

from test_gpu_model import rel_l2

pytestmark = pytest.mark.gpu


def _delta(after, before):
    return {k: after[k] - before[k] for k in after}


def _oracle_traces(a, sd, prompts):
    from oracle.voicecraft_oracle import VoiceCraftOracle
    torch.set_num_threads(min(4, torch.get_num_threads()))      # tiny models: a host with 100+ cores spends its time in thread hand-offs otherwise
    orc = VoiceCraftOracle(a, sd)
    traces, want_res = [], []
    for (xx, xl, yy) in prompts:
        tr = []
        want_res.append(orc.inference_tts(xx, xl, yy, top_k=1, stop_repetition=3, trace=tr)[0].numpy())
        traces.append(tr)
    return traces, want_res


@pytest.mark.parametrize("preset,B", [("tiny_h16", 2), ("tiny_h16", 3), ("tiny_h16", 5), ("tiny_h16", 8), ("tiny128", 4), ("tiny128", 7), ("tiny", 8),
                                      ("tiny_h16", 9), ("tiny_h16", 12), ("tiny128", 16)])
def test_finished_row_form_bf16_per_sequence_logits(preset, B):
    """Every sequence of a ragged batch teacher-forced on its own oracle trajectory: per-step head logits within 2e-2 of the
    fp32 oracle in the finished-row form (census: rows_gemm_fr launches, no LayerNorm launch in the decode steps), and again
    with the form switched off at run time (census: none) - head_dim 32 / 128 / 64, attention merges of 8 / 4 / 2 partials."""
    from voicecraft_amd import synth
    from voicecraft_amd.engine import VoiceCraftEngine
    a = synth.make_args(preset)
    sd = synth.make_state_dict(a, seed=4)
    prompts = [synth.random_prompt(a, 4 + (u % 5), 9 + 3 * (u % 7), seed=300 + u) for u in range(B)]
    traces, want_res = _oracle_traces(a, sd, prompts)
    n = max(len(t) for t in traces)
    forced = np.zeros((n, B, 4), dtype=np.int64)
    for b, tr in enumerate(traces):
        forced[: len(tr), b] = torch.stack([t["tokens"] for t in tr]).numpy()
    eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype="bf16", max_seqs=B, max_positions=256, use_graph=False)
    L = a.num_decoder_layers
    got = {}
    qp_can = (a.d_model * 2) % 1024 == 0       # the centred copy (and the paired QKV consumer behind it) needs rows of whole 1 KB requests
    for mode in ("on", "p32", "nohq", "qkv12", "unpaired", "off"):   # on: the default; p32: fp32 attention partials (rounds 4-5); nohq: the consumers fold the fp32 rows (no centred copy); qkv12: the QKV projection of 2..8 rows on the 12-channel tiles; unpaired: round 4's FFN-down producer; off: split-K slabs + LayerNorm launches
        eng.set_option("finished_rows", 0 if mode == "off" else 16)
        eng.set_option("fr_pair", 0 if mode == "unpaired" else 1)
        eng.set_option("att_p16", 0 if mode == "p32" else 1)
        eng.set_option("hq", 0 if mode == "nohq" else 1)
        eng.set_option("qkv_p8", 1 if mode == "qkv12" else 2)
        assert ("|fr=0," if mode == "off" else "|fr=16,") in eng.options()
        c0 = eng.launch_counts()
        outs, lg = eng.inference_tts_multi([p[0][0] for p in prompts], [p[2][0] for p in prompts], top_k=1, stop_repetition=3,
                                           _forced=forced, _logit_steps=n)
        c = _delta(eng.launch_counts(), c0)
        # the paired QKV consumer (round 6): layers 1.. of every step of 2..8 rows, only behind the centred copy
        if mode in ("on", "p32", "unpaired") and qp_can and B <= 8:
            assert c["rows_gemm_qp"] >= (L - 1) * (n - 1), c
        elif mode in ("nohq", "qkv12", "off") or not qp_can:
            assert c["rows_gemm_qp"] == 0, (mode, c)
        if mode == "unpaired":
            assert c["rows_gemm_frp"] == 0 and c["rows_gemm_fr"] >= 2 * L * (n - 1), c
        elif mode != "off":
            assert c["rows_gemm_fr"] + c["rows_gemm_frp"] >= 2 * L * (n - 1), c           # (the first of the n samples comes from the prefill's logits)
            if B <= 8 and mode != "unpaired":      # round 5: the FFN down-projection of 2..8 rows in the paired form (two k-tiles per MFMA fragment)
                assert c["rows_gemm_frp"] >= L * (n - 1), c
        else:
            assert c["rows_gemm_fr"] + c["rows_gemm_frp"] == 0 and (B < 3 or c["ln_rows"] >= 2 * L * (n - 1)), c
        lg = lg.cpu().numpy()
        worst = 0.0
        for b, tr in enumerate(traces):
            assert np.array_equal(outs[b][0].cpu().numpy(), want_res[b])
            want = torch.stack([t["logits"][0] for t in tr]).numpy()
            worst = max(worst, float(rel_l2(lg[: len(tr), b], want).max()))
        assert worst <= 2e-2, (mode, worst)
        got[mode] = lg
    assert np.abs(got["on"] - got["off"])[np.abs(got["off"]) < 1e3].max() < 0.25      # two roundings of the same numbers
    assert np.abs(got["on"] - got["p32"])[np.abs(got["off"]) < 1e3].max() < 0.25      # bf16 / fp32 attention partials (steps of up to 8 rows: wider batches get there as they shrink)
    assert np.abs(got["on"] - got["nohq"])[np.abs(got["off"]) < 1e3].max() < 0.25     # rows centred on the previous mean / on their own: two roundings of the same numbers
    assert np.abs(got["on"] - got["unpaired"])[np.abs(got["off"]) < 1e3].max() < 0.25    # paired / unpaired producer: another order of the same sums
    assert np.abs(got["on"] - got["qkv12"])[np.abs(got["off"]) < 1e3].max() < 0.25     # the QKV projection on 8- / 12-channel tiles: another order of the same sums


@pytest.mark.parametrize("preset,B", [("tiny_h16", 2), ("tiny_h16", 4), ("tiny128", 3), ("tiny_h16", 11), ("tiny128", 16)])
def test_finished_row_form_fp32_tokens_equal_the_oracle(preset, B):
    """Exact mode: greedy FREE-running tokens of every utterance equal the oracle's, on the captured graph (2..4 rows take the
    finished-row form there: X of the FFN down-projection is 4 x 4d fp32 values)."""
    from oracle.voicecraft_oracle import VoiceCraftOracle
    from voicecraft_amd import synth
    from voicecraft_amd.engine import VoiceCraftEngine
    a = synth.make_args(preset)
    sd = synth.make_state_dict(a, seed=4)
    prompts = [synth.random_prompt(a, 4 + (u % 5), 9 + 3 * (u % 7), seed=300 + u) for u in range(B)]
    eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype="fp32", max_seqs=B, max_positions=256)
    orc = VoiceCraftOracle(a, sd)
    want = [orc.inference_tts(xx, xl, yy, top_k=1, stop_repetition=3)[0].numpy() for (xx, xl, yy) in prompts]
    for hq, p8 in ((1, 2), (1, 1), (0, 2)):        # the consumers fold the producers' centred copy of the rows (round 6: the QKV projection of 2..8 rows on 8- / 12-channel tiles) / the fp32 rows
        eng.set_option("hq", hq)
        eng.set_option("qkv_p8", p8)
        c0 = eng.launch_counts()
        outs = eng.inference_tts_multi([p[0][0] for p in prompts], [p[2][0] for p in prompts], top_k=1, stop_repetition=3)
        c = _delta(eng.launch_counts(), c0)
        assert c["rows_gemm_fr"] + c["rows_gemm_frp"] > 0, c
        if hq == 1 and p8 == 2:
            assert B > 8 or c["rows_gemm_qp"] > 0, c       # (wider batches get to 2..8 rows only as they shrink)
        else:
            assert c["rows_gemm_qp"] == 0, (hq, p8, c)
        for w, (res, gen) in zip(want, outs):
            assert np.array_equal(res.cpu().numpy(), w), (hq, p8)


def test_options_do_not_change_tokens_and_bad_options_are_refused():
    from voicecraft_amd import synth
    from voicecraft_amd.engine import VoiceCraftEngine
    a = synth.make_args("tiny128")
    sd = synth.make_state_dict(a, seed=3)
    x, xl, y = synth.random_prompt(a, 6, 21, seed=11)
    eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype="fp32", max_seqs=2, max_positions=256)
    base = eng.inference_tts(x.cuda(), xl.cuda(), y.cuda(), top_k=1, stop_repetition=3)[0].cpu().numpy()
    for name, value in [("nt", "63,1"), ("nt", "63,0"), ("nt", "28"), ("nt", "0"), ("nt", "63"), ("graph_steps", "3"), ("graph_steps", "8"),
                        ("shrink", "0"), ("wide_gemm", "0"), ("wd_stage", "0"), ("tile_attn", "1")]:
        eng.set_option(name, value)
        got = eng.inference_tts(x.cuda(), xl.cuda(), y.cuda(), top_k=1, stop_repetition=3)[0].cpu().numpy()
        assert np.array_equal(got, base), (name, value)
    with pytest.raises(AssertionError):
        eng.set_option("no_such_option", "1")
    with pytest.raises(AssertionError):
        eng.set_option("graph_steps", "many")
    with pytest.raises(AssertionError):
        eng.set_option("attn_pf", "8")          # the prefetch roles of rounds 3-5 are gone
    with pytest.raises(AssertionError):
        eng.set_option("mt_tiles", "4")         # ... and the knobs whose measured best value became a constant
    with pytest.raises(AssertionError):
        eng.set_option("attn_nt", "1")          # ... and the K/V hint is the second value of "nt"
    eng.set_option("nt", "63,1")
    assert "|nt=63,1|" in eng.options()
    eng.set_option("nt", "28")                  # one value: the K/V policy stays
    assert "|nt=28,1|" in eng.options()


@pytest.mark.parametrize("preset,B", [("tiny", 1), ("tiny_h16", 1), ("tiny128", 20)])
def test_qkv16_image_of_prefill_and_wide_decode_passes(preset, B):
    """Options `qkv16` and `wide_heads`.  `qkv16`: prefill passes and wide decode passes (17..64 rows) run the QKV projection on a 16-channel image of
    the LayerNorm-folded matrix instead of the 12-channel tiles of the one-row kernels.  The same dot products on other MFMA lanes:
    fp32 tokens equal the oracle's in both states (one sequence: the prompt pass; 20 sequences: 20-row decode steps as well; the wide-decode
    kernel itself at d = 2048: tests/test_gpu_scale.py, 32 rows), bf16
    tokens agree between the states wherever the arg-max margin is not a rounding."""
    from voicecraft_amd import synth
    from voicecraft_amd.engine import VoiceCraftEngine
    a = synth.make_args(preset)
    sd = synth.make_state_dict(a, seed=4, head_gain=4.0)
    prompts = [synth.random_prompt(a, 4 + (u % 5), 17 + 3 * (u % 7), seed=500 + u) for u in range(B)]
    _, want_res = _oracle_traces(a, sd, prompts)
    for dtype in ("fp32", "bf16"):
        # (the 16-channel image is packed for engines that can take wide steps - max_seqs > 16 - or on request: VC_QKV16=1 at creation)
        import os
        os.environ["VC_QKV16"] = "1"
        try:
            eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype=dtype, max_seqs=B, max_positions=256)
        finally:
            os.environ.pop("VC_QKV16", None)
        res = {}
        # (wide_heads: 17..64-row steps run the heads once on the wide-decode kernel, not per 16 rows; wide_gemm 0: the weight-stationary
        # kernel of rounds 2-5)
        for q16, wh, mt in ((1, 1, 1), (0, 0, 0), (1, 0, 1), (1, 1, 0)):
            eng.set_option("qkv16", q16)
            eng.set_option("wide_heads", wh)
            eng.set_option("wide_gemm", mt)
            assert f"|q16={q16},{wh},{mt}," in eng.options()
            c0 = eng.launch_counts()
            if B == 1:
                x, xl, y = prompts[0]
                got = [eng.inference_tts(x.cuda(), xl.cuda(), y.cuda(), top_k=1, stop_repetition=3)[0].cpu().numpy()]
            else:
                outs = eng.inference_tts_multi([p[0][0] for p in prompts], [p[2][0] for p in prompts], top_k=1, stop_repetition=3)
                got = [res.cpu().numpy() for (res, gen) in outs]
            c = _delta(eng.launch_counts(), c0)
            assert c["blk64"] + c["blk64_occ2"] + c["blk128_sbs"] + c["blk128_2x2"] > 0, c          # the prompt went through the block GEMM
            if dtype == "fp32":
                for g, w in zip(got, want_res):
                    assert np.array_equal(g, w), q16
            if B > 16:       # heads of a wide step: once on the weight-stationary kernel, or 16 rows at a time on the rows-GEMM
                assert (c["mt2"] + c["mt4"] + c["wd"] > 0) if wh else (c["rows_gemm"] > 0), (wh, c)
            res[(q16, wh, mt)] = got
        if dtype == "bf16" and B > 1:
            same = sum(int(np.array_equal(g1, g0)) for g1, g0 in zip(res[(1, 1, 1)], res[(0, 0, 0)]))
            assert same >= (B * 3) // 4, (same, B)
        if dtype == "bf16" and B == 1:
            # one sequence: only the PROMPT pass reads the image (block GEMM, EPI_QKV16).  A value check instead of a token count
            # (ADVICE r05): the first step's raw head logits in both states - the same dot products on other MFMA lanes, two roundings
            x, xl, y = prompts[0]
            lg = {}
            for q16 in (0, 1):
                eng.set_option("qkv16", q16)
                lg[q16] = eng.inference_tts(x.cuda(), xl.cuda(), y.cuda(), top_k=1, stop_repetition=3, _logit_steps=1)[2].cpu().numpy()[0]
            live = np.abs(lg[0]) < 1e3
            assert np.abs(lg[1] - lg[0])[live].max() < 0.25, float(np.abs(lg[1] - lg[0])[live].max())


# launch forms of the linear layers of 17..64-row steps: rows_gemm_wd_k (round 6: X through its LDS stage - the default - or straight from L2), and the weight-stationary
# rows_gemm_mt_k of rounds 2-5 (two weight tiles per workgroup)
WIDE_FORMS = [(("wide_gemm", 1), ("wd_stage", 1)), (("wide_gemm", 1), ("wd_stage", 0)), (("wide_gemm", 0),)]


def _free_running_multi(eng, prompts):
    outs = eng.inference_tts_multi([p[0][0] for p in prompts], [p[2][0] for p in prompts], top_k=1, stop_repetition=3)
    return [res.cpu().numpy() for (res, gen) in outs]


@pytest.mark.parametrize("preset,B", [("tiny_h16", 48), ("tiny128", 64), ("tiny", 33)])
def test_wide_decode_33_to_64_rows_fp32_tokens_equal_the_oracle(preset, B):
    """33..64 sequences per step (row tiles 3 and 4 of the wide-decode kernel, the 64-row attention grid, the sampler on 64
    workgroups, the heads beyond 32 rows): exact mode, FREE-running greedy tokens of every sequence equal its own oracle run, in
    every state of the options that pick the wide step's launch form."""
    from voicecraft_amd import synth
    from voicecraft_amd.engine import VoiceCraftEngine
    a = synth.make_args(preset)
    sd = synth.make_state_dict(a, seed=4, head_gain=4.0)
    prompts = [synth.random_prompt(a, 4 + (u % 5), 17 + 3 * (u % 7), seed=800 + u) for u in range(B)]
    _, want_res = _oracle_traces(a, sd, prompts)
    eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype="fp32", max_seqs=B, max_positions=256)
    for wh in (1, 0):
        for mt in WIDE_FORMS:
            eng.set_option("wide_heads", wh)
            for name, value in mt:
                eng.set_option(name, value)
            c0 = eng.launch_counts()
            got = _free_running_multi(eng, prompts)
            c = _delta(eng.launch_counts(), c0)
            assert (c["wd"] > 0 and c["mt2"] + c["mt4"] == 0) if mt[0][1] else (c["mt2"] + c["mt4"] > 0 and c["wd"] == 0), (wh, mt, c)
            for u, (g, w) in enumerate(zip(got, want_res)):
                assert np.array_equal(g, w), (wh, mt, u)



@pytest.mark.parametrize("preset,B,gsteps", [("tiny128", 24, 8), ("tiny_h16", 12, 3)])
def test_sampled_tokens_of_a_shrinking_batch_are_the_same_in_every_run(preset, B, gsteps):
    """bf16, top-k sampling, live terminators, a batch that shrinks: the width a step runs at decides which kernels round its logits, so the
    step at which the host re-packs the batch must not depend on how far the device has run ahead of it.  The sampler leaves "sequences still
    live" in the slot of its batch of graph_steps steps and the host reads the slot of a batch it has seen END (SampleArgs.step_ctr): six
    runs of one seeded call, with the host disturbed differently before each, give the same tokens and the same number of re-packs."""
    import time
    from voicecraft_amd import synth
    from voicecraft_amd.engine import VoiceCraftEngine
    a = synth.make_args(preset)
    sd = synth.make_state_dict(a, seed=4, mute_eos=False, boost=[(0, 2051, 0.45)])
    prompts = [synth.random_prompt(a, 4 + (u % 5), 9 + 3 * (u % 7), seed=700 + u) for u in range(B)]
    eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype="bf16", max_seqs=B, max_positions=256)
    eng.set_option("graph_steps", gsteps)
    runs, repacks = [], []
    for i in range(6):
        if i % 2:
            time.sleep(0.02 * i)                      # another phase between the host's loop and the device's
        outs = eng.inference_tts_multi([p[0][0] for p in prompts], [p[2][0] for p in prompts], top_k=40, stop_repetition=3, _seed=11)
        runs.append([res.cpu().numpy() for (res, gen) in outs])
        repacks.append(int(eng.debug_read("host_ms", (8,), torch.float64)[6]))
    assert repacks[0] >= 1 and len(set(repacks)) == 1, repacks
    lens = [r.shape[2] for r in runs[0]]
    assert len(set(lens)) > 1, lens                    # sequences really retire at different steps
    for i in range(1, 6):
        for u in range(B):
            assert runs[i][u].shape == runs[0][u].shape and np.array_equal(runs[i][u], runs[0][u]), (i, u)


_RAGGED_ORACLE = {}      # (preset, B) -> the oracle's outputs: shared by the graph / eager runs of a shape


@pytest.mark.parametrize("preset,B,graph", [("tiny_h16", 64, True), ("tiny_h16", 64, False), ("tiny128", 40, True), ("tiny128", 20, False), ("tiny", 9, True)])
def test_wide_batch_whose_sequences_retire_at_different_steps(preset, B, graph):
    """A batch with LIVE terminators (the `boost` of oracle/gen_golden.py's un-muted cases): the sequences end anywhere between
    ~11 and ~50 generated frames, so most rows of a 17+-row step go idle long before the last one ends - sampled / arg-max
    terminator, min-length guard, EOG tail and retirement of each row inside a wide step.  Exact mode: every sequence must equal
    its own oracle run (models/voicecraft.py:1041-1045 per sequence), on the captured graph and eagerly."""
    from oracle.voicecraft_oracle import VoiceCraftOracle
    from voicecraft_amd import synth
    from voicecraft_amd.engine import VoiceCraftEngine
    a = synth.make_args(preset)
    sd = synth.make_state_dict(a, seed=4, mute_eos=False, boost=[(0, 2051, 0.45)])
    prompts = [synth.random_prompt(a, 4 + (u % 5), 9 + 3 * (u % 7), seed=700 + u) for u in range(B)]
    if (preset, B) not in _RAGGED_ORACLE:
        torch.set_num_threads(min(4, torch.get_num_threads()))
        orc = VoiceCraftOracle(a, sd)
        _RAGGED_ORACLE[(preset, B)] = [orc.inference_tts(xx, xl, yy, top_k=1, stop_repetition=3)[0].numpy() for (xx, xl, yy) in prompts]
    want = _RAGGED_ORACLE[(preset, B)]
    lens = [w.shape[2] - p[2].shape[1] for w, p in zip(want, prompts)]
    assert min(lens) * 2 <= max(lens), lens                       # the workload really is ragged
    eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype="fp32", max_seqs=B, max_positions=256, use_graph=graph)
    for shrink in (1, 0):      # 1 (default): the live sequences are re-packed onto narrower steps as the others retire; 0: fixed width
        eng.set_option("shrink", shrink)
        got = _free_running_multi(eng, prompts)
        for u, (g, w) in enumerate(zip(got, want)):
            assert g.shape == w.shape and np.array_equal(g, w), (shrink, u, g.shape, w.shape)
        assert eng.last_steps == max(lens) + a.n_codebooks
        repacks = int(eng.debug_read("host_ms", (8,), torch.float64)[6])
        assert (repacks >= 1) if shrink else (repacks == 0), (shrink, repacks)
