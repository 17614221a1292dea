"""-m gpu: the opt-in stream engine (VC_STREAM=1: the batch-1 decode step as ONE persistent launch over all layers,
vc_stream.hip) against the oracle and against the launch path - bf16, teacher-forced per-step logits within the 2e-2 bar,
at head_dim 128 (16 attention splits per head), head_dim 32 (4 splits, 4 lanes per cached row), one layer, and the full
giga830M shape; the launch census must show the persistent kernel.  Also the experimental weight prefetcher of the launch
path (VC_PREFETCH): results must not change."""
import numpy as np
import pytest
import torch

from test_gpu_model import rel_l2

pytestmark = pytest.mark.gpu


def run_both(a, sd, x, xl, y, forced, n, monkeypatch, env):
    from voicecraft_amd.engine import VoiceCraftEngine
    out = {}
    for mode in ("launch", "alt"):
        for k, v in env.items():
            if mode == "alt":
                monkeypatch.setenv(k, v)
            else:
                monkeypatch.delenv(k, raising=False)
        eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype="bf16", max_seqs=1, max_positions=1024)
        c0 = eng.launch_counts()
        res, gen, lg = eng.inference_tts(x.cuda(), xl.cuda(), y.cuda(), top_k=1, stop_repetition=3, _forced=forced, _logit_steps=n)
        out[mode] = (res.cpu().numpy(), lg.cpu().numpy(), eng.launch_counts()["persist"] - c0["persist"])
        del eng
    for k in env:
        monkeypatch.delenv(k, raising=False)
    return out


@pytest.mark.parametrize("preset,layers", [("tiny128", 2), ("tiny_h16", 2), ("tiny128", 1)])
def test_stream_engine_matches_the_oracle_and_the_launch_path(preset, layers, monkeypatch):
    from oracle.voicecraft_oracle import VoiceCraftOracle
    from voicecraft_amd import synth
    a = synth.make_args(preset)
    a.num_decoder_layers = layers
    sd = synth.make_state_dict(a, seed=3)
    x, xl, y = synth.random_prompt(a, 7, 20, seed=5)
    trace = []
    res_o, _ = VoiceCraftOracle(a, sd).inference_tts(x, xl, y, top_k=1, stop_repetition=3, trace=trace)
    want = torch.stack([t["logits"][0] for t in trace]).numpy()
    forced = torch.stack([t["tokens"] for t in trace]).numpy()
    out = run_both(a, sd, x, xl, y, forced, len(trace), monkeypatch, {"VC_STREAM": "1"})
    assert out["alt"][2] > 0 and out["launch"][2] == 0
    assert np.array_equal(out["alt"][0], res_o.numpy())
    assert rel_l2(out["alt"][1], want).max() <= 2e-2
    assert rel_l2(out["alt"][1], out["launch"][1]).max() <= 1e-2


def test_stream_engine_full_size(monkeypatch):
    from oracle.voicecraft_oracle import VoiceCraftOracle
    from voicecraft_amd import synth
    a = synth.make_args("giga830M")
    sd = synth.make_state_dict(a, seed=0, fast=True)
    x, xl, y = synth.random_prompt(a, 80, 150, seed=1)
    n, K = 40, a.n_codebooks
    forced = np.random.RandomState(3).randint(0, 2048, size=(n, K)).astype(np.int64)
    for j in range(K):
        forced[n - K + j, :j] = a.empty_token
        forced[n - K + j, j] = a.eos
    steps = [0, 1, 20, n - 1]
    torch.set_num_threads(min(16, torch.get_num_threads() or 1) or 1)
    want = VoiceCraftOracle(a, sd).tts_logits_for_trajectory(x, y, forced, steps=steps).numpy()
    out = run_both(a, sd, x, xl, y, forced, n, monkeypatch, {"VC_STREAM": "1"})
    assert out["alt"][2] > 0
    assert rel_l2(out["alt"][1][steps], want).max() <= 2e-2


def test_weight_prefetcher_does_not_change_results(monkeypatch):
    from voicecraft_amd import synth
    a = synth.make_args("tiny128")
    sd = synth.make_state_dict(a, seed=3)
    x, xl, y = synth.random_prompt(a, 7, 20, seed=5)
    forced = np.random.RandomState(1).randint(0, 2048, size=(30, 4)).astype(np.int64)
    for j in range(4):
        forced[26 + j, :j] = a.empty_token
        forced[26 + j, j] = a.eos
    out = run_both(a, sd, x, xl, y, forced, 30, monkeypatch, {"VC_PREFETCH": "4"})
    assert np.array_equal(out["alt"][0], out["launch"][0])
    assert np.array_equal(out["alt"][1], out["launch"][1])
