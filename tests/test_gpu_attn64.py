"""-m gpu: the second prefill attention kernel (tile_attn64_k, option tile_attn = 2: 64 query rows per workgroup, transposed
score product, P in registers; bf16, head_dim 128) against the first one and against the oracle - the first decode step's head
logits depend on the prefill attention of EVERY prompt row through every layer.  One long prompt (several 64-row blocks, a ragged
last block), an editing prompt (rearranged pieces), several ragged utterances in one row stream with and without a shared text
prefix (blocks of different sequences side by side), and a timing line of the attention microbenchmark for both kernels."""
import numpy as np
import pytest
import torch

from test_gpu_model import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model():
    from oracle.voicecraft_oracle import VoiceCraftOracle
    from voicecraft_amd import synth
    a = synth.make_args("tiny128")          # d 512, 4 heads of 128, 2 layers
    sd = synth.make_state_dict(a, seed=5)
    return a, sd, VoiceCraftOracle(a, sd)


def both(eng, fn):
    out = {}
    for k in (1, 2):
        eng.set_option("tile_attn", f"{k},128")        # (the default takes the second kernel from 768 prompt rows per call)
        out[k] = fn()
    eng.set_option("tile_attn", "2,768")
    return out


@pytest.mark.parametrize("Lx,T", [(30, 500), (17, 173), (40, 215)])
def test_long_prompt_first_step_logits(model, Lx, T):
    from voicecraft_amd import synth
    from voicecraft_amd.engine import VoiceCraftEngine
    a, sd, orc = model
    x, xl, y = synth.random_prompt(a, Lx, T, seed=Lx)
    tr = []
    orc.inference_tts(x, xl, y, top_k=1, stop_repetition=3, trace=tr, max_steps=2)
    want = tr[0]["logits"][0].numpy()[None]
    eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype="bf16", max_seqs=1, max_positions=1024)
    forced = torch.stack([t["tokens"] for t in tr if "tokens" in t]).numpy()
    got = both(eng, lambda: eng.inference_tts(x.cuda(), xl.cuda(), y.cuda(), top_k=1, stop_repetition=3, _forced=forced, _logit_steps=1)[2].cpu().numpy())
    assert rel_l2(got[1], want).max() <= 2e-2 and rel_l2(got[2], want).max() <= 2e-2, (rel_l2(got[1], want), rel_l2(got[2], want))
    assert rel_l2(got[2], got[1]).max() <= 1e-2, rel_l2(got[2], got[1])


def test_editing_prompt_and_a_ragged_batch_with_a_shared_prefix(model):
    from voicecraft_amd import synth
    from voicecraft_amd.engine import VoiceCraftEngine
    a, sd, orc = model
    eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype="bf16", max_seqs=5, max_positions=1024)
    # (a) editing: two spans of a 300-frame utterance (rearranged prompt of ~330 columns)
    x, xl, y = synth.random_prompt(a, 25, 300, seed=9)
    mi = torch.tensor([[[60, 90], [200, 240]]], dtype=torch.int64)
    got = both(eng, lambda: eng.inference(x.cuda(), xl.cuda(), y.cuda(), mi, top_k=1, _logit_steps=1, _seed=3)[1].cpu().numpy())
    assert rel_l2(got[2][None], got[1][None]).max() <= 1e-2
    # (b) five ragged utterances in one row stream, then the same with a shared text prefix
    prompts = [synth.random_prompt(a, 12 + 3 * u, 40 + 37 * u, seed=70 + u) for u in range(5)]
    xs, ys = [p[0][0] for p in prompts], [p[2][0] for p in prompts]
    got = both(eng, lambda: eng.inference_tts_multi(xs, ys, top_k=1, stop_repetition=3, _logit_steps=1, _seed=3)[1].cpu().numpy())
    assert rel_l2(got[2][0], got[1][0]).max() <= 1e-2, rel_l2(got[2][0], got[1][0])
    for u in (0, 2, 4):
        tr = []
        orc.inference_tts(prompts[u][0], prompts[u][1], prompts[u][2], top_k=1, stop_repetition=3, trace=tr, max_steps=1)
        assert rel_l2(got[2][0, u][None], tr[0]["logits"][0].numpy()[None]).max() <= 2e-2, u
    pre = torch.arange(10, dtype=torch.int64) % 7
    xs2 = [torch.cat([pre, v]) for v in xs]
    got = both(eng, lambda: eng.inference_tts_multi(xs2, ys, top_k=1, stop_repetition=3, _logit_steps=1, _seed=3, _shared_text_prefix=10)[1].cpu().numpy())
    assert rel_l2(got[2][0], got[1][0]).max() <= 1e-2, rel_l2(got[2][0], got[1][0])


def test_pass_size_that_is_not_a_multiple_of_64_keeps_the_first_kernel(model):
    """`prefill_rows` accepts any multiple of 16.  With 1040 rows per pass a 64-row block of the second pass would
    straddle one prompt's tail padding and the next prompt's first rows; the engine must not pick tile_attn64_k then (census) and
    the ragged batch's first-step logits must equal the default pass size's."""
    from voicecraft_amd import synth
    from voicecraft_amd.engine import VoiceCraftEngine
    a, sd, orc = model
    eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype="bf16", max_seqs=4, max_positions=1024, use_graph=False)
    prompts = [synth.random_prompt(a, 20 + 5 * u, 790 - 250 * u, seed=90 + u) for u in range(4)]      # 811 + 566 + 321 + 76 rows: one >= 768
    xs, ys = [p[0][0] for p in prompts], [p[2][0] for p in prompts]
    run = lambda: eng.inference_tts_multi(xs, ys, top_k=1, stop_repetition=3, _logit_steps=1, _seed=3)[1].cpu().numpy()
    eng.set_option("tile_attn", "2,768")
    c0 = eng.launch_counts()
    base = run()
    assert eng.launch_counts()["tile_attn64"] - c0["tile_attn64"] == a.num_decoder_layers          # the default pass size: one pass, second kernel
    eng.set_option("prefill_rows", "1040")
    c0 = eng.launch_counts()
    got = run()
    c = {k: eng.launch_counts()[k] - c0[k] for k in c0}
    eng.set_option("prefill_rows", "2048")
    assert c["tile_attn"] == 2 * a.num_decoder_layers and c["tile_attn64"] == 0, c          # two passes, both on tile_attn_k
    assert rel_l2(got[0], base[0]).max() <= 1e-2, rel_l2(got[0], base[0])
    for u in (1, 3):
        tr = []
        orc.inference_tts(prompts[u][0], prompts[u][1], prompts[u][2], top_k=1, stop_repetition=3, trace=tr, max_steps=1)
        assert rel_l2(got[0, u][None], tr[0]["logits"][0].numpy()[None]).max() <= 2e-2, u


def test_attention_microbenchmark_of_both_kernels(model, capsys):
    from voicecraft_amd import synth
    from voicecraft_amd.engine import VoiceCraftEngine
    a = synth.make_args("giga830M")
    a.num_decoder_layers = 1
    sd = synth.make_state_dict(a, seed=0, perturb=False, fast=True)
    eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype="bf16", max_seqs=1, max_positions=2304)
    lines = []
    for rows in (512, 800, 2048):
        t = both(eng, lambda: eng.bench_kernel("pf_attn", n_rows=rows, iters=32))
        lines.append(f"pf_attn {rows} rows: tile_attn_k {t[1][0] * 1e3:.1f} us, tile_attn64_k {t[2][0] * 1e3:.1f} us ({t[1][1] / (t[2][0] * 1e-3) / 1e12:.0f} TFLOP/s)")
    with capsys.disabled():
        print("\n" + "\n".join(lines))
    import os
    os.makedirs("gpurun_out", exist_ok=True)
    open("gpurun_out/attn64_probe.log", "w").write("\n".join(lines) + "\n")
