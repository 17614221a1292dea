"""-m gpu: the engine stages of one TTS request chained (what `inference_tts_scale.inference_one_sample`, :42-105, does between
its phonemizer and its file writer) - AudioTokenizer.encode -> VoiceCraftEngine.inference_tts / inference_tts_batch ->
AudioTokenizer.decode of prompt + generated and of the generated part - through tests/one_sample_chain.py, with integer
phoneme ids and a synthetic waveform in place of the two front-ends (out of scope)."""
import numpy as np
import pytest
import torch

from voicecraft_amd import synth

pytestmark = pytest.mark.gpu

CFG = dict(top_k=1, top_p=1.0, temperature=1.0, stop_repetition=3, kvcache=1, codec_sr=50, silence_tokens="[1388,1898,131]",
           sample_batch_size=1)


@pytest.fixture(scope="module")
def parts():
    from voicecraft_amd.codec import AudioTokenizer
    from voicecraft_amd.engine import VoiceCraftEngine
    a = synth.make_args("tiny128")
    sd = synth.make_state_dict(a, seed=6, mute_special=True)
    eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype="fp32", max_seqs=4, max_positions=1024)
    tok = AudioTokenizer(synth.make_codec_state_dict(0), device="cuda:0", max_seconds=20.0)
    return a, sd, eng, tok


def test_one_sample_chain_equals_its_three_stages_and_the_oracle(parts):
    """The chain must give exactly what the three stages give when called one after the other, the model stage must
    equal the CPU oracle on the SAME encoded prompt (fp32, greedy), and the lengths must follow the reference's
    identities (voicecraft.py:1146-1147: concat = prompt + generated; 320 samples per frame)."""
    from oracle.voicecraft_oracle import VoiceCraftOracle
    from one_sample_chain import OneSampleChain
    a, sd, eng, tok = parts
    torch.manual_seed(0)
    wav = torch.randn(1, 16000 * 3 + 123) * 0.1                 # 3 s + a ragged tail; the prompt is the first 2 s
    text = np.random.RandomState(1).randint(0, 100, size=(12,))
    out = OneSampleChain(eng, tok, a.n_codebooks, "cuda:0").run(text, wav, CFG, n_prompt_samples=32000)
    concat, gen = out.wave_all, out.wave_new
    T = 100                                                      # 32000 samples / 320
    assert out.prompt_codes.shape[1] == T and out.new_codes.shape[-1] == 12 * 10 - T   # the length cap (10 frames per phoneme)
    assert set(out.seconds) == {"encode", "model", "decode", "total"}
    Tg = int(out.new_codes.shape[-1])
    assert concat.shape == (1, 1, 320 * (T + Tg)) and gen.shape == (1, 1, 320 * Tg)
    assert torch.isfinite(concat).all() and torch.isfinite(gen).all()
    # stage by stage
    codes = tok.encode(wav[:, :32000].reshape(1, 1, -1).cuda())[0][0]                    # [1,K,T]
    y = codes.transpose(2, 1)
    x = torch.from_numpy(text.astype(np.int64)).unsqueeze(0)
    res, g = eng.inference_tts(x.cuda(), torch.tensor([12]).cuda(), y, top_k=1, stop_repetition=3)
    assert torch.equal(tok.decode([(res, None)]), concat) and torch.equal(tok.decode([(g, None)]), gen)
    assert torch.equal(res[:, :, :T], codes)                     # the prompt's codes come back untouched
    assert int(g.max()) < 2048 and int(g.min()) >= 0
    want = VoiceCraftOracle(a, sd).inference_tts(x, torch.tensor([12]), y.cpu(), top_k=1, stop_repetition=3)[0]
    assert np.array_equal(res.cpu().numpy(), want.numpy())


def test_one_sample_chain_best_of_n(parts):
    from one_sample_chain import OneSampleChain
    a, sd, eng, tok = parts
    torch.manual_seed(1)
    wav = torch.randn(16000 * 2) * 0.1
    cfg = dict(CFG, top_k=40, sample_batch_size=3, silence_tokens=[1388, 1898, 131])
    out = OneSampleChain(eng, tok, a.n_codebooks, "cuda:0").run(torch.arange(15), wav, cfg)
    concat, gen = out.wave_all, out.wave_new
    assert concat.shape[-1] - gen.shape[-1] == 32000 and gen.shape[-1] % 320 == 0 and gen.shape[-1] > 0
