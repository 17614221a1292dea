"""EnCodec boundary, CPU side: weight-norm folding and key normalisation against the
transformers.EncodecModel restatement, and the integer facts the reference pins
(hop 320, 50 Hz, K=4, 2048 codes: data/phonemize_encodec_encode_hf.py:11-13, config.py:51)."""
import numpy as np
import pytest
import torch

from oracle import encodec_oracle as eo
from voicecraft_amd import synth
from voicecraft_amd.codec import DEFAULT_CFG, fold_weight_norm, normalize_state_dict


@pytest.fixture(scope="module")
def model_and_sd():
    sd = synth.make_codec_state_dict(0)
    return eo.build(sd), sd


def test_folded_weights_equal_module_weights(model_and_sd):
    m, sd = model_and_sd
    n = normalize_state_dict(sd)
    seen = 0
    for name, mod in m.named_modules():
        if hasattr(mod, "conv") and hasattr(mod.conv, "weight") and name:
            w = n[name + ".conv.weight"]
            assert torch.allclose(w, mod.conv.weight.detach(), atol=1e-6), name
            seen += 1
    assert seen == 28
    assert sum(p.numel() for p in m.parameters()) == 56_819_138       # "56M parameters" (README.md:198)


def test_audiocraft_style_keys_are_mapped():
    g, v = torch.rand(8, 1, 1) + 0.5, torch.randn(8, 4, 3)
    sd = {"encoder.model.3.conv.conv.weight_g": g, "encoder.model.3.conv.conv.weight_v": v,
          "encoder.model.3.conv.conv.bias": torch.zeros(8),
          "decoder.model.3.convtr.convtr.weight_g": g, "decoder.model.3.convtr.convtr.weight_v": v,
          "quantizer.vq.layers.2._codebook.embed": torch.zeros(4, 2)}
    n = normalize_state_dict(sd)
    assert set(n) == {"encoder.layers.3.conv.weight", "encoder.layers.3.conv.bias", "decoder.layers.3.conv.weight",
                      "quantizer.layers.2.codebook.embed"}
    assert torch.allclose(n["encoder.layers.3.conv.weight"], fold_weight_norm(g, v))
    assert torch.allclose(n["encoder.layers.3.conv.weight"].reshape(8, -1).norm(dim=1), g.reshape(-1), atol=1e-5)


@pytest.mark.parametrize("n", [16000, 16001, 47999, 320 * 7])
def test_frame_arithmetic(model_and_sd, n):
    m, _ = model_and_sd
    torch.manual_seed(n)
    codes, z = eo.encode(m, torch.randn(1, 1, n) * 0.1)
    T = -(-n // 320)
    assert codes.shape == (DEFAULT_CFG["n_q"], T) and z.shape == (T, 128)
    assert codes.min() >= 0 and codes.max() < 2048
    wav = eo.decode(m, codes)
    assert wav.shape == (320 * T,)


def test_codes_text_format_round_trip(tmp_path):
    """K lines of space-separated ints, no trailing newline (data/phonemize_encodec_encode_hf.py:50-54), read back the way
    the reference's dataset does (data/gigaspeech.py:41-62, incl. the special_first shift)."""
    from voicecraft_amd.codec import read_codes_txt, write_codes_txt
    codes = torch.from_numpy(np.random.RandomState(0).randint(0, 2048, size=(4, 37)).astype(np.int64))
    p = tmp_path / "seg.txt"
    write_codes_txt(codes, str(p))
    text = p.read_text()
    assert text.count("\n") == 3 and not text.endswith("\n")
    assert text.split("\n")[0] == " ".join(str(int(v)) for v in codes[0])
    assert read_codes_txt(str(p), 4) == codes.tolist()
    assert read_codes_txt(str(p), 4, special_first=1, n_special=4) == (codes + 4).tolist()
    assert read_codes_txt(str(p), 3) == codes[:3].tolist()           # "k < n_codebooks"
    with pytest.raises(AssertionError):
        read_codes_txt(str(p), 5)


def test_bulk_encode_follows_the_reference_batching():
    """bulk_encode = sort longest first, batches of batch_size, zero-pad to the batch's longest, ONE encode per batch
    (two halves when the longest clip exceeds max_len), cut each clip to round(seconds * 50) frames - checked with a
    stand-in tokenizer that records what it is handed."""
    from voicecraft_amd.codec import bulk_encode

    class Tok:
        sample_rate = 16000
        calls = []

        def encode(self, wav):                     # [B,1,N] -> [( [B,K,T], None )]: frame t of clip b = b-th row marker + t
            self.calls.append(tuple(wav.shape))
            B, _, N = wav.shape
            T = -(-N // 320)
            first = (wav[:, 0, 0] * 1000).round().long()            # each clip starts with its id / 1000
            codes = first[:, None, None] * 10000 + torch.arange(T)[None, None, :].expand(B, 4, T)
            return [(codes, None)]

    lens = [4000, 16000, 8000, 1000, 12000]
    wavs = [torch.full((n,), i / 1000.0) for i, n in enumerate(lens)]
    tok = Tok()
    out = bulk_encode(tok, wavs, batch_size=2, max_len=10000)
    # batches (longest first): [16000, 12000] -> split in two halves (longest > max_len, 2 clips), [8000, 4000], [1000]
    assert tok.calls == [(1, 1, 16000), (1, 1, 16000), (2, 1, 8000), (1, 1, 1000)]
    for i, n in enumerate(lens):
        T = round(n / 16000 * 50)
        assert out[i].shape == (4, T)
        assert torch.equal(out[i][0], i * 10000 + torch.arange(T))
