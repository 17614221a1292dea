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


def audiocraft_checkpoint_keys():
    """The tensor names of an audiocraft EncodecModel `best_state` at the VoiceCraft codec shape (encodec_4cb2048_giga.th;
    audiocraft @ c5157b5, README.md:105), written out from its constructors: SEANetEncoder.model = [conv, (resblock,
    ELU, down-conv) x 4 ratios (reversed), LSTM, ELU, conv]; SEANetDecoder.model = [conv, LSTM, (ELU, up-convtr,
    resblock) x 4, ELU, conv]; StreamableConv1d -> NormConv1d (`.conv.conv`, weight_norm: weight_g / weight_v),
    StreamableConvTranspose1d (`.convtr.convtr`); SEANetResnetBlock.block = [ELU, conv k3, ELU, conv k1] with an
    identity skip (true_skip); StreamableLSTM.lstm; ResidualVectorQuantizer.vq.layers[q]._codebook with its EMA
    buffers.  -> {name: shape}"""
    F, ratios, H, Kc, D = 64, [8, 5, 4, 2], 128, 2048, 128
    keys = {}

    def conv(prefix, co, ci, k, tr=False):
        inner = "convtr.convtr" if tr else "conv.conv"
        keys[f"{prefix}.{inner}.weight_g"] = ((ci if tr else co), 1, 1)
        keys[f"{prefix}.{inner}.weight_v"] = (ci, co, k) if tr else (co, ci, k)
        keys[f"{prefix}.{inner}.bias"] = (co,)

    def res(prefix, dim):
        conv(f"{prefix}.block.1", dim // 2, dim, 3)
        conv(f"{prefix}.block.3", dim, dim // 2, 1)

    def lstm(prefix, h):
        for n in range(2):
            keys[f"{prefix}.lstm.weight_ih_l{n}"] = (4 * h, h)
            keys[f"{prefix}.lstm.weight_hh_l{n}"] = (4 * h, h)
            keys[f"{prefix}.lstm.bias_ih_l{n}"] = (4 * h,)
            keys[f"{prefix}.lstm.bias_hh_l{n}"] = (4 * h,)

    conv("encoder.model.0", F, 1, 7)
    i, ch = 1, F
    for r in reversed(ratios):
        res(f"encoder.model.{i}", ch)
        conv(f"encoder.model.{i + 2}", 2 * ch, ch, 2 * r)
        i, ch = i + 3, 2 * ch
    lstm(f"encoder.model.{i}", ch)
    conv(f"encoder.model.{i + 2}", H, ch, 7)
    conv("decoder.model.0", ch, H, 7)
    lstm("decoder.model.1", ch)
    i = 2
    for r in ratios:
        conv(f"decoder.model.{i + 1}", ch // 2, ch, 2 * r, tr=True)
        res(f"decoder.model.{i + 2}", ch // 2)
        i, ch = i + 3, ch // 2
    conv(f"decoder.model.{i + 1}", 1, F, 7)
    for q in range(4):
        p = f"quantizer.vq.layers.{q}._codebook"
        keys[p + ".inited"] = (1,)
        keys[p + ".cluster_size"] = (Kc,)
        keys[p + ".embed"] = (Kc, D)
        keys[p + ".embed_avg"] = (Kc, D)
    return keys


def test_a_whole_audiocraft_checkpoint_key_set_loads_with_strict_coverage():
    """Every tensor name an audiocraft checkpoint of this codec holds - 28 convolutions (weight_g / weight_v / bias), two
    2-layer LSTMs, 4 codebooks with their EMA buffers: 112 tensors - must map onto EXACTLY the canonical names the
    engine's strict loader expects (voicecraft_amd.codec.expected_keys), with the folded weights' shapes right; the
    restatement's own (transformers) names of the same architecture must normalise to the same set.  No arithmetic is
    pinned by this (parity with audiocraft's outputs stays unpinned): it pins the NAME layout the loader accepts."""
    from voicecraft_amd.codec import expected_keys
    names = audiocraft_checkpoint_keys()
    assert len(names) == 28 * 3 + 2 * 8 + 4 * 4
    rs = np.random.RandomState(0)
    sd = {k: torch.from_numpy(rs.standard_normal(size=s).astype(np.float32)) for k, s in names.items()}
    n = normalize_state_dict(sd)
    buffers = (".inited", ".cluster_size", ".embed_avg")
    got = {k for k in n if not k.endswith(buffers)}
    want = expected_keys(DEFAULT_CFG)
    assert got == want, (sorted(got - want)[:5], sorted(want - got)[:5])
    ours = {k for k in normalize_state_dict(synth.make_codec_state_dict(0)) if not k.endswith(buffers)}
    assert ours == want
    # shapes of the folded convolution weights = the restatement's module weights
    m = eo.build(synth.make_codec_state_dict(0))
    for name, mod in m.named_modules():
        if hasattr(mod, "conv") and hasattr(mod.conv, "weight") and name:
            assert tuple(n[name + ".conv.weight"].shape) == tuple(mod.conv.weight.shape), name
    g, v = sd["encoder.model.3.conv.conv.weight_g"], sd["encoder.model.3.conv.conv.weight_v"]
    assert torch.allclose(n["encoder.layers.3.conv.weight"], fold_weight_norm(g, v))
