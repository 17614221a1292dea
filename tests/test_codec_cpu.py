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
