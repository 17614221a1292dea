"""EnCodec — CPU oracle.  TEST INFRASTRUCTURE ONLY.  **Parity unpinned.**

The reference's codec is audiocraft's EncodecModel (audiocraft @ c5157b5bf14bf83449c17ea1eeb66c19fb4bc7f0,
`README.md:105`, `data/tokenizer.py:109-110`), which is neither vendored in the reference tree nor
installable here, and no reference test pins its outputs.  The restatement used as oracle is the
published `transformers.EncodecModel` implementation (transformers is part of this image), configured to
the VoiceCraft codec shape (16 kHz mono, 64 filters, ratios 8/5/4/2, 2-layer LSTM,
4 x 2048 x 128 RVQ, weight norm, reflect padding, non-causal; 56.8 M parameters).  Which of
use_causal_conv / pad_mode / use_conv_shortcut the real checkpoint used cannot be known from the tree.
"""
from __future__ import annotations

import torch


def build(state_dict: dict[str, torch.Tensor] | None = None, **overrides):
    """overrides: the architecture switches the reference tree does not pin (use_causal_conv, pad_mode,
    use_conv_shortcut, num_residual_layers, dilation_growth_rate), as EncodecConfig names them."""
    from transformers import EncodecConfig, EncodecModel
    kw = dict(target_bandwidths=[2.2], sampling_rate=16000, audio_channels=1, normalize=False,
              chunk_length_s=None, hidden_size=128, num_filters=64, num_residual_layers=1,
              upsampling_ratios=[8, 5, 4, 2], norm_type="weight_norm", kernel_size=7, last_kernel_size=7,
              residual_kernel_size=3, dilation_growth_rate=2, use_causal_conv=False, pad_mode="reflect",
              compress=2, num_lstm_layers=2, trim_right_ratio=1.0, codebook_size=2048, codebook_dim=128,
              use_conv_shortcut=False)
    kw.update(overrides)
    cfg = EncodecConfig(**kw)
    m = EncodecModel(cfg).eval()
    if state_dict is not None:
        missing, unexpected = m.load_state_dict(state_dict, strict=False)
        assert not unexpected, unexpected
        ok = ("stride", "kernel_size", "padding_total", "inited", "cluster_size", "embed_avg")   # buffers without information
        assert all(k.endswith(ok) for k in missing), missing
    return m


@torch.no_grad()
def encode(m, wav: torch.Tensor):
    """wav [1,1,N] -> (codes int64 [K,T], latent fp32 [T,128])"""
    z = m.encoder(wav)                                   # [1,128,T]
    codes = m.quantizer.encode(z, None)                  # [K,1,T]
    return codes[:, 0], z[0].transpose(0, 1).contiguous()


@torch.no_grad()
def decode(m, codes: torch.Tensor):
    """codes int64 [K,T] -> wav fp32 [320*T]"""
    q = m.quantizer.decode(codes.unsqueeze(1))           # [1,128,T]
    return m.decoder(q)[0, 0]
