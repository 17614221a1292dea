"""Model-shape presets and synthetic checkpoints.

There is no network, so benchmarks and tests run on random-init weights of the reference
architecture.  Values are drawn with numpy's legacy `RandomState` (a frozen stream: the same seed
gives the same checkpoint here, on the GPU box and in the golden-vector generator), with the
distributions of the reference's default initialisation (models/modules/activation.py:280-290,
transformer.py:53-56, voicecraft.py:152, torch's Linear/Embedding defaults).  `perturb=True`
additionally randomises the tensors the default init leaves at 0/1 (biases, LayerNorm affine,
alpha) so that every term of the computation is exercised by the parity tests.
"""
from __future__ import annotations

from argparse import Namespace

import numpy as np
import torch

PRESETS = {
    # name: (d_model, nhead, num_decoder_layers)
    "tiny": (256, 4, 2),          # head_dim 64
    "tiny128": (512, 4, 2),       # head_dim 128
    "tiny_h16": (512, 16, 2),     # 16 heads of 32: batched decode reaches the 4- and 2-split attention merges
    "giga330M": (1024, 16, 24),   # assumed shape: not stated in the reference tree
    "giga830M": (2048, 16, 16),   # z_scripts/e830M.sh:34-37
}


def make_args(preset: str = "giga830M", *, eos: int = 2051, n_special: int = 4, reduced_eog: int = 1,
              n_codebooks: int = 4, max_n_spans: int = 3) -> Namespace:
    """The `args` Namespace the reference pickles next to a checkpoint (config.py:55-84)."""
    d, h, l = PRESETS[preset]
    return Namespace(
        d_model=d, nhead=h, num_decoder_layers=l, n_codebooks=n_codebooks, audio_embedding_dim=d,
        audio_vocab_size=2048, text_vocab_size=100, text_pad_token=100, empty_token=2048, eog=2049,
        audio_pad_token=2050, eos=eos, n_special=n_special, special_first=0, reduced_eog=reduced_eog,
        max_n_spans=max_n_spans, encodec_sr=50, shuffle_mask_embedding=0,
        text_embedding_dropout=0.0, audio_embedding_dropout=0.0, text_positional_embedding_dropout=0.0,
        audio_positional_embedding_dropout=0.0, trm_dropout=0.0,
    )


def make_state_dict(args: Namespace, seed: int = 0, perturb: bool = True, mute_eos: bool = True,
                    head_gain: float = 1.0, fast: bool = False, boost=None, mute_special: bool = False) -> dict[str, torch.Tensor]:
    """fp32 state_dict with the reference's keys and shapes.

    mute_eos: bias of the terminator logit = -1e4 on every head, so that no terminator is ever
      sampled and generation stops on the reference's own length cap (deterministic T_gen,
      BASELINE.md §4.2).
    head_gain: scales the last head matrices; > 1 gives peaked distributions (clear arg-max margins).
    fast: draw with torch's CPU generator instead of numpy's RandomState (10x faster for the
      multi-GB presets; deterministic for a given torch build, not used by the golden fixtures).
    mute_special: the same bias on EVERY special token (empty / eog / pad / eos), so that generated frames hold plain
      codes only and can be handed to the codec (a trained model never emits them mid-utterance; random weights do).
    boost: iterable of (codebook, token, delta): `predict_layer[codebook][2].bias[token] += delta` after
      everything else.  With mute_eos=False this makes the terminator / a silence token likely, so
      that the end-of-generation and silence-penalty branches of the state machine actually fire
      (voicecraft.py:1024-1045); codebook -1 = every head.
    """
    rs = np.random.RandomState(seed)
    tg = torch.Generator().manual_seed(seed) if fast else None
    d, L, K = args.d_model, args.num_decoder_layers, args.n_codebooks
    V = args.audio_vocab_size + args.n_special
    P = args.audio_vocab_size // 2
    sd: dict[str, torch.Tensor] = {}

    def uni(shape, bound):
        if fast:
            return (torch.rand(shape, generator=tg, dtype=torch.float32) * 2 - 1) * bound
        return torch.from_numpy(rs.uniform(-bound, bound, size=shape).astype(np.float32))

    def nrm(shape, std=1.0):
        if fast:
            return torch.randn(shape, generator=tg, dtype=torch.float32) * std
        return torch.from_numpy((rs.standard_normal(size=shape) * std).astype(np.float32))

    def small(shape, base):
        if perturb:
            return torch.from_numpy((base + 0.1 * rs.standard_normal(size=shape)).astype(np.float32))
        return torch.full(shape, float(base), dtype=torch.float32)

    sd["mask_embedding"] = nrm((args.max_n_spans, d))
    sd["text_embedding.word_embeddings.weight"] = nrm((args.text_vocab_size + 1, d))
    for k in range(K):
        sd[f"audio_embedding.{k}.word_embeddings.weight"] = nrm((V, d))
    sd["text_positional_embedding.alpha"] = small((1,), 1.0)
    sd["audio_positional_embedding.alpha"] = small((1,), 1.0)
    for l in range(L):
        p = f"decoder.layers.{l}."
        sd[p + "self_attn.in_proj_weight"] = uni((3 * d, d), (6.0 / (3 * d + d)) ** 0.5)   # xavier_uniform
        sd[p + "self_attn.in_proj_bias"] = small((3 * d,), 0.0)
        sd[p + "self_attn.out_proj.weight"] = uni((d, d), d ** -0.5)
        sd[p + "self_attn.out_proj.bias"] = small((d,), 0.0)
        sd[p + "linear1.weight"] = uni((4 * d, d), d ** -0.5)
        sd[p + "linear1.bias"] = uni((4 * d,), d ** -0.5)
        sd[p + "linear2.weight"] = uni((d, 4 * d), (4 * d) ** -0.5)
        sd[p + "linear2.bias"] = uni((d,), (4 * d) ** -0.5)
        sd[p + "norm1.weight"] = small((d,), 1.0)
        sd[p + "norm1.bias"] = small((d,), 0.0)
        sd[p + "norm2.weight"] = small((d,), 1.0)
        sd[p + "norm2.bias"] = small((d,), 0.0)
    sd["decoder.norm.weight"] = small((d,), 1.0)
    sd["decoder.norm.bias"] = small((d,), 0.0)
    term = args.eos if args.eos > 0 else args.eog
    for k in range(K):
        p = f"predict_layer.{k}."
        sd[p + "0.weight"] = uni((P, d), d ** -0.5)
        sd[p + "0.bias"] = uni((P,), d ** -0.5)
        sd[p + "2.weight"] = uni((V, P), P ** -0.5) * head_gain
        b = uni((V,), P ** -0.5)
        if mute_eos:
            b[term] = -1e4
        if mute_special:
            b[args.audio_vocab_size:] = -1e4
        for (bk, tok, delta) in (boost or ()):
            if bk == k or bk < 0:
                b[int(tok)] += float(delta)
        sd[p + "2.bias"] = b
    return sd


def random_prompt(args: Namespace, Lx: int, T: int, seed: int = 1):
    """x int64 [1,Lx], x_lens [1], y int64 [1,T,K] — the synthetic inputs of BASELINE.md §4.2."""
    rs = np.random.RandomState(seed)
    x = torch.from_numpy(rs.randint(0, args.text_vocab_size, size=(1, Lx)).astype(np.int64))
    y = torch.from_numpy(rs.randint(0, args.audio_vocab_size, size=(1, T, args.n_codebooks)).astype(np.int64))
    return x, torch.tensor([Lx], dtype=torch.int64), y


def make_codec_state_dict(seed: int = 0, use_conv_shortcut: bool = False, num_residual_layers: int = 1) -> dict[str, torch.Tensor]:
    """Synthetic EnCodec weights in transformers.EncodecModel naming (weight-norm g/v pairs, LSTM,
    codebooks) at the VoiceCraft codec shape.  Magnitudes keep activations O(1) through 16 layers."""
    rs = np.random.RandomState(seed)
    sd: dict[str, torch.Tensor] = {}

    def t(a):
        return torch.from_numpy(np.ascontiguousarray(a).astype(np.float32))

    def conv(prefix, co, ci, k, transposed=False):
        shape = (ci, co, k) if transposed else (co, ci, k)
        fan_in = ci * k if not transposed else ci * k / max(1, k // 2)
        v = rs.standard_normal(size=shape)
        g = np.sqrt((v.reshape(shape[0], -1) ** 2).sum(1)) * (1.4 / np.sqrt(fan_in)) * (0.8 + 0.4 * rs.rand(shape[0]))
        sd[prefix + ".conv.parametrizations.weight.original0"] = t(g.reshape(-1, 1, 1))
        sd[prefix + ".conv.parametrizations.weight.original1"] = t(v)
        sd[prefix + ".conv.bias"] = t(0.05 * rs.standard_normal(size=(co,)))

    def lstm(prefix, h, layers=2):
        for n in range(layers):
            b = h ** -0.5
            sd[f"{prefix}.lstm.weight_ih_l{n}"] = t(rs.uniform(-b, b, size=(4 * h, h)))
            sd[f"{prefix}.lstm.weight_hh_l{n}"] = t(rs.uniform(-b, b, size=(4 * h, h)))
            sd[f"{prefix}.lstm.bias_ih_l{n}"] = t(rs.uniform(-b, b, size=(4 * h,)))
            sd[f"{prefix}.lstm.bias_hh_l{n}"] = t(rs.uniform(-b, b, size=(4 * h,)))

    F, ratios, hidden = 64, [8, 5, 4, 2], 128
    conv("encoder.layers.0", F, 1, 7)
    idx, ch = 1, F
    def res_unit(prefix, dim):
        conv(prefix + ".block.1", dim // 2, dim, 3)
        conv(prefix + ".block.3", dim, dim // 2, 1)
        if use_conv_shortcut:
            conv(prefix + ".shortcut", dim, dim, 1)

    for r in reversed(ratios):
        for _ in range(num_residual_layers):
            res_unit(f"encoder.layers.{idx}", ch)
            idx += 1
        idx += 1
        conv(f"encoder.layers.{idx}", ch * 2, ch, 2 * r)
        idx += 1
        ch *= 2
    lstm(f"encoder.layers.{idx}", ch)
    idx += 2
    conv(f"encoder.layers.{idx}", hidden, ch, 7)
    conv("decoder.layers.0", ch, hidden, 7)
    lstm("decoder.layers.1", ch)
    idx = 2
    for r in ratios:
        idx += 1
        conv(f"decoder.layers.{idx}", ch // 2, ch, 2 * r, transposed=True)
        idx += 1
        for _ in range(num_residual_layers):
            res_unit(f"decoder.layers.{idx}", ch // 2)
            idx += 1
        ch //= 2
    idx += 1
    conv(f"decoder.layers.{idx}", 1, F, 7)
    for q in range(4):
        sd[f"quantizer.layers.{q}.codebook.embed"] = t(rs.standard_normal(size=(2048, hidden)) * (0.6 ** q))
    return sd
