"""EnCodec tokenizer on the MI355X engine — mirror of the reference's `AudioTokenizer`
(data/tokenizer.py:101-133): `.encode(wav[B,1,N]) -> [(codes[B,K,T], None)]`,
`.decode([(codes[B,K,T], None)]) -> wav[B,1,hop*T]`, `.sample_rate`, `.channels`, `.device`.

The reference loads audiocraft's `encodec_4cb2048_giga.th`; neither audiocraft nor the checkpoint is
available offline, so weights arrive as a state_dict.  Two key schemes are accepted:
  * transformers.EncodecModel names (`encoder.layers.3.conv.parametrizations.weight.original0/1`,
    or already-folded `...conv.weight`) — the CPU restatement the parity tests use;
  * audiocraft names (`encoder.model.3.conv.conv.weight_g/_v`, `decoder.model.3.convtr.convtr.*`,
    `quantizer.vq.layers.0._codebook.embed`) — mapped best-effort, UNVERIFIED against a real
    checkpoint (none is reachable from this environment).
Weight norm is folded here (w = g * v / ||v||, torch._weight_norm semantics); all compute is in
libvcengine.so (include/vc_codec.h).  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import re

import torch

from . import _lib

VC_CODEC_MAX_RATIOS = 8


class CodecCfg(C.Structure):
    _fields_ = [("sample_rate", C.c_int32), ("n_filters", C.c_int32), ("n_ratios", C.c_int32),
                ("ratios", C.c_int32 * VC_CODEC_MAX_RATIOS), ("hidden", C.c_int32), ("n_q", C.c_int32),
                ("codebook_size", C.c_int32), ("lstm_layers", C.c_int32), ("kernel_size", C.c_int32),
                ("last_kernel_size", C.c_int32), ("residual_kernel_size", C.c_int32), ("compress", C.c_int32),
                ("max_samples", C.c_int32), ("causal", C.c_int32), ("pad_reflect", C.c_int32), ("conv_shortcut", C.c_int32),
                ("num_residual_layers", C.c_int32), ("dilation_growth_rate", C.c_int32), ("max_batch", C.c_int32)]


PROTOTYPES = {
    "vc_codec_create": (C.c_int, [C.POINTER(CodecCfg), C.c_int, C.POINTER(C.c_void_p)]),
    "vc_codec_destroy": (None, [C.c_void_p]),
    "vc_codec_last_error": (C.c_char_p, [C.c_void_p]),
    "vc_codec_load_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.c_int]),
    "vc_codec_finalize": (C.c_int, [C.c_void_p]),
    "vc_codec_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p]),
    "vc_codec_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "vc_codec_encode_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p]),
    "vc_codec_decode_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "vc_codec_debug_latent": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "vc_codec_last_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "vc_codec_last_lstm_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_double)]),
}

# the VoiceCraft codec (README.md:198 of the reference; config.py:51; phonemize_encodec_encode_hf.py:11-13)
# The last five are the architecture switches the reference tree does not pin; names and defaults as
# in transformers.EncodecConfig at the VoiceCraft codec shape.
DEFAULT_CFG = dict(sample_rate=16000, n_filters=64, ratios=[8, 5, 4, 2], hidden=128, n_q=4, codebook_size=2048,
                   lstm_layers=2, kernel_size=7, last_kernel_size=7, residual_kernel_size=3, compress=2,
                   use_causal_conv=False, pad_mode="reflect", use_conv_shortcut=False, num_residual_layers=1,
                   dilation_growth_rate=2)


def _bind(lib):
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


def fold_weight_norm(g: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """w = g * v / ||v||, the norm taken over every dim but 0 (torch.nn.utils.parametrizations.weight_norm, dim=0)."""
    norm = v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, *([1] * (v.dim() - 1)))
    return v * (g / norm)


def normalize_state_dict(sd: dict[str, torch.Tensor]) -> dict[str, torch.Tensor]:
    """-> {canonical key: fp32 tensor}, canonical = transformers module path with folded `.weight`."""
    out: dict[str, torch.Tensor] = {}
    pend: dict[str, dict[str, torch.Tensor]] = {}
    for k, t in sd.items():
        if not torch.is_tensor(t) or not t.is_floating_point():
            continue
        k2 = k
        # audiocraft -> transformers naming (best effort)
        k2 = re.sub(r"^(encoder|decoder)\.model\.", r"\1.layers.", k2)
        k2 = k2.replace(".convtr.convtr.", ".conv.").replace(".conv.conv.", ".conv.")
        k2 = re.sub(r"^quantizer\.vq\.layers\.(\d+)\._codebook\.", r"quantizer.layers.\1.codebook.", k2)
        m = re.match(r"(.*\.conv)\.(?:parametrizations\.weight\.original([01])|weight_([gv]))$", k2)
        if m:
            which = "g" if (m.group(2) == "0" or m.group(3) == "g") else "v"
            pend.setdefault(m.group(1), {})[which] = t.detach().float()
            continue
        out[k2] = t.detach().float()
    for base, gv in pend.items():
        out[base + ".weight"] = fold_weight_norm(gv["g"], gv["v"])
    return out


def expected_keys(cf: dict) -> set[str]:
    """Canonical tensor names of the configured architecture (module indices as transformers.EncodecEncoder /
    EncodecDecoder enumerate them)."""
    keys: set[str] = set()
    R, NR = len(cf["ratios"]), int(cf["num_residual_layers"])

    def conv(prefix):
        keys.update({prefix + ".conv.weight", prefix + ".conv.bias"})

    def unit(prefix):
        conv(prefix + ".block.1"); conv(prefix + ".block.3")
        if cf["use_conv_shortcut"]:
            conv(prefix + ".shortcut")

    def lstm(prefix):
        for n in range(int(cf["lstm_layers"])):
            keys.update({f"{prefix}.lstm.{w}_l{n}" for w in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")})

    conv("encoder.layers.0")
    idx = 1
    for _ in range(R):
        for _ in range(NR):
            unit(f"encoder.layers.{idx}"); idx += 1
        idx += 1
        conv(f"encoder.layers.{idx}"); idx += 1
    lstm(f"encoder.layers.{idx}"); idx += 2
    conv(f"encoder.layers.{idx}")
    conv("decoder.layers.0"); lstm("decoder.layers.1")
    idx = 2
    for _ in range(R):
        idx += 1
        conv(f"decoder.layers.{idx}"); idx += 1
        for _ in range(NR):
            unit(f"decoder.layers.{idx}"); idx += 1
    idx += 1
    conv(f"decoder.layers.{idx}")
    keys.update({f"quantizer.layers.{q}.codebook.embed" for q in range(int(cf["n_q"]))})
    return keys


# ---------------------------------------------------------------------- dataset on-disk format + bulk encode
def write_codes_txt(codes, path: str) -> None:
    """K lines of space-separated ints, no newline after the last one - the file the reference's dataset encoder
    writes per utterance (write_array_to_txt_file, data/phonemize_encodec_encode_hf.py:50-54)."""
    rows = codes.tolist() if hasattr(codes, "tolist") else [list(r) for r in codes]
    with open(path, "w") as f:
        for a in rows[:-1]:
            f.write(" ".join(map(str, a)) + "\n")
        f.write(" ".join(map(str, rows[-1])))


def read_codes_txt(path: str, n_codebooks: int, special_first: int = 0, n_special: int = 0):
    """The reader of that file as the reference's dataset does it (data/gigaspeech.py:41-62): the first
    n_codebooks lines, ints, shifted by n_special when special_first."""
    with open(path, "r") as e:
        encos = [l.strip().split() for k, l in enumerate(e.readlines()) if k < n_codebooks]
    assert len(encos) == n_codebooks, path
    add = int(n_special) if special_first else 0
    return [[int(n) + add for n in l] for l in encos]


@torch.no_grad()
def bulk_encode(tokenizer, wavs, batch_size: int = 8, max_len: int | None = None, code_sr: int = 50):
    """Dataset encode as data/phonemize_encodec_encode_hf.py:40-46,186-206 does it: clips sorted longest first, taken
    batch_size at a time, zero-padded to the batch's longest (pad_sequence), ONE encode per batch - split in two halves
    when the batch's longest clip exceeds max_len samples and it holds more than one clip - and every clip's codes cut to
    round(seconds * code_sr) frames.  wavs: list of 1-D fp32 tensors; returns the list of int64 [K, T_i] in input order."""
    sr = tokenizer.sample_rate
    lens = [int(w.numel()) for w in wavs]
    order = sorted(range(len(wavs)), key=lambda i: lens[i])[::-1]            # np.argsort(lens)[::-1]
    out = [None] * len(wavs)
    for b0 in range(0, len(order), batch_size):
        ids = order[b0: b0 + batch_size]
        padded = torch.nn.utils.rnn.pad_sequence([wavs[i].reshape(-1).float() for i in ids], batch_first=True).unsqueeze(1)
        if max_len is not None and max(lens[i] for i in ids) > max_len and len(ids) > 1:
            half = len(padded) // 2
            codes = torch.cat([tokenizer.encode(padded[:half])[0][0], tokenizer.encode(padded[half:])[0][0]], dim=0)
        else:
            codes = tokenizer.encode(padded)[0][0]
        for j, i in enumerate(ids):
            actual = round(lens[i] / sr * code_sr)
            out[i] = codes[j, :, :actual].cpu()
    return out


class AudioTokenizer:
    """EnCodec audio on the HIP engine.  `state_dict`: codec weights (see module docstring)."""

    def __init__(self, state_dict: dict[str, torch.Tensor], device="cuda:0", max_seconds: float = 20.0, cfg: dict | None = None,
                 max_batch: int = 8):
        self.lib = _bind(_lib.load())
        cf = dict(DEFAULT_CFG, **(cfg or {}))
        assert cf["pad_mode"] in ("reflect", "constant"), cf["pad_mode"]
        self.max_batch = int(max_batch)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("AudioTokenizer runs on an MI355X only (device must be cuda:N); there is no CPU path")
        index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", index)
        self.sample_rate, self.channels = cf["sample_rate"], 1
        self.n_q = cf["n_q"]
        self.hop = 1
        for r in cf["ratios"]:
            self.hop *= r
        self.max_samples = int(max_seconds * self.sample_rate)
        c = CodecCfg(sample_rate=cf["sample_rate"], n_filters=cf["n_filters"], n_ratios=len(cf["ratios"]), hidden=cf["hidden"],
                     n_q=cf["n_q"], codebook_size=cf["codebook_size"], lstm_layers=cf["lstm_layers"], kernel_size=cf["kernel_size"],
                     last_kernel_size=cf["last_kernel_size"], residual_kernel_size=cf["residual_kernel_size"],
                     compress=cf["compress"], max_samples=self.max_samples, causal=int(bool(cf["use_causal_conv"])),
                     pad_reflect=int(cf["pad_mode"] == "reflect"), conv_shortcut=int(bool(cf["use_conv_shortcut"])),
                     num_residual_layers=int(cf["num_residual_layers"]), dilation_growth_rate=int(cf["dilation_growth_rate"]),
                     max_batch=self.max_batch)
        for i, r in enumerate(cf["ratios"]):
            c.ratios[i] = r
        self._h = C.c_void_p()
        self._check(self.lib.vc_codec_create(C.byref(c), index, C.byref(self._h)), "vc_codec_create", None)
        want = expected_keys(cf)
        have = {k: t for k, t in normalize_state_dict(state_dict).items()
                if not k.endswith((".inited", ".cluster_size", ".embed_avg", ".stride", ".kernel_size", ".padding_total"))}
        # strict coverage: a wrong architecture switch or a mis-mapped audiocraft key must fail here, not decode noise
        missing, unexpected = sorted(want - set(have)), sorted(set(have) - want)
        if missing or unexpected:
            raise AssertionError(f"codec state_dict does not match the configured architecture: missing {missing[:6]}"
                                 f"{'...' if len(missing) > 6 else ''}, unexpected {unexpected[:6]}{'...' if len(unexpected) > 6 else ''}")
        for key, t in have.items():
            t = t.contiguous()
            shape = (C.c_int64 * t.dim())(*t.shape)
            self._check(self.lib.vc_codec_load_tensor(self._h, key.encode(), C.c_void_p(t.data_ptr()), int(t.is_cuda), shape, t.dim()),
                        f"vc_codec_load_tensor({key})")
        self._check(self.lib.vc_codec_finalize(self._h), "vc_codec_finalize")

    def _check(self, rc, what, handle="self"):
        if rc == 0:
            return
        msg = self.lib.vc_codec_last_error(self._h if handle == "self" else None)
        text = msg.decode() if msg else ""
        if rc == -1:
            raise AssertionError(f"{what}: {text}")
        raise _lib.EngineError(f"{what} failed (code {rc}): {text}")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self.lib.vc_codec_destroy(h)
            except Exception:  # pragma: no cover
                pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @torch.no_grad()
    def encode(self, wav: torch.Tensor):
        """wav fp32 [B,1,N] -> [(codes int64 [B,K,T], None)]   (data/tokenizer.py:127-129).  The B clips (a padded batch,
        as data/phonemize_encodec_encode_hf.py:186-198 feeds the reference) go through the stacks together."""
        assert wav.ndim == 3 and wav.shape[1] == self.channels, wav.shape
        B, n = int(wav.shape[0]), int(wav.shape[2])
        cap = (n + self.hop - 1) // self.hop
        outs = []
        for b0 in range(0, B, self.max_batch):
            w = wav[b0: b0 + self.max_batch, 0].to(self.device, torch.float32).contiguous()
            nb = int(w.shape[0])
            codes = torch.empty((nb, self.n_q, cap), dtype=torch.int64, device=self.device)
            T = C.c_int(0)
            self._check(self.lib.vc_codec_encode_batch(self._h, C.c_void_p(w.data_ptr()), nb, n, C.c_void_p(codes.data_ptr()), cap,
                                                       C.byref(T), self._stream()), "vc_codec_encode_batch")
            outs.append(codes[:, :, : T.value])
        return [(torch.cat(outs, dim=0), None)]

    @torch.no_grad()
    def decode(self, frames):
        """frames = [(codes int64 [B,K,T], None)] -> wav fp32 [B,1,hop*T]   (data/tokenizer.py:131-133)"""
        codes = frames[0][0]
        assert codes.ndim == 3 and codes.shape[1] == self.n_q, codes.shape
        B, T = int(codes.shape[0]), int(codes.shape[2])
        outs = []
        for b0 in range(0, B, self.max_batch):
            cd = codes[b0: b0 + self.max_batch].to(self.device, torch.int64).contiguous()
            nb = int(cd.shape[0])
            wav = torch.empty((nb, T * self.hop), dtype=torch.float32, device=self.device)
            self._check(self.lib.vc_codec_decode_batch(self._h, C.c_void_p(cd.data_ptr()), nb, T, C.c_void_p(wav.data_ptr()),
                                                       T * self.hop, self._stream()), "vc_codec_decode_batch")
            outs.append(wav)
        return torch.cat(outs, dim=0).unsqueeze(1)

    def last_latent(self, T: int, hidden: int = 128) -> torch.Tensor:
        out = torch.empty((T, hidden), dtype=torch.float32)
        self._check(self.lib.vc_codec_debug_latent(self._h, C.c_void_p(out.data_ptr()), out.numel()), "vc_codec_debug_latent")
        return out

    def last_lstm_ms(self):
        ms, by = C.c_float(0), C.c_double(0)
        self.lib.vc_codec_last_lstm_ms(self._h, C.byref(ms), C.byref(by))
        return ms.value, by.value

    def last_ms(self) -> float:
        ms = C.c_float(0)
        self.lib.vc_codec_last_ms(self._h, C.byref(ms))
        return ms.value
