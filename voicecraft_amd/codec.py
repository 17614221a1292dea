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
                ("max_samples", C.c_int32)]


PROTOTYPES = {
    "vc_codec_create": (C.c_int, [C.POINTER(CodecCfg), C.c_int, C.POINTER(C.c_void_p)]),
    "vc_codec_destroy": (None, [C.c_void_p]),
    "vc_codec_last_error": (C.c_char_p, [C.c_void_p]),
    "vc_codec_load_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.c_int]),
    "vc_codec_finalize": (C.c_int, [C.c_void_p]),
    "vc_codec_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p]),
    "vc_codec_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "vc_codec_debug_latent": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "vc_codec_last_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
}

# the VoiceCraft codec (README.md:198 of the reference; config.py:51; phonemize_encodec_encode_hf.py:11-13)
DEFAULT_CFG = dict(sample_rate=16000, n_filters=64, ratios=[8, 5, 4, 2], hidden=128, n_q=4, codebook_size=2048,
                   lstm_layers=2, kernel_size=7, last_kernel_size=7, residual_kernel_size=3, compress=2)


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


class AudioTokenizer:
    """EnCodec audio on the HIP engine.  `state_dict`: codec weights (see module docstring)."""

    def __init__(self, state_dict: dict[str, torch.Tensor], device="cuda:0", max_seconds: float = 20.0, cfg: dict | None = None):
        self.lib = _bind(_lib.load())
        cf = dict(DEFAULT_CFG, **(cfg or {}))
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
                     compress=cf["compress"], max_samples=self.max_samples)
        for i, r in enumerate(cf["ratios"]):
            c.ratios[i] = r
        self._h = C.c_void_p()
        self._check(self.lib.vc_codec_create(C.byref(c), index, C.byref(self._h)), "vc_codec_create", None)
        for key, t in normalize_state_dict(state_dict).items():
            if key.endswith((".inited", ".cluster_size", ".embed_avg")):
                continue
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
        """wav fp32 [B,1,N] -> [(codes int64 [B,K,T], None)]   (data/tokenizer.py:127-129)"""
        assert wav.ndim == 3 and wav.shape[1] == self.channels, wav.shape
        outs = []
        for b in range(wav.shape[0]):
            w = wav[b, 0].to(self.device, torch.float32).contiguous()
            n = int(w.numel())
            cap = (n + self.hop - 1) // self.hop
            codes = torch.empty((self.n_q, cap), dtype=torch.int64, device=self.device)
            T = C.c_int(0)
            self._check(self.lib.vc_codec_encode(self._h, C.c_void_p(w.data_ptr()), n, C.c_void_p(codes.data_ptr()), cap,
                                                 C.byref(T), self._stream()), "vc_codec_encode")
            outs.append(codes[:, : T.value])
        return [(torch.stack(outs, dim=0), None)]

    @torch.no_grad()
    def decode(self, frames):
        """frames = [(codes int64 [B,K,T], None)] -> wav fp32 [B,1,hop*T]   (data/tokenizer.py:131-133)"""
        codes = frames[0][0]
        assert codes.ndim == 3 and codes.shape[1] == self.n_q, codes.shape
        outs = []
        for b in range(codes.shape[0]):
            cd = codes[b].to(self.device, torch.int64).contiguous()
            T = int(cd.shape[1])
            wav = torch.empty((T * self.hop,), dtype=torch.float32, device=self.device)
            self._check(self.lib.vc_codec_decode(self._h, C.c_void_p(cd.data_ptr()), T, C.c_void_p(wav.data_ptr()), int(wav.numel()),
                                                 self._stream()), "vc_codec_decode")
            outs.append(wav)
        return torch.stack(outs, dim=0).unsqueeze(1)

    def last_latent(self, T: int, hidden: int = 128) -> torch.Tensor:
        out = torch.empty((T, hidden), dtype=torch.float32)
        self._check(self.lib.vc_codec_debug_latent(self._h, C.c_void_p(out.data_ptr()), out.numel()), "vc_codec_debug_latent")
        return out

    def last_ms(self) -> float:
        ms = C.c_float(0)
        self.lib.vc_codec_last_ms(self._h, C.byref(ms))
        return ms.value
