"""Python mirror of the reference's model interface for the decode path.

`VoiceCraftEngine` exposes `inference_tts`, `inference_tts_batch` and `inference` with the
signatures, argument meaning, return shapes and AssertionErrors of `models.voicecraft.VoiceCraft`
(models/voicecraft.py:908, :1156, :561), so `inference_tts_scale.inference_one_sample`
(inference_tts_scale.py:42-105) and its editing twin run unchanged on it.  All compute happens in
libvcengine.so (HIP, gfx950) behind the C ABI of include/vc_engine.h; torch is used only to hold
device memory for inputs/outputs, to read the state_dict and to name the current stream.
There is no CPU or eager fallback: without the library every constructor raises.
"""
from __future__ import annotations

import ast
import ctypes as C
import logging
import math
import random
from argparse import Namespace
from typing import Iterable

import torch

from . import _lib
from ._lib import EngineError, ModelCfg, SampleCfg, check

_DTYPES = {"bf16": _lib.VC_DTYPE_BF16, "bfloat16": _lib.VC_DTYPE_BF16, torch.bfloat16: _lib.VC_DTYPE_BF16,
           "fp32": _lib.VC_DTYPE_F32, "float32": _lib.VC_DTYPE_F32, torch.float32: _lib.VC_DTYPE_F32}


def _sine_table(n: int, d: int) -> torch.Tensor:
    # same torch ops as SinePositionalEmbedding.extend_pe (models/modules/embedding.py:77-90)
    pos = torch.arange(0, n, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    pe = torch.zeros(n, d)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe.contiguous()


class VoiceCraftEngine:
    """Drop-in for `VoiceCraft(args)` + `load_state_dict` + `.to(device).eval()` on the inference path.

    Capacities (the reference grows its tensors as it goes; the engine preallocates):
      max_seqs       concurrent sequences (best-of-N samples / utterances of inference_tts_multi)
      max_positions  cached positions per sequence = Lx + prompt columns + generated steps.  A call whose
                     PROMPT does not fit raises EngineError(VC_ECAP) at once; a call whose worst case (the
                     reference's length cap, 10 frames per phoneme) does not fit still runs and raises only
                     if generation really reaches the end of the cache before the terminator.  The default
                     covers Lx <= 370 phonemes in the worst case (11*Lx + 6 positions).
    """

    def __init__(self, args: Namespace | dict, state_dict: dict[str, torch.Tensor], device="cuda:0",
                 dtype="bf16", max_seqs: int = 8, max_positions: int = 4096, use_graph: bool = True):
        self.lib = _lib.load()
        a = Namespace(**args) if isinstance(args, dict) else Namespace(**vars(args))
        # the same normalisation VoiceCraft.__init__ applies (models/voicecraft.py:117-127)
        if not getattr(a, "special_first", False):
            a.special_first = 0
        if not getattr(a, "n_special", False):
            a.n_special = 3
        a.eos = getattr(a, "eos", -1)
        if isinstance(a.audio_vocab_size, str):
            a.audio_vocab_size = int(ast.literal_eval(a.audio_vocab_size))   # the reference eval()s it (voicecraft.py:126-127)
        assert a.text_pad_token == a.text_vocab_size, (a.text_vocab_size, a.text_pad_token)
        assert a.audio_vocab_size == a.empty_token, a.empty_token
        assert a.eog == a.audio_vocab_size + 1, a.eog
        assert a.audio_pad_token == a.audio_vocab_size + 2, a.audio_pad_token
        if a.eos > 0:
            assert a.eos != a.audio_pad_token and a.eos != a.empty_token, a.eos
        self.args = a
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("VoiceCraftEngine runs on an MI355X only (device must be cuda:N); there is no CPU path")
        self.compute_dtype = _DTYPES[dtype]
        self.use_graph = bool(use_graph)
        self.max_seqs, self.max_positions = int(max_seqs), int(max_positions)
        cfg = ModelCfg(
            d_model=a.d_model, nhead=a.nhead, num_layers=a.num_decoder_layers, n_codebooks=a.n_codebooks,
            audio_vocab_size=a.audio_vocab_size, n_special=int(a.n_special), text_rows=a.text_vocab_size + 1,
            head_hidden=a.audio_vocab_size // 2, empty_token=a.empty_token, eog=a.eog,
            audio_pad_token=a.audio_pad_token, eos=a.eos if a.eos > 0 else -1,
            reduced_eog=int(getattr(a, "reduced_eog", 0) or 0), encodec_sr=int(a.encodec_sr),
            max_n_spans=int(a.max_n_spans), max_seqs=self.max_seqs, max_positions=self.max_positions)
        self._h = C.c_void_p()
        index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", index)
        check(self.lib.vc_create(C.byref(cfg), index, C.byref(self._h)), None, "vc_create")
        self._load(state_dict)
        self.last_steps = 0

    # ------------------------------------------------------------------ nn.Module look-alikes
    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self.lib.vc_destroy(h)
            except Exception:  # pragma: no cover
                pass

    # ------------------------------------------------------------------ weights
    def _load(self, state_dict) -> None:
        keep = []
        for key, t in state_dict.items():
            if not torch.is_tensor(t) or not t.is_floating_point():
                continue                      # eog / eos buffers, torchmetrics state
            if key.startswith("accuracy_metrics"):
                continue
            t = t.detach().to(torch.float32).contiguous()
            keep.append(t)
            shape = (C.c_int64 * max(1, t.dim()))(*t.shape)
            check(self.lib.vc_load_tensor(self._h, key.encode(), C.c_void_p(t.data_ptr()), int(t.is_cuda),
                                          _lib.VC_DTYPE_F32, shape, t.dim()), self._h, f"vc_load_tensor({key})")
        pe = _sine_table(self.max_positions, self.args.d_model)
        shape = (C.c_int64 * 2)(*pe.shape)
        check(self.lib.vc_load_tensor(self._h, b"pe", C.c_void_p(pe.data_ptr()), 0, _lib.VC_DTYPE_F32, shape, 2),
              self._h, "vc_load_tensor(pe)")
        check(self.lib.vc_finalize_weights(self._h, self.compute_dtype), self._h, "vc_finalize_weights")

    # ------------------------------------------------------------------ helpers
    def _stream(self) -> C.c_void_p:
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _sample_cfg(self, top_k, top_p, temperature, stop_repetition, silence_tokens, seed=None,
                    forced_mode="tokens") -> SampleCfg:
        sc = SampleCfg()
        sc.forced_mode = {"tokens": 0, "draws": 1}[forced_mode]
        sc.top_k = int(top_k)
        sc.top_p = float(top_p)
        sc.temperature = float(temperature)
        sc.stop_repetition = int(stop_repetition)
        sil = list(silence_tokens)
        assert len(sil) <= _lib.VC_MAX_SILENCE, f"at most {_lib.VC_MAX_SILENCE} silence tokens are supported, got {len(sil)}"
        sc.n_silence = len(sil)
        for i, v in enumerate(sil):
            sc.silence_tokens[i] = int(v)
        # the reference draws from torch's global generator (seed_everything, inference_tts_scale.py:128-135);
        # one draw from it keys the device Philox stream, so torch.manual_seed still makes runs repeatable
        sc.seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if seed is None else int(seed)
        sc.use_graph = int(self.use_graph)
        sc.poll_every = 0
        return sc

    def _prep(self, x, x_lens, y):
        assert x.ndim == 2, x.shape
        assert x_lens.ndim == 1, x_lens.shape
        assert y.ndim == 3, y.shape
        if self.args.special_first:
            y = y + int(self.args.n_special)
        assert y.shape[0] == 1 and y.shape[2] == self.args.n_codebooks, y.shape
        assert x.shape[0] == 1, x.shape
        Lx = int(x_lens[0])
        xd = x[0, :Lx].to(self.device, torch.int64).contiguous()
        yd = y[0].to(self.device, torch.int64).contiguous()          # [T,K], time-major as given
        return xd, Lx, yd, int(yd.shape[0])

    def _forced_arg(self, forced, B: int):
        """[steps,K] or [steps,B,K] int64 -> (device tensor kept alive by the caller, pointer, n_steps)."""
        if forced is None:
            return None, None, 0
        K = self.args.n_codebooks
        fd = torch.as_tensor(forced, dtype=torch.int64).reshape(-1, B, K).to(self.device).contiguous()
        return fd, C.c_void_p(fd.data_ptr()), int(fd.shape[0])

    def _gen_budget(self, Lx: int, n_cols: int, mult: int, spans: int = 1) -> int:
        K = self.args.n_codebooks
        return max(0, Lx * mult - n_cols + 1) + spans * (K + 4) + 8

    # ------------------------------------------------------------------ TTS
    @torch.no_grad()
    def inference_tts(self, x, x_lens, y, top_k: int = -100, top_p: float = 1.0, temperature: float = 1.0,
                      stop_repetition: int = 3, kvcache: int = 1, silence_tokens: Iterable[int] = (1388, 1898, 131),
                      *kargs, _n_samples: int = 1, _forced=None, _logit_steps: int = 0, _seed=None,
                      _forced_mode: str = "tokens"):
        """models/voicecraft.py:908.  `kvcache` is accepted and ignored: the cache is always on
        (kvcache=0 and kvcache=1 give identical tokens in the reference)."""
        xd, Lx, yd, T = self._prep(x, x_lens, y)
        logging.info(f"silence tokens: {list(silence_tokens)}, note that if you are not using the pretrained encodec 6f79c6a8, make sure you specified it yourself, rather than using the default")
        K = self.args.n_codebooks
        sc = self._sample_cfg(top_k, top_p, temperature, stop_repetition, silence_tokens, _seed, _forced_mode)
        cap = T + self._gen_budget(Lx, T + 1, self.args.encodec_sr // 5)
        res = torch.empty((K, cap), dtype=torch.int64, device=self.device)
        nb = int(_n_samples)
        fd, forced_ptr, n_forced = self._forced_arg(_forced, nb)
        logits = None
        if _logit_steps > 0:
            V = self.args.audio_vocab_size + int(self.args.n_special)
            logits = torch.zeros((_logit_steps, nb, K, V), dtype=torch.float32, device=self.device)
        gen_len, n_steps = C.c_int(0), C.c_int(0)
        rc = self.lib.vc_tts(self._h, C.c_void_p(xd.data_ptr()), Lx, C.c_void_p(yd.data_ptr()), T, C.byref(sc),
                             int(_n_samples), forced_ptr, n_forced, C.c_void_p(res.data_ptr()), cap, C.byref(gen_len),
                             C.c_void_p(logits.data_ptr()) if logits is not None else None, int(_logit_steps),
                             C.byref(n_steps), self._stream())
        check(rc, self._h, "vc_tts")
        self.last_steps = n_steps.value
        Tg = gen_len.value
        out = res[:, : T + Tg].unsqueeze(0)
        gen = res[:, T: T + Tg].unsqueeze(0)
        expected_y_len = T + Tg
        assert out.shape == torch.Size((1, K, expected_y_len)), out.shape
        if self.args.special_first:
            out, gen = out - int(self.args.n_special), gen - int(self.args.n_special)
        if logits is not None:
            return out, gen, (logits[:, 0] if nb == 1 else logits)
        return out, gen

    @torch.no_grad()
    def inference_tts_batch(self, x, x_lens, y, top_k: int = -100, top_p: float = 1.0, temperature: float = 1.0,
                            stop_repetition: int = 3, kvcache: int = 1, batch_size: int = 5,
                            silence_tokens: Iterable[int] = (1388, 1898, 131), *kargs, _seed=None, _forced=None,
                            _forced_mode: str = "tokens", _logit_steps: int = 0):
        """models/voicecraft.py:1156 — best-of-N sampling of ONE utterance; returns the kept sample."""
        return self.inference_tts(x, x_lens, y, top_k, top_p, temperature, stop_repetition, kvcache,
                                  silence_tokens, _n_samples=int(batch_size), _seed=_seed, _forced=_forced,
                                  _forced_mode=_forced_mode, _logit_steps=_logit_steps)

    @torch.no_grad()
    def inference_tts_multi(self, xs, ys, top_k: int = -100, top_p: float = 1.0, temperature: float = 1.0,
                            stop_repetition: int = 3, silence_tokens: Iterable[int] = (1388, 1898, 131), _seed=None,
                            _forced=None, _forced_mode: str = "tokens", _logit_steps: int = 0, _shared_text_prefix: int = 0):
        """B different utterances as one batch (not in the reference: SURVEY.md §8f-1).
        xs: list of int64 [Lx_i]; ys: list of int64 [T_i,K].  Returns list of (res [1,K,T_i+Tg_i], gen)
        (+ the raw head logits [steps,B,K,V] as a third value when _logit_steps > 0)."""
        B = len(xs)
        assert B == len(ys) and 1 <= B <= self.max_seqs, (B, self.max_seqs)
        K = self.args.n_codebooks
        xcat = torch.cat([torch.as_tensor(v, dtype=torch.int64).reshape(-1) for v in xs]).to(self.device).contiguous()
        ycat = torch.cat([torch.as_tensor(v, dtype=torch.int64).reshape(-1, K) for v in ys]).to(self.device).contiguous()
        if self.args.special_first:
            ycat = ycat + int(self.args.n_special)
        xo, yo = [0], [0]
        cap = 0
        for xv, yv in zip(xs, ys):
            Lx, T = int(torch.as_tensor(xv).numel()), int(torch.as_tensor(yv).reshape(-1, K).shape[0])
            xo.append(xo[-1] + Lx)
            yo.append(yo[-1] + T)
            cap = max(cap, T + self._gen_budget(Lx, T + 1, self.args.encodec_sr // 5))
        x_off = (C.c_int32 * (B + 1))(*xo)
        y_off = (C.c_int32 * (B + 1))(*yo)
        sc = self._sample_cfg(top_k, top_p, temperature, stop_repetition, silence_tokens, _seed, _forced_mode)
        res = torch.empty((B, K, cap), dtype=torch.int64, device=self.device)
        gen_len = (C.c_int * B)()
        n_steps = C.c_int(0)
        fd, forced_ptr, n_forced = self._forced_arg(_forced, B)
        logits = None
        if _logit_steps > 0:
            V = self.args.audio_vocab_size + int(self.args.n_special)
            logits = torch.zeros((_logit_steps, B, K, V), dtype=torch.float32, device=self.device)
        rc = self.lib.vc_tts_multi(self._h, B, C.c_void_p(xcat.data_ptr()), x_off, C.c_void_p(ycat.data_ptr()), y_off,
                                   C.byref(sc), int(_shared_text_prefix), forced_ptr, n_forced, C.c_void_p(res.data_ptr()), cap, gen_len,
                                   C.c_void_p(logits.data_ptr()) if logits is not None else None, int(_logit_steps),
                                   C.byref(n_steps), self._stream())
        check(rc, self._h, "vc_tts_multi")
        self.last_steps = n_steps.value
        outs = []
        for b in range(B):
            T, Tg = yo[b + 1] - yo[b], gen_len[b]
            r, g = res[b, :, : T + Tg].unsqueeze(0), res[b, :, T: T + Tg].unsqueeze(0)
            if self.args.special_first:
                r, g = r - int(self.args.n_special), g - int(self.args.n_special)
            outs.append((r, g))
        if logits is not None:
            return outs, logits
        return outs

    # ---- the training objective, teacher-forced
    @torch.no_grad()
    def forward(self, batch, mask_intervals=None, mask_values=None, _per_row: bool = False, mask_sampler=None):
        """`VoiceCraft.forward` (models/voicecraft.py:472-559) as an evaluation pass: batch = {"x" [B,Lx], "x_lens" [B],
        "y" [B,K,T], "y_lens" [B]} exactly as the reference's collate gives it; returns the reference's dict
        (`loss` = sum over codebooks of weight * summed cross-entropy, `top10acc`, `top10acc_by_codebook`,
        `effective_ntoken`).  The reference SAMPLES the masked spans inside (`prepare_mask_intervals`, :198-237 - training
        data augmentation, out of this engine's scope); here they are an argument: `mask_intervals[i]` = the (start, end)
        frame pairs of utterance i.  `mask_values[i]` = the utterance's `emb_inds_use` (default 0..M-1; the reference
        shuffles them when `shuffle_mask_embedding` is set).  `mask_sampler`: a callable `y_lens -> mask_intervals` used when
        `mask_intervals` is None (the reference's own `prepare_mask_intervals` fits).  No gradients: this engine does not train."""
        import ast
        import random
        if mask_intervals is None and mask_sampler is not None:
            # `model(batch)` of the reference draws the spans itself (prepare_mask_intervals, models/voicecraft.py:198-237: random
            # training-time augmentation).  The sampler is not part of this engine, but evaluation code that owns one can hand it in:
            # mask_sampler(y_lens) -> one list of (start, end) frame pairs per utterance, e.g. the bound method
            # `reference_model.prepare_mask_intervals` (same argument, same return value)
            mask_intervals = [[(int(s0), int(e0)) for (s0, e0) in iv] for iv in mask_sampler(batch["y_lens"])]
        if mask_intervals is None:
            raise TypeError("VoiceCraftEngine.forward(batch, mask_intervals=... | mask_sampler=...): the masked spans are an argument here - "
                            "pass one list of (start, end) frame pairs per utterance, or a callable that draws them from y_lens (the "
                            "reference samples them inside prepare_mask_intervals, which this inference engine does not reproduce)")
        x, x_lens, y, y_lens = batch["x"], batch["x_lens"], batch["y"], batch["y_lens"]
        if len(x) == 0:
            return None
        K = self.args.n_codebooks
        assert x.ndim == 2, x.shape
        assert x_lens.ndim == 1, x_lens.shape
        assert y.ndim == 3 and y.shape[1] == K, y.shape
        assert y_lens.ndim == 1, y_lens.shape
        B = int(x.shape[0])
        assert B <= self.max_seqs, (B, self.max_seqs)
        assert mask_intervals is not None and len(mask_intervals) == B, "one list of (start, end) spans per utterance"
        if mask_values is None:
            mask_values = []
            for iv in mask_intervals:
                inds = list(range(self.args.max_n_spans))
                if getattr(self.args, "shuffle_mask_embedding", 0):
                    random.shuffle(inds)                                   # insert_mask, :270-272
                mask_values.append(inds[: len(iv)])
        xs = [x[i, : int(x_lens[i])].to(torch.int64).reshape(-1) for i in range(B)]
        ys = [y[i, :, : int(y_lens[i])].to(torch.int64).transpose(0, 1).reshape(-1, K) for i in range(B)]     # time-major
        xcat = torch.cat(xs).to(self.device).contiguous()
        ycat = torch.cat(ys).to(self.device).contiguous()
        if self.args.special_first:
            ycat = ycat + int(self.args.n_special)
        xo, yo, so = [0], [0], [0]
        flat_iv, flat_mv = [], []
        rows_cap = 0
        for i in range(B):
            xo.append(xo[-1] + int(xs[i].numel()))
            yo.append(yo[-1] + int(ys[i].shape[0]))
            M = len(mask_intervals[i])
            assert M == len(mask_values[i]) and M >= 1, (M, mask_values[i])
            so.append(so[-1] + M)
            for (s0, e0) in mask_intervals[i]:
                flat_iv += [int(s0), int(e0)]
            flat_mv += [int(v) for v in mask_values[i]]
            rows_cap += ((int(xs[i].numel()) + int(ys[i].shape[0]) + (2 * M + 1) * (K + 1) + 2 * M) + 15) // 16 * 16
        c32 = C.c_int32
        nll_sum = (C.c_double * K)()
        hits = (C.c_int64 * K)()
        n_targets = C.c_int64(0)
        n_rows = C.c_int64(0)
        nll = tgt = None
        if _per_row:
            nll = torch.zeros((rows_cap, K), dtype=torch.float32, device=self.device)
            tgt = torch.full((rows_cap, K), -1, dtype=torch.int32, device=self.device)
        rc = self.lib.vc_eval_forward(self._h, B, C.c_void_p(xcat.data_ptr()), (c32 * (B + 1))(*xo),
                                      C.c_void_p(ycat.data_ptr()), (c32 * (B + 1))(*yo),
                                      (c32 * len(flat_iv))(*flat_iv), (c32 * (B + 1))(*so), (c32 * len(flat_mv))(*flat_mv),
                                      nll_sum, hits, C.byref(n_targets),
                                      C.c_void_p(nll.data_ptr()) if nll is not None else None,
                                      C.c_void_p(tgt.data_ptr()) if tgt is not None else None,
                                      rows_cap if _per_row else 0, C.byref(n_rows), self._stream())
        check(rc, self._h, "vc_eval_forward")
        cw = getattr(self.args, "codebook_weight", None)
        cw = [float(w) for w in ast.literal_eval(cw)] if cw else [1.0] * K
        dev = self.device
        by_cb = [torch.tensor(float(hits[k]), device=dev) for k in range(K)]
        out = {
            "loss": torch.tensor(sum(nll_sum[k] * cw[k] for k in range(K)), dtype=torch.float32, device=dev),
            "top10acc": torch.tensor(float(sum(hits[k] for k in range(K))), device=dev),
            "top10acc_by_codebook": by_cb,
            "effective_ntoken": torch.tensor(int(n_targets.value) * K).to(dev),
        }
        if _per_row:
            out["_nll_rows"], out["_tgt_rows"] = nll[: n_rows.value], tgt[: n_rows.value]
            out["_nll_sum"] = [float(nll_sum[k]) for k in range(K)]
        return out

    @torch.no_grad()
    def inference_tts_long(self, x_prompt, x_sentences, y, top_k: int = -100, top_p: float = 1.0, temperature: float = 1.0,
                           stop_repetition: int = 3, silence_tokens: Iterable[int] = (1388, 1898, 131), reuse_prefix: bool = True,
                           _seed=None):
        """Sentence-chained "Long TTS" (gradio_app.py:231-236, :249-313): the reference synthesises every sentence with
        its own `inference_one_sample` call on the text [transcript of the voice prompt ; sentence] and the SAME audio
        prompt.  Here all sentences are decoded together (chunks of max_seqs: the weights are streamed once per step
        for all of them), each row following inference_tts exactly, and with reuse_prefix the K/V of the shared
        transcript prefix are computed once per chunk and read by every sentence (include/vc_engine.h, vc_tts_multi).

        x_prompt int64 [Lp] phonemes of the voice prompt's transcript, x_sentences list of int64 [Ls_i],
        y int64 [1,T,K] (or [T,K]) codes of the voice prompt.  Returns a list of (res [1,K,T+Tg_i], gen [1,K,Tg_i])."""
        xp = torch.as_tensor(x_prompt, dtype=torch.int64).reshape(-1)
        K = self.args.n_codebooks
        yy = torch.as_tensor(y, dtype=torch.int64).reshape(-1, K)
        outs = []
        for c0 in range(0, len(x_sentences), self.max_seqs):
            chunk = x_sentences[c0: c0 + self.max_seqs]
            xs = [torch.cat([xp, torch.as_tensor(v, dtype=torch.int64).reshape(-1)]) for v in chunk]
            for v in xs:
                assert v.numel() > xp.numel(), "every sentence needs at least one phoneme"
            share = int(xp.numel()) if (reuse_prefix and len(chunk) > 1) else 0
            outs += self.inference_tts_multi(xs, [yy] * len(chunk), top_k, top_p, temperature, stop_repetition, silence_tokens,
                                             _seed=None if _seed is None else _seed + c0, _shared_text_prefix=share)
        return outs

    # ------------------------------------------------------------------ editing
    @torch.no_grad()
    def inference(self, x, x_lens, y, mask_interval, top_k: int = -100, top_p: float = 1.0, temperature: float = 1.0,
                  stop_repetition: int = -1, kvcache: int = 1, silence_tokens: Iterable[int] = (1388, 1898, 131),
                  _forced=None, _logit_steps: int = 0, _seed=None, _forced_mode: str = "tokens"):
        """models/voicecraft.py:561."""
        xd, Lx, yd, T = self._prep(x, x_lens, y)
        assert mask_interval.shape == torch.Size((1, mask_interval.shape[1], 2)), mask_interval
        logging.info(f"silence tokens: {list(silence_tokens)}, note that if you are not using the pretrained encodec 6f79c6a8, make sure you specified it yourself, rather than using the default")
        K = self.args.n_codebooks
        ivs = [(int(a), int(b)) for a, b in mask_interval[0].tolist()]
        M = len(ivs)
        # insert_mask (models/voicecraft.py:264-288): which mask_embedding row each placeholder uses
        emb_inds = list(range(int(self.args.max_n_spans)))
        if getattr(self.args, "shuffle_mask_embedding", 0):
            random.shuffle(emb_inds)
        use = emb_inds[:M]
        assert len(use) == M, f"{M} spans but max_n_spans is {self.args.max_n_spans}"
        mask_value = use + use
        # a zero-length piece makes the reference raise inside build_pattern_sequence (codebooks_patterns.py:174)
        starts = [iv[0] for iv in ivs] + [T]
        ends = [0] + [iv[1] for iv in ivs]
        eos, reduced = self.args.eos, int(getattr(self.args, "reduced_eog", 0) or 0)
        for i, (s, e) in enumerate(zip(ends, starts)):
            has_term = (i == M) if (eos > 0 or reduced) else True
            if e - s + int(has_term) <= 0:
                raise IndexError("index is out of bounds for dimension with size 0 (zero-length non-masked piece)")
        sc = self._sample_cfg(top_k, top_p, temperature, stop_repetition, silence_tokens, _seed, _forced_mode)
        flat = [v for iv in ivs for v in iv]
        iv_arr = (C.c_int32 * (2 * M))(*flat)
        mv_arr = (C.c_int32 * (2 * M))(*mask_value)
        n_cols = T + 2 * (M + 1) + (M + 1) * K + 1
        cap = T + self._gen_budget(Lx, n_cols, 10, spans=M)
        res = torch.empty((K, cap), dtype=torch.int64, device=self.device)
        fd, forced_ptr, n_forced = self._forced_arg(_forced, 1)
        logits = None
        if _logit_steps > 0:
            V = self.args.audio_vocab_size + int(self.args.n_special)
            logits = torch.zeros((_logit_steps, K, V), dtype=torch.float32, device=self.device)
        res_len, n_steps = C.c_int(0), C.c_int(0)
        rc = self.lib.vc_edit(self._h, C.c_void_p(xd.data_ptr()), Lx, C.c_void_p(yd.data_ptr()), T, iv_arr, M, mv_arr,
                              C.byref(sc), forced_ptr, n_forced, C.c_void_p(res.data_ptr()), cap, C.byref(res_len),
                              C.c_void_p(logits.data_ptr()) if logits is not None else None, int(_logit_steps),
                              C.byref(n_steps), self._stream())
        check(rc, self._h, "vc_edit")
        self.last_steps = n_steps.value
        out = res[:, : res_len.value].unsqueeze(0)
        if self.args.special_first:
            out = out - int(self.args.n_special)
        if logits is not None:
            return out, logits
        return out

    # ------------------------------------------------------------------ measurement hooks
    def set_option(self, name: str, value) -> None:
        """Run-time launch-shape option of the decode step (include/vc_engine.h vc_set_option), e.g. ("fr_one", "0") or
        ("tile_attn", "2,768").  Exact-mode tokens never depend on one; bf16 logits move by rounding where a form re-orders sums
        (include/vc_engine.h).  bench.py --ab toggles one inside a process."""
        check(self.lib.vc_set_option(self._h, str(name).encode(), str(value).encode()), self._h, f"vc_set_option({name})")

    def options(self) -> str:
        """The engine's option state as text, `key=v,v,...|key=...`: g = steps per graph, ls = ln_split_rows, ab = attention workgroups
        aimed at (several rows, one row), nt = (weight mask, K/V rows), fr = finished-row form (max rows, consumer tiles, split rows,
        paired producer), ta = prefill attention (kernel, min rows), r1 = one-row step (fr_one, ln_trim, attn_fast, qkv_p8), q16 = many-row
        steps (qkv16, wide_heads, mt_tiles, wide_gemm, wd_stage), sh = shrink.  bench.py turns it into a JSON object (`config.engine_options`)."""
        return bytes(self.debug_read("options", (256,), torch.uint8).tolist()).split(b"\0")[0].decode()

    def last_timing_ms(self):
        ms = (C.c_float * 3)()
        check(self.lib.vc_last_timing(self._h, ms), self._h, "vc_last_timing")
        return {"prefill_ms": ms[0], "decode_ms": ms[1], "total_ms": ms[2]}

    def bench_kernel(self, which: str, n_rows: int = 1, iters: int = 50):
        ms, nbytes = C.c_float(0), C.c_double(0)
        check(self.lib.vc_bench_kernel(self._h, which.encode(), n_rows, iters, C.byref(ms), C.byref(nbytes), self._stream()),
              self._h, "vc_bench_kernel")
        return ms.value, nbytes.value

    LAUNCH_FORMS = ("rows_gemm", "mt2", "mt4", "blk64", "blk128_sbs", "blk128_2x2", "blk64_occ2", "ln_rows", "rows_attn",
                    "tile_attn", "rows_gemm_fr", "big256", "big128", "row_gemm_fr1", "tile_attn64", "rows_gemm_frp", "wd", "rows_gemm_qp")

    def launch_counts(self) -> dict:
        """Process-wide census of the kernel FORMS launched so far (vc_common.h VC_LC_*): the parity tests take the
        difference around a call to assert which form a benchmarked shape really runs on."""
        c = self.debug_read("launch_counts", (len(self.LAUNCH_FORMS),), torch.int64)
        return {n: int(c[i]) for i, n in enumerate(self.LAUNCH_FORMS)}

    def debug_read(self, name: str, shape, dtype=torch.float32) -> torch.Tensor:
        out = torch.empty(shape, dtype=dtype)
        check(self.lib.vc_debug_read(self._h, name.encode(), C.c_void_p(out.data_ptr()), out.numel() * out.element_size()),
              self._h, "vc_debug_read")
        return out


def debug_sample(logits: torch.Tensor, n_draws: int, top_k: int = -100, top_p: float = 1.0, temperature: float = 1.0,
                 seed: int = 0) -> torch.Tensor:
    """n_draws independent tokens from ONE fp32 logits row [V] on the GPU through the product sampler
    (vc_debug_sample; tests only): int32 [n_draws]."""
    lib = _lib.load()
    assert logits.is_cuda and logits.dtype == torch.float32 and logits.ndim == 1
    logits = logits.contiguous()
    sc = SampleCfg()
    sc.top_k, sc.top_p, sc.temperature, sc.seed = int(top_k), float(top_p), float(temperature), int(seed)
    out = torch.empty((n_draws,), dtype=torch.int32, device=logits.device)
    rc = lib.vc_debug_sample(C.c_void_p(logits.data_ptr()), int(logits.numel()), C.byref(sc), int(n_draws),
                             C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream(logits.device).cuda_stream))
    if rc != 0:
        raise AssertionError(f"vc_debug_sample rejected its arguments (code {rc})")
    return out


def box_probe(device="cuda:0", hops: int = 256) -> dict:
    """The box's dependent-load latency as ONE workgroup on an idle chip sees it (vc_box_probe): over a 128 KB ring (cache-resident)
    and a 256 MB ring (memory), plus the shader clock that lone workgroup ran at."""
    lib = _lib.load()
    dev = torch.device(device)
    out = {}
    with torch.cuda.device(dev):
        s = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        for name, nbytes in (("l2", 128 << 10), ("hbm", 256 << 20)):
            res = (C.c_float * 2)()
            rc = lib.vc_box_probe(C.c_longlong(nbytes), int(hops), res, s)
            if rc != 0:
                raise EngineError(f"vc_box_probe failed (code {rc})")
            out[f"dependent_load_ns_{name}"] = round(float(res[0]), 1)
            out[f"shader_mhz_{name}_walk"] = round(float(res[1]), 0)
    return out


# ---------------------------------------------------------------------- pattern ops (no engine needed)
def _pattern_call(fn, *args):
    rc = fn(*args)
    if rc != 0:
        raise AssertionError(f"pattern kernel rejected its arguments (code {rc})")


def pattern_shift(z: torch.Tensor, special: int) -> torch.Tensor:
    """z int64 [B,K,T] on the GPU -> [B,K,T+K] (Pattern.build_pattern_sequence values, codebooks_patterns.py:151-176)."""
    lib = _lib.load()
    assert z.is_cuda and z.dtype == torch.int64 and z.ndim == 3
    z = z.contiguous()
    B, K, T = z.shape
    out = torch.empty((B, K, T + K), dtype=torch.int64, device=z.device)
    _pattern_call(lib.vc_pattern_shift, C.c_void_p(z.data_ptr()), B, K, T, int(special), C.c_void_p(out.data_ptr()),
                  C.c_void_p(torch.cuda.current_stream(z.device).cuda_stream))
    return out


def pattern_revert(s: torch.Tensor, T: int, special: int) -> torch.Tensor:
    """s int64 [B,K,S] -> [B,K,T] (Pattern.revert_pattern_sequence values, codebooks_patterns.py:222-245)."""
    lib = _lib.load()
    assert s.is_cuda and s.dtype == torch.int64 and s.ndim == 3
    s = s.contiguous()
    B, K, S = s.shape
    out = torch.empty((B, K, T), dtype=torch.int64, device=s.device)
    _pattern_call(lib.vc_pattern_revert, C.c_void_p(s.data_ptr()), B, K, S, int(T), int(special),
                  C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream(s.device).cuda_stream))
    return out


def pattern_unshift(span: torch.Tensor) -> torch.Tensor:
    """span int64 [N,K] (one row per decode step) -> [K,N-K] (models/voicecraft.py:1125-1139)."""
    lib = _lib.load()
    assert span.is_cuda and span.dtype == torch.int64 and span.ndim == 2
    span = span.contiguous()
    N, K = span.shape
    assert N >= K, f"a span has at least K={K} steps, got {N}"      # voicecraft.py:1137
    out = torch.empty((K, N - K), dtype=torch.int64, device=span.device)
    _pattern_call(lib.vc_pattern_unshift, C.c_void_p(span.data_ptr()), N, K, C.c_void_p(out.data_ptr()),
                  C.c_void_p(torch.cuda.current_stream(span.device).cuda_stream))
    return out
