"""VoiceCraft decode path — CPU oracle (torch fp32).  TEST INFRASTRUCTURE ONLY.

A functional restatement of the reference's inference path over a plain state_dict:

  block / decoder stack   <- models/modules/transformer.py:266-343, :417-488 (pre-LN, ReLU FFN)
  attention with cache    <- models/modules/activation.py:513-652
  embeddings              <- models/modules/embedding.py:22-48, :51-97
  samplers                <- models/voicecraft.py:26-68 (top_k_top_p_filtering), :71-86 (topk_sampling)
  sequence rearrangement  <- models/voicecraft.py:239-320 (rearrange/shift/insert_mask/cat_y/embed_y)
  generation loops        <- models/voicecraft.py:561-906 (inference), :908-1153 (inference_tts),
                             :1156-1439 (inference_tts_batch)
  training objective      <- models/voicecraft.py:472-559 (forward) with the mask intervals given instead of drawn
                             (:198-237), :322-404 (prepare_input_target / remove_mask / revert_pattern),
                             codebooks_patterns.py:178-266 (revert_pattern_logits); top-10 accuracy restates
                             torchmetrics 0.11.1 MulticlassAccuracy(top_k=10, average="micro") (README.md:113)

The three public generation methods of the reference share one loop here (`_run`); the ATen ops,
their order and their operand shapes are kept identical to the reference, including the per-step
`torch.cat` growth of the fp32 KV cache (voicecraft.py:1081), so that (a) greedy trajectories are
bit-identical to the reference on the same CPU (pinned by tests/golden) and (b) timing this oracle
is a fair "port" CPU baseline.  `trace` collects per-step raw logits and emitted tokens for
teacher-forced parity of the HIP engine.
"""
from __future__ import annotations

import math
from argparse import Namespace
from typing import Any

import numpy as np
import torch
import torch.nn.functional as F

from .pattern import delayed_shift, unshift_span


# --------------------------------------------------------------------------- samplers
def filter_top_k_top_p(logits: torch.Tensor, top_k: int = 0, top_p: float = 1.0) -> torch.Tensor:
    """In place, rows = distributions.  Ties at the k-th value survive; the first token whose
    cumulative mass crosses top_p survives (voicecraft.py:38-67)."""
    neg = -float("inf")
    if top_k > 0:
        k = min(max(top_k, 1), logits.size(-1))
        kth = torch.topk(logits, k)[0][..., -1, None]
        logits[logits < kth] = neg
    if top_p < 1.0:
        srt, order = torch.sort(logits, descending=True)
        cum = torch.cumsum(F.softmax(srt, dim=-1), dim=-1)
        drop = cum > top_p
        drop[..., 1:] = drop[..., :-1].clone()
        drop[..., 0] = 0
        logits[drop.scatter(1, order, drop)] = neg
    return logits


def draw(logits: torch.Tensor, top_k: int, top_p: float, temperature: float) -> torch.Tensor:
    """[R,V] -> [R,1] ids (voicecraft.py:71-86).  Division by the temperature makes a copy, exactly
    as in the reference, so the caller's tensor only sees the filter when temperature == 1."""
    if temperature != 1.0:
        logits = logits / temperature
    logits = filter_top_k_top_p(logits, top_k=top_k, top_p=top_p)
    return torch.multinomial(F.softmax(logits, dim=-1), num_samples=1)


# --------------------------------------------------------------------------- model
def sine_table(n: int, d: int) -> torch.Tensor:
    """embedding.py:69-92"""
    pos = torch.arange(0, n, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    pe = torch.zeros(n, d)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe.unsqueeze(0)


class VoiceCraftOracle:
    def __init__(self, args: Namespace | dict, state_dict: dict[str, torch.Tensor]):
        a = dict(vars(args)) if isinstance(args, Namespace) else dict(args)
        self.d = int(a["d_model"]); self.H = int(a["nhead"]); self.L = int(a["num_decoder_layers"])
        self.K = int(a["n_codebooks"])
        av = a["audio_vocab_size"]
        self.audio_vocab = int(eval(av)) if isinstance(av, str) else int(av)     # voicecraft.py:126-127
        self.n_special = int(a.get("n_special") or 3)                           # :119-120
        self.V = self.audio_vocab + self.n_special
        self.empty = int(a["empty_token"]); self.eog = int(a["eog"]); self.pad = int(a["audio_pad_token"])
        self.eos = int(a.get("eos", -1))
        self.reduced_eog = int(a.get("reduced_eog", 0) or 0)
        self.special_first = int(a.get("special_first", 0) or 0)
        self.encodec_sr = int(a["encodec_sr"])
        self.max_n_spans = int(a["max_n_spans"])
        self.sd = {k: v.detach().to(torch.float32) for k, v in state_dict.items() if v.is_floating_point()}
        self.pe = sine_table(4000, self.d)
        self.hd = self.d // self.H

    # ---- small pieces
    def _pos(self, emb: torch.Tensor, which: str) -> torch.Tensor:
        n = emb.size(1)
        if n > self.pe.size(1):
            self.pe = sine_table(n, self.d)
        return emb * 1.0 + self.sd[f"{which}_positional_embedding.alpha"] * self.pe[:, :n]

    def _embed_cols(self, cols: torch.Tensor) -> torch.Tensor:
        """cols [K,S,B] int64 -> [B,S,d]: per-codebook lookup, stack, sum (voicecraft.py:311-316)."""
        e = torch.stack([F.embedding(cols[k], self.sd[f"audio_embedding.{k}.word_embeddings.weight"])
                         for k in range(self.K)], dim=0)
        return e.sum(dim=0).transpose(1, 0)

    def _attn(self, l: int, x: torch.Tensor, mask: torch.Tensor, past_l):
        """x [B,n,d] (already normed), mask float [B,H,n,S]; returns (out [B,n,d], present [2,B,H,n,hd])."""
        p = f"decoder.layers.{l}.self_attn."
        B, n, d = x.shape
        H, hd = self.H, self.hd
        xt = x.transpose(1, 0)                                            # [n,B,d]
        proj = F.linear(xt, self.sd[p + "in_proj_weight"], self.sd[p + "in_proj_bias"])
        proj = proj.unflatten(-1, (3, d)).unsqueeze(0).transpose(0, -2).squeeze(-2).contiguous()
        q, k, v = proj[0], proj[1], proj[2]
        q = q.view(n, B * H, hd).transpose(0, 1).view(B, H, n, hd)
        k = k.view(n, B * H, hd).transpose(0, 1).view(B, H, n, hd)
        v = v.view(n, B * H, hd).transpose(0, 1).view(B, H, n, hd)
        present = torch.stack([k, v], dim=0)
        if past_l is not None:
            k = torch.cat([past_l[0], k], dim=-2)
            v = torch.cat([past_l[1], v], dim=-2)
        o = F.scaled_dot_product_attention(q, k, v, mask, 0.0, is_causal=False)
        o = o.permute(2, 0, 1, 3).contiguous().view(B * n, d)
        o = F.linear(o, self.sd[p + "out_proj.weight"], self.sd[p + "out_proj.bias"]).view(n, B, d)
        return o.transpose(1, 0), present

    def _stack(self, x: torch.Tensor, mask: torch.Tensor, past):
        """Decoder stack + final norm. past: None or [L,2,B,H,S,hd]. Returns (out, present [L,2,B,H,n,hd])."""
        pres = []
        for l in range(self.L):
            p = f"decoder.layers.{l}."
            a, pr = self._attn(l, F.layer_norm(x, (self.d,), self.sd[p + "norm1.weight"], self.sd[p + "norm1.bias"], 1e-5),
                               mask, None if past is None else past[l])
            x = x + a
            h = F.layer_norm(x, (self.d,), self.sd[p + "norm2.weight"], self.sd[p + "norm2.bias"], 1e-5)
            h = F.linear(F.relu(F.linear(h, self.sd[p + "linear1.weight"], self.sd[p + "linear1.bias"])),
                         self.sd[p + "linear2.weight"], self.sd[p + "linear2.bias"])
            x = x + h
            pres.append(pr)
        x = F.layer_norm(x, (self.d,), self.sd["decoder.norm.weight"], self.sd["decoder.norm.bias"], 1e-5)
        return x, torch.stack(pres, dim=0)

    def _heads(self, h_last: torch.Tensor) -> torch.Tensor:
        """h_last [B,1,d] -> [B,K,V] (voicecraft.py:181-185, :1084-1086)."""
        outs = []
        for k in range(self.K):
            p = f"predict_layer.{k}."
            z = F.linear(h_last, self.sd[p + "0.weight"], self.sd[p + "0.bias"])
            z = F.linear(F.gelu(z), self.sd[p + "2.weight"], self.sd[p + "2.bias"])
            outs.append(z)
        return torch.stack(outs, dim=1).squeeze(2)

    def _causal_rows(self, B: int, S: int, n_last: int) -> torch.Tensor:
        """float mask [B,H,n_last,S]: last n_last rows of the lower-triangular mask (voicecraft.py:419-447)."""
        tri = torch.triu(torch.ones(S, S), diagonal=1).bool()
        m = torch.zeros(S, S).masked_fill_(tri, float("-inf"))
        return m[-n_last:].unsqueeze(0).unsqueeze(0).expand(B, self.H, n_last, S).contiguous()

    # ---- the shared generation loop
    def _run(self, x, cols0, mask_cols, *, mode, more_mask, n_spans, B, top_k, top_p, temperature,
             stop_repetition, kvcache, silence_tokens, trace, forced=None, max_steps=None, forced_draws=None):
        """x [1,Lx]; cols0 [K,S0] prompt columns; mask_cols {col: mask_embedding row}.
        forced [steps][K]: the step's FINAL tokens are replaced (teacher forcing for logits parity);
        forced_draws [steps][B][K]: the raw draws of topk_sampling are replaced and the state machine
        (overrides, termination, keep) runs on them - the replay of a recorded reference run.
        Returns (spans: list of [N,K] int arrays per finished span, kept sample index)."""
        K, V = self.K, self.V
        tts = mode == "tts"
        term = (self.eos if self.eos > 0 else self.eog) if tts else self.eog
        kill = (self.eog if self.eos > 0 else -1) if tts else (self.eos if self.eos > 0 else -1)
        cap_mult = (self.encodec_sr // 5) if tts else 10
        Lx = x.size(1)
        x_in = self._pos(F.embedding(x, self.sd["text_embedding.word_embeddings.weight"]), "text")
        y_emb = self._embed_cols(cols0.unsqueeze(-1))                               # [1,S0,d]
        if mask_cols:
            pos = sorted(mask_cols)
            y_emb[0, pos] = self.sd["mask_embedding"][[mask_cols[c] for c in pos]]
        grouped = B > 1
        if grouped:
            x_in = x_in.repeat(B, 1, 1)
            y_emb = y_emb.repeat(B, 1, 1)
        past = None
        n_new = None                         # rows to feed when the cache is warm
        cb_eog = [False] * K
        cur, spans, cur_num_gen = [[] for _ in range(B)], [], 0
        prev = [None] * B
        consec = [0] * B
        keep = None if grouped else 0
        more_mask = list(more_mask)
        step = 0
        while True:
            y_in = self._pos(y_emb, "audio")
            S = Lx + y_in.size(1)
            if kvcache and past is not None:
                xy = y_in[:, -n_new:]
                out, present = self._stack(xy, self._causal_rows(B, S, n_new), past)
                past = torch.cat([past, present.to(past.dtype)], dim=-2)
            else:
                xy = torch.cat([x_in, y_in], dim=1)
                out, present = self._stack(xy, self._causal_rows(B, S, S), None)
                out = out[:, Lx:]
                if kvcache:
                    past = present.to(torch.float32)
            logits = self._heads(out[:, -1:])                                        # [B,K,V]
            if trace is not None:
                trace.append({"logits": logits.clone()})
            n_eog = sum(cb_eog)
            if kill >= 0:
                logits[:, :, kill] = -10000.0
            y_len = y_in.size(1)
            # ---- logit edits + draw (sample_helper, voicecraft.py:1018-1067 / :718-787 / :1269-1325)
            first_cb = 1 if n_eog == 0 else n_eog + 1
            for jj in range(first_cb, K):
                logits[:, jj, term] = -10000.0
                logits[:, jj, self.empty] = -10000.0
            if n_eog == 0:
                if tts and cur_num_gen <= self.encodec_sr // 5:
                    logits[:, 0, term] = -10000.0
                for b in range(B):
                    if stop_repetition > 0 and prev[b] in silence_tokens and consec[b] > stop_repetition:
                        f = consec[b] - (stop_repetition - 1)
                        if logits[b, 0, prev[b]] < 0:
                            logits[b, 0, prev[b]] = logits[b, 0, prev[b]] * f
                        else:
                            logits[b, 0, prev[b]] = logits[b, 0, prev[b]] / f
            flat = logits.reshape(B * K, V) if grouped else logits[0]
            samples = draw(flat, top_k, top_p, temperature).reshape(B, K)
            if forced_draws is not None:
                samples = torch.as_tensor(np.asarray(forced_draws[step]), dtype=samples.dtype).reshape(B, K).clone()
            if n_eog == 0:
                for b in range(B):
                    for jj in range(1, K - cur_num_gen):
                        samples[b, K - jj] = self.empty
                    hit = (samples[b, 0] == term or torch.argmax(logits[b, 0], dim=-1) == term
                           or y_len > Lx * cap_mult)
                    if forced is not None:
                        samples[b] = torch.as_tensor(forced[step], dtype=samples.dtype)
                        hit = bool(samples[b, 0] == term)
                    if hit:
                        samples[b, 0] = term
                        cb_eog[0] = True
                        keep = b
                    t0 = int(samples[b, 0])
                    consec[b] = consec[b] + 1 if (t0 in silence_tokens and t0 == prev[b]) else 0
                    prev[b] = t0
            else:
                samples[keep, :n_eog] = self.empty
                samples[keep, n_eog] = term
                if forced is not None:
                    samples[keep] = torch.as_tensor(forced[step], dtype=samples.dtype)
                cb_eog[n_eog] = True
            cur_num_gen += 1
            step += 1
            if max_steps is not None and step >= max_steps:      # bounded sample for CPU timing only
                return None, step
            if sum(cb_eog) == 0:
                for b in range(B):
                    cur[b].append(samples[b].clone())
            else:
                if isinstance(cur[0], list):
                    cur = cur[keep]
                cur.append(samples[keep].clone())
            if trace is not None:
                trace[-1]["tokens"] = samples[keep if keep is not None else 0].clone()
            s_emb = torch.stack([F.embedding(samples[:, k:k + 1], self.sd[f"audio_embedding.{k}.word_embeddings.weight"])
                                 for k in range(K)], dim=1).sum(dim=1)                 # [B,1,d]
            n_new = 1
            if sum(cb_eog) == K:
                spans.append(torch.stack(cur, dim=0).numpy().copy())
                cb_eog = [False] * K
                cur_num_gen = 0
                cur = [[] for _ in range(B)]
                if len(spans) == n_spans:
                    break
                mv = more_mask.pop(0)
                empty_ids = torch.full((1, 1), self.empty, dtype=torch.long)
                e_emb = torch.stack([F.embedding(empty_ids, self.sd[f"audio_embedding.{k}.word_embeddings.weight"])
                                     for k in range(K)], dim=0).sum(dim=0)
                s_emb = torch.cat([s_emb, self.sd["mask_embedding"][mv].view(1, 1, -1), e_emb], dim=1)
                n_new = 3
                prev, consec = [None] * B, [0] * B
            y_emb = torch.cat([y_emb, s_emb], dim=1)
        return spans, (keep or 0)

    # ---- public mirrors of the reference API
    @torch.no_grad()
    def inference_tts(self, x, x_lens, y, top_k=-100, top_p=1.0, temperature=1.0, stop_repetition=3,
                      kvcache=1, silence_tokens=(1388, 1898, 131), batch_size=1, trace=None, forced=None,
                      max_steps=None, forced_draws=None):
        assert x.ndim == 2 and x_lens.ndim == 1 and y.ndim == 3
        if self.special_first:
            y = y + self.n_special
        yk = y.transpose(2, 1)                                               # [1,K,T]
        assert yk.shape[0] == 1 and yk.shape[1] == self.K
        T = yk.shape[2]
        cols = delayed_shift(yk.numpy(), self.empty)[0][:, :T + 1]           # drop the last K-1 columns (:967)
        cols = torch.from_numpy(np.ascontiguousarray(cols))
        assert not (cols == self.pad).any()
        spans, _ = self._run(x, cols, {}, mode="tts", more_mask=[], n_spans=1, B=batch_size, top_k=top_k,
                             top_p=top_p, temperature=temperature, stop_repetition=stop_repetition,
                             kvcache=kvcache, silence_tokens=list(silence_tokens), trace=trace, forced=forced,
                             max_steps=max_steps, forced_draws=forced_draws)
        if spans is None:
            return None, None
        gen = torch.from_numpy(unshift_span(spans[0]))                       # [K,Tg]
        res = torch.cat([yk[0], gen], dim=1).unsqueeze(0)
        if self.special_first:
            res, gen = res - self.n_special, gen - self.n_special
        return res, gen.unsqueeze(0)

    @torch.no_grad()
    def tts_logits_for_trajectory(self, x, y, tokens, steps=None):
        """Raw head logits [len(steps),K,V] that inference_tts sees at the given decode steps when the
        emitted tokens are forced to `tokens` [N,K] - computed in ONE full causal pass over
        [x ; prompt columns ; tokens] (the no-cache form of dec_forward, voicecraft.py:449-453; equal
        to the cached path by SURVEY.md §8c-2 and tests/test_oracle_golden.py).  Step s reads the hidden
        state of audio position T + s (the last prompt column for s = 0).  This is how long contexts at
        full model size are checked in seconds instead of minutes."""
        if self.special_first:
            y = y + self.n_special
        yk = y.transpose(2, 1)
        T = yk.shape[2]
        cols = delayed_shift(yk.numpy(), self.empty)[0][:, :T + 1]
        cols = torch.from_numpy(np.ascontiguousarray(cols))
        tok = torch.as_tensor(np.asarray(tokens), dtype=torch.long).reshape(-1, self.K)
        steps = list(range(tok.shape[0] + 1)) if steps is None else list(steps)
        n_tok = max(steps)                                   # step s needs tokens 0..s-1
        allc = torch.cat([cols, tok[:n_tok].t().contiguous()], dim=1)            # [K, T+1+n_tok]
        x_in = self._pos(F.embedding(x, self.sd["text_embedding.word_embeddings.weight"]), "text")
        y_in = self._pos(self._embed_cols(allc.unsqueeze(-1)), "audio")
        Lx, S = x.size(1), x.size(1) + y_in.size(1)
        out, _ = self._stack(torch.cat([x_in, y_in], dim=1), self._causal_rows(1, S, S), None)
        rows = out[:, [Lx + T + s for s in steps]]                                # [1,n,d]
        return torch.stack([self._heads(rows[:, i:i + 1])[0] for i in range(rows.size(1))], dim=0)

    def inference_tts_batch(self, x, x_lens, y, top_k=-100, top_p=1.0, temperature=1.0, stop_repetition=3,
                            kvcache=1, batch_size=5, silence_tokens=(1388, 1898, 131), trace=None, forced_draws=None):
        return self.inference_tts(x, x_lens, y, top_k, top_p, temperature, stop_repetition, kvcache,
                                  silence_tokens, batch_size=batch_size, trace=trace, forced_draws=forced_draws)

    def edit_layout(self, T: int, intervals: list[tuple[int, int]], mask_value: list[int]):
        """Pieces of the rearranged sequence (voicecraft.py:618-679).  Returns
        (non_mask [(s,e)...], prefill description list of ('piece', s, e, term) / ('mask', value) / ('empty',))."""
        M = len(intervals)
        starts = [iv[0] for iv in intervals] + [T]
        ends = [0] + [iv[1] for iv in intervals]
        non_mask = list(zip(ends, starts))
        desc: list[tuple] = []
        for i, (s, e) in enumerate(non_mask):
            if self.eos > 0:
                assert self.reduced_eog
                term = self.eos if i == M else -1
            elif self.reduced_eog:
                term = self.eog if i == M else -1
            else:
                term = self.eog
            desc.append(("piece", s, e, term))
            desc.append(("mask", mask_value[i]))
        desc.append(("empty",))
        return non_mask, desc

    def _edit_cols(self, yk, ivs, mask_value):
        """Prompt columns of an editing call (rearrange / shift / insert_mask / cat_y, voicecraft.py:618-679):
        (non_mask intervals, cols int64 [K,S0], {column: mask_embedding row})."""
        T = yk.shape[2]
        non_mask, desc = self.edit_layout(T, ivs, mask_value)
        ynp = yk[0].numpy()
        chunks, mask_cols, col = [], {}, 0
        for item in desc:
            if item[0] == "piece":
                _, s, e, term = item
                z = ynp[:, s:e]
                if term >= 0:
                    z = np.concatenate([z, np.full((self.K, 1), term, dtype=z.dtype)], axis=1)
                sh = delayed_shift(z[None], self.empty)[0]
            elif item[0] == "mask":
                sh = np.full((self.K, 1), self.eog, dtype=ynp.dtype)          # placeholder ids (:277)
                mask_cols[col] = item[1]
            else:
                sh = np.full((self.K, 1), self.empty, dtype=ynp.dtype)
            chunks.append(sh)
            col += sh.shape[1]
        cols = torch.from_numpy(np.ascontiguousarray(np.concatenate(chunks, axis=1)))
        assert not (cols == self.pad).any()
        return non_mask, cols, mask_cols

    @torch.no_grad()
    def edit_logits_for_trajectory(self, x, y, mask_interval, tokens, steps=None):
        """The editing twin of tts_logits_for_trajectory for ONE masked span: raw head logits [len(steps),K,V] that
        `inference` sees at the given decode steps of that span when the emitted tokens are forced to `tokens` [N,K],
        computed in ONE full causal pass over [x ; rearranged prompt columns ; tokens] (voicecraft.py:449-453 without a
        cache).  Step s reads the hidden state of audio column S0 - 1 + s (S0 = prompt columns)."""
        if self.special_first:
            y = y + self.n_special
        yk = y.transpose(2, 1)
        ivs = [(int(a), int(b)) for a, b in mask_interval[0].tolist()]
        assert len(ivs) == 1, "one span: a span switch feeds three rows at once (covered by the free-running tests)"
        _, cols, mask_cols = self._edit_cols(yk, ivs, list(range(self.max_n_spans))[:1] * 2)
        S0 = cols.shape[1]
        tok = torch.as_tensor(np.asarray(tokens), dtype=torch.long).reshape(-1, self.K)
        steps = list(range(tok.shape[0] + 1)) if steps is None else list(steps)
        n_tok = max(steps)
        allc = torch.cat([cols, tok[:n_tok].t().contiguous()], dim=1)
        y_emb = self._embed_cols(allc.unsqueeze(-1))
        pos = sorted(mask_cols)
        y_emb[0, pos] = self.sd["mask_embedding"][[mask_cols[c] for c in pos]]
        x_in = self._pos(F.embedding(x, self.sd["text_embedding.word_embeddings.weight"]), "text")
        y_in = self._pos(y_emb, "audio")
        Lx, S = x.size(1), x.size(1) + y_in.size(1)
        out, _ = self._stack(torch.cat([x_in, y_in], dim=1), self._causal_rows(1, S, S), None)
        rows = out[:, [Lx + S0 - 1 + s for s in steps]]
        return torch.stack([self._heads(rows[:, i:i + 1])[0] for i in range(rows.size(1))], dim=0)

    @torch.no_grad()
    def inference(self, x, x_lens, y, mask_interval, top_k=-100, top_p=1.0, temperature=1.0,
                  stop_repetition=-1, kvcache=1, silence_tokens=(1388, 1898, 131), trace=None,
                  mask_value=None, forced=None, forced_draws=None):
        assert x.ndim == 2 and x_lens.ndim == 1 and y.ndim == 3
        if self.special_first:
            y = y + self.n_special
        yk = y.transpose(2, 1)
        assert yk.shape[0] == 1 and yk.shape[1] == self.K
        assert mask_interval.shape == torch.Size((1, mask_interval.shape[1], 2))
        ivs = [(int(a), int(b)) for a, b in mask_interval[0].tolist()]
        M = len(ivs)
        if mask_value is None:
            mask_value = list(range(self.max_n_spans))[:M] * 2                 # insert_mask, shuffle off
        non_mask, cols, mask_cols = self._edit_cols(yk, ivs, mask_value)
        spans, _ = self._run(x, cols, mask_cols, mode="edit", more_mask=mask_value[M + 1:], n_spans=M, B=1,
                             top_k=top_k, top_p=top_p, temperature=temperature, stop_repetition=stop_repetition,
                             kvcache=kvcache, silence_tokens=list(silence_tokens), trace=trace, forced=forced,
                             forced_draws=forced_draws)
        out = []
        for (s, e), sp in zip(non_mask, spans):
            out.append(yk[0, :, s:e])
            out.append(torch.from_numpy(unshift_span(sp)))
        out.append(yk[0, :, non_mask[-1][0]:non_mask[-1][1]])
        res = torch.cat(out, dim=1).unsqueeze(0)
        if self.special_first:
            res = res - self.n_special
        return res


    # ---- the training objective, teacher-forced
    def train_layout(self, T: int, intervals: list[tuple[int, int]], mask_value: list[int]):
        """Pieces of one sample's training sequence (voicecraft.py:239-288): every non-masked piece, then every
        masked piece (each + `eog`), a mask placeholder after each piece but the last.  Returns a list of
        ('piece', s, e, term) / ('mask', value)."""
        M = len(intervals)
        assert 1 <= M <= self.max_n_spans and len(mask_value) == M
        starts = [iv[0] for iv in intervals] + [T]
        ends = [0] + [iv[1] for iv in intervals]
        pieces = []
        for i, (s, e) in enumerate(zip(ends, starts)):                      # non-masked pieces (:243-250)
            if self.eos > 0:
                assert self.reduced_eog
                term = self.eos if i == M else -1
            elif self.reduced_eog:
                term = self.eog if i == M else -1
            else:
                term = self.eog
            pieces.append((s, e, term))
        for (s, e) in intervals:                                            # masked pieces, always + eog
            pieces.append((s, e, self.eog))
        values = list(mask_value) + list(mask_value)                        # emb_inds_use + emb_inds_use (:274)
        desc: list[tuple] = []
        for j, (s, e, term) in enumerate(pieces):
            desc.append(("piece", s, e, term))
            if j < len(pieces) - 1:
                desc.append(("mask", values[j]))
        return desc

    @torch.no_grad()
    def forward(self, batch: dict, mask_intervals: list[list[tuple[int, int]]], mask_values: list[list[int]] | None = None,
                codebook_weight: list[float] | None = None):
        """VoiceCraft.forward (voicecraft.py:472-559) with `prepare_mask_intervals` (:198-237, random) replaced by the
        given intervals (frame indices into each sample's y) and `mask_values[i]` = the sample's `emb_inds_use`
        (:270-273; default 0..M-1, i.e. shuffle_mask_embedding = 0).  Batched and padded exactly as the reference:
        pad tokens + key-padding masks.  Returns the reference's dict."""
        x, x_lens, y, y_lens = batch["x"], batch["x_lens"], batch["y"], batch["y_lens"]
        B, K = x.shape[0], self.K
        x = x[:, : int(x_lens.max())]
        y = y[:, :, : int(y_lens.max())]
        assert y.ndim == 3 and y.shape[1] == K
        if mask_values is None:
            mask_values = [list(range(len(iv))) for iv in mask_intervals]
        # ---- text side (:497-500)
        x_input = self._pos(F.embedding(x, self.sd["text_embedding.word_embeddings.weight"]), "text")
        # ---- audio side: rearrange, shift, insert placeholders, concatenate, pad (:322-372)
        cols, targets, mask_pos = [], [], []
        for i in range(B):
            T = int(y_lens[i])
            desc = self.train_layout(T, mask_intervals[i], mask_values[i])
            yi = y[i, :, :T].numpy()
            cur, tg, mp, n = [], [], {}, 0
            for item in desc:
                if item[0] == "piece":
                    _, s0, e0, term = item
                    z = yi[:, s0:e0]
                    if term >= 0:
                        z = np.concatenate([z, np.full((K, 1), term, dtype=z.dtype)], axis=1)
                    assert z.shape[1] > 0, "empty piece (the reference raises inside get_pattern)"
                    tg.append(torch.from_numpy(np.ascontiguousarray(z)))
                    sh = delayed_shift(z[None], self.empty)[0]               # [K, n+K]
                    cur.append(sh); n += sh.shape[1]
                else:
                    mp[n] = item[1]                                           # placeholder column -> mask_embedding row
                    cur.append(np.full((K, 1), self.eog, dtype=yi.dtype)); n += 1
            cols.append(torch.from_numpy(np.concatenate(cur, axis=1)))       # [K, S_i]
            targets.append(tg); mask_pos.append(mp)
        new_y_lens = torch.tensor([c.shape[1] for c in cols])
        S = int(new_y_lens.max())
        cated = torch.full((K, S, B), self.pad, dtype=torch.int64)          # pad_sequence(..., audio_pad_token) (:305)
        for i, c in enumerate(cols):
            cated[:, : c.shape[1], i] = c
        emb = self._embed_cols(cated)                                        # [B,S,d]
        for i in range(B):
            for col, val in mask_pos[i].items():
                emb[i, col] = self.sd["mask_embedding"][val]                 # (:318-319)
        y_input = self._pos(emb, "audio")
        # ---- masks (:416-444): causal over the concatenation (x rows see no y), padded keys removed
        Lx = x.shape[1]
        tri = torch.triu(torch.ones(Lx + S, Lx + S), diagonal=1).bool()
        tri[:Lx, Lx:] = True
        pad = torch.cat([torch.arange(Lx)[None] >= x_lens[:, None], torch.arange(S)[None] >= new_y_lens[:, None]], dim=1)
        m = tri[None] | pad[:, None, :]
        mask = torch.zeros(B, Lx + S, Lx + S).masked_fill_(m, float("-inf"))[:, None].expand(B, self.H, Lx + S, Lx + S).contiguous()
        out, _ = self._stack(torch.cat([x_input, y_input], dim=1), mask, None)
        y_out = out[:, Lx:]
        logits = torch.stack([F.linear(F.gelu(F.linear(y_out, self.sd[f"predict_layer.{k}.0.weight"], self.sd[f"predict_layer.{k}.0.bias"])),
                                       self.sd[f"predict_layer.{k}.2.weight"], self.sd[f"predict_layer.{k}.2.bias"])
                              for k in range(K)], dim=1)                      # [B,K,S,V] (:516)
        # ---- drop the placeholders, revert the delay pattern per piece (:374-404): piece logits [K, n+K, V] ->
        #      [K, n, V] with out[q,t] = logits[q, t+q] (codebooks_patterns.py:209-215 with is_model_output)
        lg_all, tg_all = [], []
        for i in range(B):
            bounds = [-1] + sorted(mask_pos[i].keys()) + [int(new_y_lens[i])]
            for j in range(len(bounds) - 1):
                seg = logits[i, :, bounds[j] + 1: bounds[j + 1]]              # [K, n+K, V]
                n = seg.shape[1] - K
                assert n == targets[i][j].shape[1]
                lg_all.append(torch.stack([seg[q, q: q + n] for q in range(K)], dim=0))
                tg_all.append(targets[i][j])
        lg = torch.cat(lg_all, dim=1)                                        # [K, N, V]
        tg = torch.cat(tg_all, dim=1)                                        # [K, N]
        cw = codebook_weight if codebook_weight is not None else [1.0] * K
        loss, top10, ntok = [], [], []
        for k in range(K):
            loss.append(F.cross_entropy(lg[k], tg[k], reduction="mean"))
            hit = (lg[k].topk(10, dim=1).indices == tg[k][:, None]).any(dim=1)   # torchmetrics 0.11.1 select_topk, micro
            top10.append(hit.float().mean())
            ntok.append(lg[k].shape[0])
        by_cb = [t * n for t, n in zip(top10, ntok)]
        return {"loss": sum(l * n * c for l, n, c in zip(loss, ntok, cw)), "top10acc": sum(by_cb),
                "top10acc_by_codebook": by_cb, "effective_ntoken": torch.tensor(sum(ntok)),
                "_per_token_logits": lg, "_targets": tg,
                # test hooks: per sample, the concatenated columns [K,S_i], {placeholder column: mask_embedding row} and
                # the per-piece targets
                "_cols": cols, "_mask_pos": mask_pos, "_targets_per_sample": targets}


def prompt_columns_tts(y_TK: np.ndarray, empty: int) -> np.ndarray:
    """[T,K] -> [K,T+1] prompt columns of inference_tts (for kernel-level tests)."""
    z = np.ascontiguousarray(y_TK.T)[None]
    return delayed_shift(z, empty)[0][:, : y_TK.shape[0] + 1]
