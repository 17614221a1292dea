"""CPU: the oracle's restatement of the training objective (VoiceCraft.forward, models/voicecraft.py:472-559) against the
reference-made fixtures tests/golden/fwd_*.npz.  The reference ran with its random mask intervals replaced
by the fixture's; loss is compared to fp32 rounding, hit counts and token counts exactly."""
import ast
import os

import numpy as np
import pytest
import torch

from _util import GOLDEN, build_forward_case
from oracle.gen_golden import FORWARD_CASES
from oracle.voicecraft_oracle import VoiceCraftOracle


def weights_of(spec, K):
    return [float(w) for w in ast.literal_eval(spec["codebook_weight"])] if spec["codebook_weight"] else [1.0] * K


@pytest.mark.parametrize("name", sorted(FORWARD_CASES))
def test_oracle_forward_matches_reference(name):
    spec, args, sd, batch = build_forward_case(name)
    g = np.load(os.path.join(GOLDEN, f"{name}.npz"))
    for k in ("x", "x_lens", "y", "y_lens"):
        assert np.array_equal(g[k], batch[k].numpy()), k
    orc = VoiceCraftOracle(args, sd)
    out = orc.forward(batch, spec["spans"], codebook_weight=weights_of(spec, args.n_codebooks))
    assert int(out["effective_ntoken"]) == int(g["effective_ntoken"])
    assert abs(float(out["loss"]) - float(g["loss"])) <= 2e-6 * abs(float(g["loss"]))
    got = np.array([float(t) for t in out["top10acc_by_codebook"]])
    assert np.allclose(got, g["top10acc_by_codebook"], atol=1e-3)          # n * mean(hits): integers up to fp32 rounding
    assert abs(float(out["top10acc"]) - float(g["top10acc"])) <= 1e-3


def test_padding_does_not_change_a_sample():
    """The reference pads the batch and masks padded keys; a sample alone must give its own share of the batch loss."""
    spec, args, sd, batch = build_forward_case("fwd_b3_ragged")
    orc = VoiceCraftOracle(args, sd)
    whole = orc.forward(batch, spec["spans"])
    total, ntok = 0.0, 0
    for i in range(batch["x"].shape[0]):
        one = {"x": batch["x"][i: i + 1], "x_lens": batch["x_lens"][i: i + 1], "y": batch["y"][i: i + 1], "y_lens": batch["y_lens"][i: i + 1]}
        o = orc.forward(one, [spec["spans"][i]])
        total += float(o["loss"]); ntok += int(o["effective_ntoken"])
    assert ntok == int(whole["effective_ntoken"])
    assert abs(total - float(whole["loss"])) <= 1e-4 * abs(float(whole["loss"]))


def test_layout_of_a_training_sequence():
    spec, args, sd, _ = build_forward_case("fwd_b1_1span")
    orc = VoiceCraftOracle(args, sd)
    desc = orc.train_layout(48, [(12, 25)], [0])
    assert desc == [("piece", 0, 12, -1), ("mask", 0), ("piece", 25, 48, args.eos), ("mask", 0), ("piece", 12, 25, args.eog)]


def test_engine_host_layout_equals_the_oracle_layout():
    """vc_eval_layout (the host half of vc_eval_forward: segment table + target table of one utterance) against the oracle's
    rearranged columns, placeholder positions and targets - no GPU involved, the library is only loaded."""
    import ctypes as C
    from voicecraft_amd import _lib
    lib = _lib.load()
    for name in sorted(FORWARD_CASES):
        spec, args, sd, batch = build_forward_case(name)
        K = args.n_codebooks
        orc = VoiceCraftOracle(args, sd)
        ref = orc.forward(batch, spec["spans"])
        cfg = _lib.ModelCfg(d_model=args.d_model, nhead=args.nhead, num_layers=args.num_decoder_layers, n_codebooks=K,
                            audio_vocab_size=args.audio_vocab_size, n_special=int(args.n_special), text_rows=args.text_vocab_size + 1,
                            head_hidden=args.audio_vocab_size // 2, empty_token=args.empty_token, eog=args.eog,
                            audio_pad_token=args.audio_pad_token, eos=args.eos if args.eos > 0 else -1,
                            reduced_eog=int(args.reduced_eog or 0), encodec_sr=50, max_n_spans=args.max_n_spans, max_seqs=4, max_positions=512)
        y_off = 0
        for i, spans in enumerate(spec["spans"]):
            Lx, T, M = int(batch["x_lens"][i]), int(batch["y_lens"][i]), len(spans)
            yi = batch["y"][i, :, :T].numpy()                                   # [K,T]
            flat = (C.c_int32 * (2 * M))(*[v for se in spans for v in se])
            mv = (C.c_int32 * M)(*range(M))
            seg = (C.c_int32 * (32 * 6))()
            n_seg, n_cols = C.c_int(0), C.c_int(0)
            cap = (Lx + T + 64) * K * 2
            tgt = (C.c_int32 * cap)()
            rc = lib.vc_eval_layout(C.byref(cfg), Lx, T, flat, M, mv, y_off, seg, C.byref(n_seg), C.byref(n_cols), tgt, cap)
            assert rc == 0, (name, i, rc)
            want_cols = ref["_cols"][i].numpy()                                  # [K,S]
            S = want_cols.shape[1]
            assert n_cols.value == S, (name, i, n_cols.value, S)
            segs = np.array(seg[: 6 * n_seg.value]).reshape(-1, 6)
            # what prompt_k puts into every audio column (vc_tokens.hip: token s-1-q of the piece, else `empty`)
            got_cols = np.full((K, S), -7, dtype=np.int64)
            got_mask = {}
            for col0, ncols, src0, src_len, term, mval in segs:
                for s in range(ncols):
                    if mval >= 0:
                        got_mask[col0 + s] = int(mval)
                        got_cols[:, col0 + s] = args.eog                          # the reference's placeholder token (:283)
                        continue
                    n = src_len + (1 if term >= 0 else 0)
                    for q in range(K):
                        j = s - 1 - q
                        got_cols[q, col0 + s] = (yi[q, src0 + j] if j < src_len else term) if 0 <= j < n else args.empty_token
            assert got_mask == ref["_mask_pos"][i], (name, i)
            assert np.array_equal(got_cols, want_cols), (name, i)
            # targets, codebook by codebook in row order == the oracle's per-piece targets concatenated
            t = np.array(tgt[: (Lx + S) * K]).reshape(Lx + S, K)
            assert (t[:Lx] == -1).all()
            want_t = torch.cat(ref["_targets_per_sample"][i], dim=1).numpy()     # [K, N_i]
            flat_y = batch["y"][i, :, :T].numpy().T.reshape(-1)                  # this utterance's [frames][K], frames from y_off
            for q in range(K):
                col = t[Lx:, q]
                vals = [int(flat_y[v - y_off * K]) if v >= 0 else -(v + 2) for v in col if v != -1]
                assert vals == [int(v) for v in want_t[q]], (name, i, q)
            y_off += T


def test_the_kernel_definitions_of_loss_and_hit_agree_with_the_reference():
    """ce_rows_k's arithmetic, restated in numpy on the oracle's logits: nll = log(sum exp(v - max)) + max - v[target];
    hit = fewer than ten logits above the target's.  Summed per codebook this must give the fixture's loss and hits."""
    for name in sorted(FORWARD_CASES):
        spec, args, sd, batch = build_forward_case(name)
        g = np.load(os.path.join(GOLDEN, f"{name}.npz"))
        ref = VoiceCraftOracle(args, sd).forward(batch, spec["spans"])
        lg, tg = ref["_per_token_logits"].numpy().astype(np.float32), ref["_targets"].numpy()
        cw = weights_of(spec, args.n_codebooks)
        loss, hits = 0.0, []
        for k in range(lg.shape[0]):
            m = lg[k].max(axis=1, keepdims=True)
            tl = np.take_along_axis(lg[k], tg[k][:, None], axis=1)
            nll = np.log(np.exp(lg[k] - m).sum(axis=1, dtype=np.float32)) + m[:, 0] - tl[:, 0]
            loss += float(nll.astype(np.float64).sum()) * cw[k]
            hits.append(int(((lg[k] > tl).sum(axis=1) < 10).sum()))
        assert abs(loss - float(g["loss"])) <= 2e-5 * abs(float(g["loss"])), (name, loss, float(g["loss"]))
        assert hits == [int(round(v)) for v in g["top10acc_by_codebook"]], (name, hits, g["top10acc_by_codebook"])
