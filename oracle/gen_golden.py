"""Generates tests/golden/*.npz by running the REAL reference.  TEST INFRASTRUCTURE ONLY.

Run in the build container (needs /root/reference):   python -m oracle.gen_golden
Every fixture stores the inputs, the reference's outputs and (for model runs) the raw head logits
of every decode step, sub-sampled, plus the first steps in full.  Checkpoints are not stored: they
are regenerated from (preset, seed) by voicecraft_amd.synth, which uses a frozen RNG stream.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import ref_loader  # noqa: E402
from voicecraft_amd import synth  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
FULL_STEPS = 3       # steps whose [K,V] logits are stored in full
STRIDE = 61          # sub-sampling stride over V for all steps

# name -> spec.  `arg_kw` feeds synth.make_args, `knobs` the reference call.
MODEL_CASES = {
    "tts_greedy": dict(preset="tiny", arg_kw={}, wseed=3, prompt=(6, 21, 11), mode="tts",
                       knobs=dict(top_k=1, top_p=1.0, temperature=1.0, stop_repetition=3, kvcache=1)),
    "tts_greedy_nokv": dict(preset="tiny", arg_kw={}, wseed=3, prompt=(6, 21, 11), mode="tts",
                            knobs=dict(top_k=1, top_p=1.0, temperature=1.0, stop_repetition=3, kvcache=0)),
    "tts_greedy_hd128": dict(preset="tiny128", arg_kw={}, wseed=5, prompt=(5, 17, 12), mode="tts",
                             knobs=dict(top_k=1, top_p=1.0, temperature=1.0, stop_repetition=-1, kvcache=1)),
    "tts_sampled": dict(preset="tiny", arg_kw={}, wseed=3, prompt=(7, 30, 13), mode="tts", tseed=1234,
                        knobs=dict(top_k=40, top_p=0.9, temperature=0.8, stop_repetition=2, kvcache=1)),
    "tts_oldscheme": dict(preset="tiny", arg_kw=dict(eos=-1, n_special=3, reduced_eog=0), wseed=4,
                          prompt=(6, 19, 14), mode="tts",
                          knobs=dict(top_k=1, top_p=1.0, temperature=1.0, stop_repetition=3, kvcache=1)),
    "tts_batch_greedy": dict(preset="tiny", arg_kw={}, wseed=3, prompt=(6, 21, 11), mode="tts_batch",
                             knobs=dict(top_k=1, top_p=1.0, temperature=1.0, stop_repetition=3, kvcache=1, batch_size=3)),
    "tts_early_stop": dict(preset="tiny", arg_kw={}, wseed=6, prompt=(3, 40, 15), mode="tts",   # prompt longer than the cap
                           knobs=dict(top_k=1, top_p=1.0, temperature=1.0, stop_repetition=3, kvcache=1)),
    "edit_1span": dict(preset="tiny", arg_kw={}, wseed=3, prompt=(8, 60, 16), mode="edit", spans=[(20, 31)],
                       knobs=dict(top_k=1, top_p=1.0, temperature=1.0, stop_repetition=-1, kvcache=1)),
    "edit_2span": dict(preset="tiny", arg_kw={}, wseed=3, prompt=(9, 64, 17), mode="edit", spans=[(10, 18), (40, 47)],
                       knobs=dict(top_k=1, top_p=1.0, temperature=1.0, stop_repetition=-1, kvcache=1)),
    "edit_3span_edges": dict(preset="tiny", arg_kw={}, wseed=7, prompt=(9, 50, 18), mode="edit",
                             spans=[(1, 5), (20, 20), (44, 50)],     # 1-frame head piece, empty span, span to the end
                             # (a span starting at frame 0 makes the reference itself raise IndexError:
                             #  zero-length piece in build_pattern_sequence, codebooks_patterns.py:174)
                             knobs=dict(top_k=1, top_p=1.0, temperature=1.0, stop_repetition=-1, kvcache=1)),
    "edit_oldscheme": dict(preset="tiny", arg_kw=dict(eos=-1, n_special=3, reduced_eog=0), wseed=4,
                           prompt=(8, 55, 19), mode="edit", spans=[(15, 25)],
                           knobs=dict(top_k=1, top_p=1.0, temperature=1.0, stop_repetition=-1, kvcache=1)),
    "edit_reduced_noeos": dict(preset="tiny", arg_kw=dict(eos=-1, n_special=3, reduced_eog=1), wseed=4,
                               prompt=(8, 55, 19), mode="edit", spans=[(15, 25)],
                               knobs=dict(top_k=1, top_p=1.0, temperature=1.0, stop_repetition=-1, kvcache=1)),
    "edit_sampled": dict(preset="tiny", arg_kw={}, wseed=3, prompt=(8, 60, 16), mode="edit", spans=[(20, 31)], tseed=99,
                         knobs=dict(top_k=30, top_p=0.8, temperature=1.0, stop_repetition=2, kvcache=1)),
    # ---- round 2: the terminator is NOT muted (sd_kw), so the end-of-generation branches fire
    # (voicecraft.py:1024 min-length guard, :1041-1045 sampled / arg-max terminator, :1027-1031 silence penalty,
    #  :1296-1302 best-of-N keep).  `boost` = (codebook, token, delta) added to the head bias.
    "tts_eos_guard": dict(preset="tiny", arg_kw={}, wseed=3, prompt=(8, 21, 31), mode="tts",          # terminator wins at once:
                          sd_kw=dict(mute_eos=False, boost=[(0, 2051, 8.0)]),                        # ends when the guard releases
                          knobs=dict(top_k=1, top_p=1.0, temperature=1.0, stop_repetition=3, kvcache=1)),
    "tts_eos_greedy": dict(preset="tiny", arg_kw={}, wseed=3, prompt=(8, 21, 31), mode="tts",
                           sd_kw=dict(mute_eos=False, boost=[(0, 2051, 0.4)]),                       # ends somewhere in the middle
                           knobs=dict(top_k=1, top_p=1.0, temperature=1.0, stop_repetition=3, kvcache=1)),
    "tts_oldscheme_eog": dict(preset="tiny", arg_kw=dict(eos=-1, n_special=3, reduced_eog=0), wseed=4, prompt=(7, 19, 32),
                              mode="tts", sd_kw=dict(mute_eos=False, boost=[(0, 2049, 0.62)]),
                              knobs=dict(top_k=1, top_p=1.0, temperature=1.0, stop_repetition=-1, kvcache=1)),
    "tts_silence_sr1": dict(preset="tiny", arg_kw={}, wseed=3, prompt=(6, 15, 33), mode="tts",
                            sd_kw=dict(mute_eos=True, boost=[(0, 131, 1.2)]),
                            knobs=dict(top_k=1, top_p=1.0, temperature=1.0, stop_repetition=1, kvcache=1)),
    "tts_silence_sr2": dict(preset="tiny", arg_kw={}, wseed=3, prompt=(6, 15, 33), mode="tts",
                            sd_kw=dict(mute_eos=True, boost=[(0, 131, 1.2)]),
                            knobs=dict(top_k=1, top_p=1.0, temperature=1.0, stop_repetition=2, kvcache=1)),
    "tts_silence_sr3": dict(preset="tiny", arg_kw={}, wseed=3, prompt=(6, 15, 33), mode="tts",
                            sd_kw=dict(mute_eos=False, boost=[(0, 1388, 1.2), (0, 2051, 0.5)]),
                            knobs=dict(top_k=1, top_p=1.0, temperature=1.0, stop_repetition=3, kvcache=1,
                                       silence_tokens=[1388, 1898, 131])),
    "edit_eog_greedy": dict(preset="tiny", arg_kw={}, wseed=3, prompt=(9, 64, 34), mode="edit", spans=[(10, 18), (40, 47)],
                            sd_kw=dict(mute_eos=False, boost=[(0, 2049, 0.55)]),
                            knobs=dict(top_k=1, top_p=1.0, temperature=1.0, stop_repetition=2, kvcache=1)),
    "edit_eog_oldscheme": dict(preset="tiny", arg_kw=dict(eos=-1, n_special=3, reduced_eog=0), wseed=4, prompt=(8, 55, 35),
                               mode="edit", spans=[(15, 25)], sd_kw=dict(mute_eos=False, boost=[(0, 2049, 0.45)]),
                               knobs=dict(top_k=1, top_p=1.0, temperature=1.0, stop_repetition=-1, kvcache=1)),
    # sampled runs: the reference's own draws are recorded (`draws`) and replayed through the device
    # state machine (vc_sample_cfg.forced_mode = draws), so everything but the RNG stream is compared
    "tts_sampled_eos": dict(preset="tiny", arg_kw={}, wseed=3, prompt=(9, 30, 36), mode="tts", tseed=4321,
                            sd_kw=dict(mute_eos=False, boost=[(0, 2051, 1.5), (0, 131, 4.0)]),
                            knobs=dict(top_k=40, top_p=0.9, temperature=0.8, stop_repetition=1, kvcache=1)),
    "tts_batch4_sampled": dict(preset="tiny", arg_kw={}, wseed=3, prompt=(9, 30, 37), mode="tts_batch", tseed=777,
                               sd_kw=dict(mute_eos=False, boost=[(0, 2051, 3.0), (0, 131, 4.0)]),
                               knobs=dict(top_k=0, top_p=0.95, temperature=1.0, stop_repetition=2, kvcache=1, batch_size=4)),
    "tts_batch3_sampled_b": dict(preset="tiny", arg_kw={}, wseed=5, prompt=(8, 25, 38), mode="tts_batch", tseed=778,
                                 sd_kw=dict(mute_eos=False, boost=[(0, 700, 4.0), (0, 2051, 2.2)]),
                                 knobs=dict(top_k=20, top_p=1.0, temperature=1.3, stop_repetition=3, kvcache=1, batch_size=3)),
    "edit_sampled_eog": dict(preset="tiny", arg_kw={}, wseed=3, prompt=(8, 60, 39), mode="edit", spans=[(12, 20), (30, 41)], tseed=98,
                             sd_kw=dict(mute_eos=False, boost=[(0, 500, 3.0), (0, 2049, 2.0)]),
                             knobs=dict(top_k=30, top_p=0.8, temperature=1.0, stop_repetition=2, kvcache=1)),
}


# Training objective, teacher-forced (VoiceCraft.forward, models/voicecraft.py:472-559).  The reference draws
# its mask intervals at random (:198-237); the fixtures fix them by replacing `prepare_mask_intervals` on the live model.
# `samples` = (Lx, T, seed) per utterance of the batch, `spans` = its mask intervals (frame indices).
# The audio tokens of these batches are folded into [0, 16) and twelve of those ids get a bias boost on every head, so
# that the target is among the ten largest logits for roughly half of the positions: the top-10 metric is exercised with
# hits AND misses (with flat random logits it would be 0 almost everywhere).
_BOOST12 = [(-1, t, 4.0) for t in range(12)]
FORWARD_CASES = {
    "fwd_b1_1span": dict(preset="tiny", arg_kw={}, wseed=11, sd_kw=dict(mute_eos=False, boost=_BOOST12),
                         samples=[(9, 48, 21)], spans=[[(12, 25)]], codebook_weight=None),
    "fwd_b3_ragged": dict(preset="tiny", arg_kw={}, wseed=12, sd_kw=dict(mute_eos=False, boost=_BOOST12),
                          samples=[(12, 60, 22), (7, 41, 23), (10, 33, 24)],
                          spans=[[(10, 20), (35, 40)], [(5, 17)], [(3, 8), (12, 19), (25, 30)]],
                          codebook_weight="[5,1,0.5,0.1]"),        # the released models' weights (z_scripts/e830M.sh)
    "fwd_oldscheme": dict(preset="tiny", arg_kw=dict(eos=-1, n_special=3, reduced_eog=0), wseed=13,
                          sd_kw=dict(mute_eos=False, boost=_BOOST12),
                          samples=[(8, 50, 25), (11, 37, 26)], spans=[[(20, 31)], [(4, 9), (20, 28)]], codebook_weight=None),
    "fwd_hd128": dict(preset="tiny128", arg_kw={}, wseed=14, sd_kw=dict(mute_eos=False, boost=_BOOST12),
                      samples=[(10, 70, 27), (6, 52, 28)], spans=[[(1, 6), (30, 44)], [(26, 51)]],
                      codebook_weight="[5,1,0.5,0.1]"),
}


def forward_inputs(spec, args):
    """Padded batch of a FORWARD case (shared with the tests)."""
    parts = [synth.random_prompt(args, Lx, T, seed=sd) for (Lx, T, sd) in spec["samples"]]
    B, K = len(parts), args.n_codebooks
    x_lens = torch.tensor([p[0].shape[1] for p in parts], dtype=torch.int64)
    y_lens = torch.tensor([p[2].shape[1] for p in parts], dtype=torch.int64)
    x = torch.full((B, int(x_lens.max())), args.text_pad_token, dtype=torch.int64)
    y = torch.full((B, K, int(y_lens.max())), args.audio_pad_token, dtype=torch.int64)
    for i, (xi, _, yi) in enumerate(parts):
        x[i, : xi.shape[1]] = xi[0]
        y[i, :, : yi.shape[1]] = yi[0].transpose(0, 1) % 16
    return dict(x=x, x_lens=x_lens, y=y, y_lens=y_lens)


def run_reference_forward(spec):
    args = synth.make_args(spec["preset"], **spec["arg_kw"])
    args.codebook_weight = spec["codebook_weight"]
    sd = case_state_dict(spec, args)
    model = ref_loader.build_reference_model(args, sd)
    batch = forward_inputs(spec, args)
    spans = spec["spans"]

    def fixed_intervals(y_lens):               # what prepare_mask_intervals (:198-237) returns, for the chosen spans
        mi = [list(v) for v in spans]
        nmi = []
        for i, v in enumerate(spans):
            T = int(y_lens[i])
            nmi.append(list(zip([0] + [e for _, e in v], [s for s, _ in v] + [T])))
        return mi, nmi
    model.prepare_mask_intervals = fixed_intervals
    with torch.no_grad():
        ref = model.forward(batch)
    out = {k: v.numpy() for k, v in batch.items()}
    out["loss"] = np.float64(float(ref["loss"]))
    out["top10acc"] = np.float64(float(ref["top10acc"]))
    out["top10acc_by_codebook"] = np.array([float(t) for t in ref["top10acc_by_codebook"]], dtype=np.float64)
    out["effective_ntoken"] = np.int64(int(ref["effective_ntoken"]))
    return out


def case_state_dict(spec, args):
    """The synthetic checkpoint of a golden case (shared with tests/_util.py)."""
    kw = dict(mute_eos=True)
    kw.update(spec.get("sd_kw", {}))
    return synth.make_state_dict(args, seed=spec["wseed"], perturb=True, **kw)


def run_reference_case(spec):
    args = synth.make_args(spec["preset"], **spec["arg_kw"])
    # round-1 cases mute the terminator (lengths set by the reference's own cap);
    # the round-2 cases un-mute it through spec["sd_kw"]
    sd = case_state_dict(spec, args)
    model = ref_loader.build_reference_model(args, sd)
    vc_mod, _ = ref_loader.import_reference()
    draws = []
    orig_sampling = vc_mod.topk_sampling

    def recording_sampling(*a, **k):          # the reference's raw draws, before the state machine's overrides
        o = orig_sampling(*a, **k)
        draws.append(o.detach().clone().reshape(-1))
        return o
    vc_mod.topk_sampling = recording_sampling
    Lx, T, pseed = spec["prompt"]
    x, x_lens, y = synth.random_prompt(args, Lx, T, seed=pseed)
    captured = []
    hooks = [m.register_forward_hook(lambda mod, inp, out: captured.append(out.detach().clone()))
             for m in model.predict_layer]
    if "tseed" in spec:
        torch.manual_seed(spec["tseed"])
    kn = dict(spec["knobs"])
    out = {}
    with torch.no_grad():
        if spec["mode"] == "tts":
            res, gen = model.inference_tts(x, x_lens, y, **kn)
            out["res"], out["gen"] = res.numpy(), gen.numpy()
        elif spec["mode"] == "tts_batch":
            res, gen = model.inference_tts_batch(x, x_lens, y, **kn)
            out["res"], out["gen"] = res.numpy(), gen.numpy()
        else:
            mi = torch.tensor([spec["spans"]], dtype=torch.int64)
            res = model.inference(x, x_lens, y, mi, **kn)
            out["res"] = res.numpy()
            out["mask_interval"] = mi.numpy()
    for h in hooks:
        h.remove()
    vc_mod.topk_sampling = orig_sampling
    K = args.n_codebooks
    out["draws"] = torch.stack(draws).reshape(len(draws), -1, K).numpy().astype(np.int64)      # [steps,B,K]
    steps = len(captured) // K
    lg = torch.stack([torch.stack([captured[s * K + k] for k in range(K)], dim=0) for s in range(steps)], dim=0)
    lg = lg.reshape(steps, K, -1, lg.shape[-1])          # [steps,K,B,V]
    out["logits_full"] = lg[:FULL_STEPS, :, 0].numpy().astype(np.float32)
    out["logits_sub"] = lg[:, :, 0, ::STRIDE].numpy().astype(np.float32)
    out["n_steps"] = np.int64(steps)
    out["x"], out["y"] = x.numpy(), y.numpy()
    return out


def gen_pattern():
    _, cp = ref_loader.import_reference()
    rs = np.random.RandomState(0)
    out = {}
    for K in (4, 3, 8):
        prov = cp.DelayedPatternProvider(n_q=K)
        for T in (0, 1, 2, 3, 4, 5, 6, 7, 33, 150):
            if T == 0:
                continue        # the reference's Pattern cannot be built for T=0 with K>1 edge; covered by closed form only
            z = rs.randint(0, 2048, size=(2, K, T)).astype(np.int64)
            pat = prov.get_pattern(T)
            vals, idx, mask = pat.build_pattern_sequence(torch.from_numpy(z), 2048, keep_only_valid_steps=False)
            out[f"z_K{K}_T{T}"] = z
            out[f"shift_K{K}_T{T}"] = vals.numpy()
            rv, _, _ = pat.revert_pattern_sequence(vals, 2048, keep_only_valid_steps=False)
            out[f"revert_K{K}_T{T}"] = rv.numpy()
            # revert of a truncated sequence (S < T+K), as inference_tts cuts it (voicecraft.py:967)
            cut = vals[:, :, : T + 1]
            rv2, _, _ = pat.revert_pattern_sequence(cut.contiguous(), 2048, keep_only_valid_steps=False)
            out[f"revertcut_K{K}_T{T}"] = rv2.numpy()
    np.savez_compressed(os.path.join(GOLDEN, "pattern.npz"), **out)
    print("pattern.npz:", len(out), "arrays")


def gen_sampler():
    vc, _ = ref_loader.import_reference()
    out = {}
    rs = np.random.RandomState(7)
    lg = torch.from_numpy(rs.standard_normal(size=(6, 257)).astype(np.float32) * 3)
    lg[0, 5] = lg[0, 9]                                       # a tie
    out["logits"] = lg.numpy().copy()
    for name, (k, p) in {"k5": (5, 1.0), "k40": (40, 1.0), "p08": (0, 0.8), "k20p06": (20, 0.6), "none": (-100, 1.0)}.items():
        out[f"filt_{name}"] = vc.top_k_top_p_filtering(lg.clone(), top_k=k, top_p=p).numpy()
    out["kat_topk"] = vc.top_k_top_p_filtering(torch.tensor([[1., 3., 3., 2., 0.]]), top_k=2).numpy()
    out["kat_topp"] = vc.top_k_top_p_filtering(torch.log(torch.tensor([[.5, .3, .15, .05]])), top_p=0.8).numpy()
    torch.manual_seed(5)
    out["draws_seed5"] = torch.stack([vc.topk_sampling(lg.clone(), top_k=10, top_p=0.9, temperature=0.7) for _ in range(8)]).numpy()
    np.savez_compressed(os.path.join(GOLDEN, "sampler.npz"), **out)
    print("sampler.npz:", len(out), "arrays")


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    torch.set_num_threads(8)
    if "--forward-only" in sys.argv:            # the training-objective fixtures only (the others are unchanged)
        for name, spec in FORWARD_CASES.items():
            out = run_reference_forward(spec)
            np.savez_compressed(os.path.join(GOLDEN, f"{name}.npz"), **out)
            print(f"{name}.npz: loss={float(out['loss']):.4f} top10={out['top10acc_by_codebook']} ntoken={int(out['effective_ntoken'])}")
        return
    gen_pattern()
    gen_sampler()
    for name, spec in MODEL_CASES.items():
        out = run_reference_case(spec)
        np.savez_compressed(os.path.join(GOLDEN, f"model_{name}.npz"), **out)
        print(f"model_{name}.npz: steps={int(out['n_steps'])} res={out['res'].shape}")
    for name, spec in FORWARD_CASES.items():
        out = run_reference_forward(spec)
        np.savez_compressed(os.path.join(GOLDEN, f"{name}.npz"), **out)
        print(f"{name}.npz: loss={float(out['loss']):.4f} top10={out['top10acc_by_codebook']} ntoken={int(out['effective_ntoken'])}")


if __name__ == "__main__":
    main()
