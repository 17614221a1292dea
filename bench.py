#!/usr/bin/env python
"""bench.py — codec-tokens/s of the VoiceCraft TTS decode path on MI355X (BASELINE.json metric).

A "step" is one whole `inference_tts` call (prompt build + prefill + every decode step + un-shift)
on the synthetic 16 s workload of BASELINE config 3: giga830M shape, bf16, batch 1 per GPU,
top_k=40, Lx=80 phonemes, 150 prompt frames -> 650 generated frames (the reference's own length
cap ends generation).  codec tokens = K * generated frames.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python bench.py --gpus N ...            (no launcher: starts its own N ranks through torch.distributed.run)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU, every rank decodes its own utterances (utterance u -> rank u mod N;
no data-path collective), then ONE RCCL all_gather of the padded token blocks inside the timed
region.  Weak scaling: per-GPU work is fixed.  Rank 0 prints exactly one JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # the host driver only supports dmabuf IPC (RCCL / tensor sharing across ranks)

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_PEAK_TF = {"bf16": 2500.0, "fp32": 157.3}     # dense MFMA peaks (same guide)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--preset", default="giga830M")
    p.add_argument("--dtype", default="bf16")
    p.add_argument("--batch", type=int, default=1, help="utterances decoded together per GPU (config 5 uses 8)")
    p.add_argument("--lx", type=int, default=80)
    p.add_argument("--prompt-frames", type=int, default=150)
    p.add_argument("--top-k", type=int, default=40)
    p.add_argument("--mode", default="tts", choices=["tts", "edit"],
                   help="edit: BASELINE config 4 - one masked span in the middle of a 16 s utterance is re-generated")
    p.add_argument("--no-graph", action="store_true")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-steps", type=int, default=64, help="decode steps of the bounded CPU sample (half at the start, half at the end of the run's context)")
    p.add_argument("--cpu-threads", type=int, default=16, help="torch intra-op threads for the CPU baseline (capped at the core count)")
    p.add_argument("--no-codec", action="store_true", help="skip the EnCodec encode/decode timing block")
    p.add_argument("--no-configs", action="store_true", help="skip the block that times BASELINE configs 1, 2, 4 and 5's per-GPU share on their own engines")
    p.add_argument("--dump", default=None, help="rank 0 writes the token blocks gathered in the last step to this .npz (tests)")
    p.add_argument("--cpu-baseline-only", action="store_true",
                   help="time only the CPU leg (the port, and the unmodified reference when VC_REFERENCE_ROOT names its tree); needs no GPU")
    p.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                   help="collective backend for N > 1: nccl = RCCL over xGMI (one GPU per rank); gloo = host collectives, usable when "
                        "several ranks share ONE GPU (VC_RANKS_SHARE_DEVICE=1: the single-GPU test of the N > 1 code path)")
    p.add_argument("--ab", default="auto", metavar="KNOB=A:B",
                   help="in-process A/B of one engine option (vc_set_option), e.g. fr_one=0:1 - interleaved pairs of whole calls, "
                        "reported as the `ab` object of the JSON line.  auto (default, N = 1 only): the default-ON launch-shape feature "
                        "of this run's step - fr_one=0:1 at one row per step, finished_rows=0:16 at 2..16 rows, wide_gemm=0:1 above; none: skip")
    p.add_argument("--ab-pairs", type=int, default=7)
    return p.parse_args()


def lib_stamp():
    """First 16 hex digits of the digest of the library's sources (voicecraft_amd/build.py writes it next to the .so): the committed
    in-situ trace and PMC pass are quoted only when they were taken on THIS build."""
    try:
        with open(os.path.join(ROOT, "voicecraft_amd", ".build_stamp")) as f:
            return f.read()[:16]
    except Exception:
        return None


def pmc_traffic(kernel, args):
    """HBM bytes per launch of `kernel` from the committed PMC pass (rocprofv3 --pmc FETCH_SIZE in its own run of THIS
    command, x2-corrected for gfx950 as the microarchitecture guide prescribes).  None unless the pass on file was taken
    on the same preset / dtype / batch / Lx / prompt length as this run (profiles/pmc_traffic.json "config") AND on this build of the
    library ("lib_stamp").  Returned with its source file and box: it is a committed measurement, not one of this run."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            j = json.load(f)
        want = {"preset": args.preset, "dtype": args.dtype, "batch": args.batch, "lx": args.lx,
                "prompt_frames": args.prompt_frames, "mode": args.mode}
        if j.get("config") != want or j.get("lib_stamp") != lib_stamp():
            return None
        return {"bytes": int(j["kernels"][kernel]["fetch_bytes_per_launch"]), "source": j.get("source"), "box": j.get("box"),
                "lib_stamp": j.get("lib_stamp")}
    except Exception:
        return None


def in_situ(kernel, args):
    """The same kernel's IN-SITU average from the committed rocprofv3 kernel trace of THIS command (profiles/in_situ.json, written by
    tools/in_situ_to_json.py from a tools/prof_decode.sh summary): every launch of the decode loop, rotating layers, caches as the
    step leaves them - the microbenchmark of `roofline.achieved` runs the kernel back to back on a quiet chip and comes out ~5 %
    higher.  None unless the trace on file was taken on this run's preset / dtype / batch / Lx / prompt length / mode and on this
    build of the library ("lib_stamp" = digest of its sources)."""
    try:
        with open(os.path.join(ROOT, "profiles", "in_situ.json")) as f:
            j = json.load(f)
        want = {"preset": args.preset, "dtype": args.dtype, "batch": args.batch, "lx": args.lx,
                "prompt_frames": args.prompt_frames, "mode": args.mode}
        if j.get("config") != want or j.get("lib_stamp") != lib_stamp():
            return None
        k = j["kernels"][kernel]
        out = {"kernel": k["name"], "avg_us": k["avg_us"], "calls": k["calls"], "source": j.get("source"), "box": j.get("box"),
               "lib_stamp": j.get("lib_stamp")}
        if k.get("algorithmic_bytes"):
            out["frac"] = round(k["algorithmic_bytes"] / (k["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
        return out
    except Exception:
        return None


# option name -> (key of the option-state text, positions of its values there)
OPTION_STATE = {"graph_steps": ("g", (0,)), "nt": ("nt", (0, 1)), "finished_rows": ("fr", (0,)), "fr_pair": ("fr", (1,)), "att_p16": ("fr", (2,)), "hq": ("fr", (3,)),
                "tile_attn": ("ta", (0, 1)), "fr_one": ("r1", (0,)), "attn_fast": ("r1", (1,)), "qkv_p8": ("r1", (2,)), "qkv16": ("q16", (0,)),
                "wide_heads": ("q16", (1,)), "wide_gemm": ("q16", (2,)), "wd_stage": ("q16", (3,)), "shrink": ("sh", (0,))}


def option_value(text, knob):
    """The value of one option in the engine's option-state text, as vc_set_option takes it; None for an option the text does not carry."""
    try:
        key, idx = OPTION_STATE[knob]
        parts = dict(p.split("=", 1) for p in text.split("|"))
        vals = parts[key].split(",")
        return ",".join(vals[i] for i in idx)
    except Exception:
        return None


def box_block(dev):
    """What a LONE workgroup sees on this box (vc_box_probe): the sampler at one sequence is one, and its time varied 2x between boxes
    for the same code (DESIGN 4.3 f).  Reported so that the driver's line carries the box's state next to the sampler's time."""
    from voicecraft_amd.engine import box_probe
    try:
        return box_probe(dev)
    except Exception as e:      # reporting only
        return {"error": str(e)}


def sampler_block(wl, box):
    """The sampler launch of the one-sequence step, from its own in-kernel stamps (VC_SAMPLER_TS: chip-wide 100 MHz counter at entry
    and exit of the workgroup, and the mean of its shader-clock phases over every step of one extra call)."""
    import numpy as np
    os.environ["VC_SAMPLER_TS"] = "1"
    try:
        wl.call(4242)
        ts = wl.eng.debug_read("kernel_ts", (64,), dtype=torch.int64).numpy()
    finally:
        os.environ.pop("VC_SAMPLER_TS", None)
    wall = ts[32:34]
    acc = ts[48:59]
    names = ["state_parked_row_in_lds", "edits_argmax", "filter_draw", "cond_sync", "advance_thread0", "next_row_embedding", "store_state"]
    out = {"last_step_in_kernel_us": round(float(wall[1] - wall[0]) / 100.0, 2)}
    if acc[8] > 0:
        mhz = (box or {}).get("shader_mhz_l2_walk") or 0
        out["mean_in_kernel_shader_clk"] = int(acc[7] / acc[8])
        out["mean_phases_shader_clk"] = {n: int(v / acc[8]) for n, v in zip(names, acc[:7])}
        if mhz:
            out["mean_in_kernel_us_at_probe_clock"] = round(float(acc[7] / acc[8]) / mhz, 2)
        out["steps_averaged"] = int(acc[8])
    return out


def options_object(text):
    """The engine's option state (a compact text, vc_debug_read "options") as a JSON object."""
    names = {"g": ("graph_steps", None), "nt": ("nt", ["weights_mask", "attn_kv"]), "fr": ("finished_rows", ["max_rows", "paired", "bf16_partials", "centred_copy"]),
             "ta": ("tile_attn", ["kernel", "min_rows"]), "r1": ("one_row", ["fr_one", "attn_fast", "qkv_p8"]),
             "q16": ("many_rows", ["qkv16", "wide_heads", "wide_gemm", "wd_stage"]), "sh": ("shrink", None)}
    out = {"text": text}
    try:
        for part in text.split("|"):
            key, vs = part.split("=", 1)
            vals = [int(v) for v in vs.split(",")]
            nm, fields = names.get(key, (key, None))
            out[nm] = vals[0] if fields is None and len(vals) == 1 else (dict(zip(fields, vals)) if fields else vals)
    except Exception:       # reporting only
        pass
    return out


def reference_ratio_note(args):
    """The port timed against the UNMODIFIED reference (same windows, same threads) in the build container, where the
    reference tree exists: profiles/cpu_reference_vs_port.json, written by `bench.py --cpu-baseline-only` under
    VC_REFERENCE_ROOT (log: profiles/r04_cpu_reference_vs_port.log)."""
    try:
        with open(os.path.join(ROOT, "profiles", "cpu_reference_vs_port.json")) as f:
            j = json.load(f)
        key = f"{args.preset}/lx{args.lx}/t{args.prompt_frames}/k{args.top_k}"
        r = j.get(key) or j[j["default"]]
        return (f"measured against the unmodified reference on the build box ({r['cores']} threads, {r['key']}): port {r['port']} tok/s, "
                f"reference {r['reference']} tok/s = {r['ratio']}x (profiles/r04_cpu_reference_vs_port.log)")
    except Exception:
        return "not timed against the unmodified reference in this tree"


def cpu_baseline(args, sd, a, x, x_lens, y):
    """The oracle (a port of the reference's CPU path: same ATen ops incl. the per-step KV torch.cat) timed on this
    box's host cores over a bounded sample of the same workload.  The reference's cost per step GROWS with the context
    (O(S) cache copies, BASELINE.md §2: 43 -> 27 tok/s), so THREE windows of the run are sampled - its start, its middle
    and its end (each a call whose prompt is as long as the real run's context at that point) - and the whole run is
    priced by Simpson's rule over the three per-step costs.  When the unmodified reference tree is reachable
    (VC_REFERENCE_ROOT; never on the GPU box) the same windows are timed on it too and both numbers are printed."""
    from oracle.voicecraft_oracle import VoiceCraftOracle
    from voicecraft_amd import synth
    torch.manual_seed(0)
    torch.set_num_threads(max(1, min(args.cpu_threads, os.cpu_count() or 1)))
    K = a.n_codebooks
    n = max(4, args.cpu_steps // 3)
    total_steps = 10 * args.lx - args.prompt_frames + K
    kn = dict(top_k=args.top_k, top_p=1.0, temperature=1.0, stop_repetition=3, kvcache=1)
    windows = [0, max(0, (total_steps - K - n) // 2), max(0, total_steps - K - n)]      # generated frames before each window

    def sample(run):      # run(x, x_lens, y, max_steps, stamps): appends time.perf_counter() to `stamps` once per decode step, where
        per, t_prefill = [], None       # the step's logits exist (so the prompt pass is the time up to the first stamp)
        for w0 in windows:
            yy = y if w0 == 0 else synth.random_prompt(a, args.lx, args.prompt_frames + w0, seed=7)[2]
            stamps = []
            t0 = time.perf_counter()
            run(x, x_lens, yy, n + 1, stamps)
            per.append((stamps[-1] - stamps[0]) / max(1, len(stamps) - 1))
            if w0 == 0:
                t_prefill = stamps[0] - t0
        est = t_prefill + total_steps * (per[0] + 4 * per[1] + per[2]) / 6.0
        return t_prefill, per, est

    class _Stamped(list):     # the oracle appends one trace entry per step, right after the heads
        def __init__(self, stamps):
            super().__init__()
            self.stamps = stamps

        def append(self, item):
            self.stamps.append(time.perf_counter())
            super().append(item)

    orc = VoiceCraftOracle(a, sd)
    t_prefill, per, est = sample(lambda xx, xl, yy, ms, st: orc.inference_tts(xx, xl, yy, max_steps=ms, trace=_Stamped(st), **kn))
    tokens = K * (10 * args.lx - args.prompt_frames)
    out = {
        "value": round(tokens / est, 2), "unit": "codec-tokens/s", "cores": torch.get_num_threads(), "kind": "port",
        "prefill_s": round(t_prefill, 2), "ms_per_step": [round(v * 1e3) for v in per], "whole_run_s": round(est, 1),
        "sample": (f"{args.preset} fp32, Lx={args.lx}, {args.prompt_frames} prompt frames: prefill {t_prefill:.2f} s; {n} decode steps at the "
                   f"start / middle / end of the run's context ({', '.join(str(args.lx + args.prompt_frames + 1 + w) for w in windows)} positions): "
                   f"{per[0] * 1e3:.0f} / {per[1] * 1e3:.0f} / {per[2] * 1e3:.0f} ms per step (stamped inside one call each); whole run of {total_steps} steps by Simpson's rule = {est:.1f} s. "
                   "The oracle is a port of the reference's CPU path; " + reference_ratio_note(args) +
                   "; the reference itself measured 27.1 tok/s on 8 cores (BASELINE.md §2)"),
    }
    ref_root = os.environ.get("VC_REFERENCE_ROOT", "")
    if ref_root and os.path.isfile(os.path.join(ref_root, "models", "voicecraft.py")):
        try:       # the unmodified reference on the same windows (a bounded run: its loop is cut by a step-counting sampler hook)
            from oracle import ref_loader
            model = ref_loader.build_reference_model(a, sd)
            vc, _ = ref_loader.import_reference()

            class _Stop(Exception):
                pass

            def run_ref(xx, xl, yy, max_steps, stamps):
                calls, orig = [0], vc.topk_sampling

                def counted(*aa, **kk):
                    stamps.append(time.perf_counter())
                    calls[0] += 1
                    if calls[0] >= max_steps:
                        raise _Stop()
                    return orig(*aa, **kk)
                vc.topk_sampling = counted
                try:
                    with torch.no_grad():
                        model.inference_tts(xx, xl, yy, silence_tokens=[1388, 1898, 131], **kn)
                except _Stop:
                    pass
                finally:
                    vc.topk_sampling = orig
            rpre, rper, rest = sample(run_ref)
            out["reference"] = {"value": round(tokens / rest, 2), "kind": "reference", "prefill_s": round(rpre, 2),
                                "ms_per_step": [round(v * 1e3) for v in rper], "whole_run_s": round(rest, 1),
                                "port_speed_over_reference": round((tokens / est) / (tokens / rest), 3)}
        except Exception as e:      # reporting only
            out["reference"] = {"error": str(e)}
    return out


def conv_flops_encode(n_samples, F=64, ratios=(8, 5, 4, 2), hidden=128, k=7, rk=3):
    """FLOPs (2 per multiply-add) of the SEANet encoder's convolutions for one clip at the VoiceCraft codec shape
    (the restatement's layer list: first conv, per stage one residual unit [k=3 conv to dim/2, 1x1 conv back] and the
    strided down-sampling conv (kernel 2 x stride), final conv; the LSTM is priced separately)."""
    L, ch, fl = n_samples, F, 0.0
    fl += 2.0 * L * 1 * F * k
    for r in reversed(ratios):
        fl += 2.0 * L * ch * (ch // 2) * rk + 2.0 * L * (ch // 2) * ch
        Lo = -(-L // r)
        fl += 2.0 * Lo * ch * (2 * ch) * (2 * r)
        L, ch = Lo, 2 * ch
    fl += 2.0 * L * ch * hidden * k
    return fl, L, ch


def codec_block(dev):
    """EnCodec encode / decode of 16 s of audio (synthetic weights at the VoiceCraft codec shape, fp32 MFMA), SURVEY §8d
    'report codec encode/decode separately'.  The implicit-GEMM convolutions (conv_gemm_k, fp32 MFMA) carry an MFMA roofline
    object; the LSTM recurrence is reported as what bounds it (persistent launch: hand-off latency per step)."""
    from voicecraft_amd import synth
    from voicecraft_amd.codec import AudioTokenizer
    tok = AudioTokenizer(synth.make_codec_state_dict(0), device=dev, max_seconds=16.5, max_batch=1)
    torch.manual_seed(0)
    n = 16 * 16000
    wav = (torch.randn(1, 1, n) * 0.1).to(dev)
    codes = tok.encode(wav)[0][0]
    enc, lst = [], []
    for _ in range(3):
        tok.encode(wav)
        enc.append(tok.last_ms())
        lst.append(tok.last_lstm_ms())
    i = min(range(3), key=lambda j: enc[j])
    lstm_ms, lstm_bytes = lst[i]
    T = int(codes.shape[2])
    dec = []
    for _ in range(3):
        tok.decode([(codes, None)])
        dec.append(tok.last_ms())
    e_ms, d_ms = enc[i], min(dec)
    step_us = lstm_ms * 1e3 / max(1, T)
    persistent = os.environ.get("VC_LSTM_WAVE") is None
    conv_fl, _, ch = conv_flops_encode(n)
    conv_fl += 2.0 * T * ch * 4 * ch            # the LSTM's layer-0 input projection runs on conv_gemm_k as a 1x1 convolution
    conv_ms = max(1e-6, e_ms - lstm_ms)
    return {"audio_s": 16.0, "frames": T, "encode_ms": round(e_ms, 2), "decode_ms": round(d_ms, 2),
            "rtf_encode": round(e_ms / 16e3, 6), "rtf_decode": round(d_ms / 16e3, 6), "dtype": "f32", "parity": "unpinned (audiocraft not vendored)",
            # the persistent LSTM keeps its weights in registers: nothing is streamed per step, so there is no HBM roofline to
            # quote - a step IS one all-to-all hand-off of the hidden vector between the workgroups, and that latency is the bound
            "lstm": ({"kernel": "lstm_persist_k (ONE persistent launch for the whole sequence, both layers)", "bound": "hand-off latency",
                      "us_per_recurrence_step": round(step_us, 2), "steps": T,
                      "note": "one {value, epoch} granule hand-off round per step; the launch-per-step form re-reads 50 MB of weights per step (8.8 us)"}
                     if persistent else
                     {"kernel": "lstm_wave_k (one launch per recurrence step of both LSTM layers)", "bound": "hbm",
                      "us_per_recurrence_step": round(step_us, 2), "steps": T, "bytes_per_launch": lstm_bytes,
                      "achieved": round(lstm_bytes / (step_us * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": round(lstm_bytes / (step_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None}),
            "conv_roofline": {"bound": "mfma", "kernel": "conv_gemm_k (every implicit-GEMM convolution of one encode, fp32 MFMA)",
                              "achieved": round(conv_fl / (conv_ms * 1e-3) / 1e12, 2), "peak": MFMA_PEAK_TF["fp32"], "unit": "TFLOP/s",
                              "frac": round(conv_fl / (conv_ms * 1e-3) / 1e12 / MFMA_PEAK_TF["fp32"], 4), "traffic": None,
                              "flops": conv_fl, "ms": round(conv_ms, 2),
                              "note": "encode wall minus the LSTM launch; includes the first conv, the RVQ search and ~30 launch boundaries"}}


def one_sample_block(eng, a, dev, args):
    """One whole TTS request on the bench's model (what inference_tts_scale.py:42-105 does between its phonemizer and its file
    writer): encode a 3 s synthetic voice prompt, generate to the reference's length cap, decode prompt + generated and the
    generated part.  The model's special-token logits are NOT muted here (the bench checkpoint only mutes the terminator), so
    generated frames may hold a special id once in a while; they are mapped to code 0 before the codec, which times the same work."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from one_sample_chain import OneSampleChain
    from voicecraft_amd import synth
    from voicecraft_amd.codec import AudioTokenizer

    class _Codes0(AudioTokenizer):
        def decode(self, frames):
            return super().decode([(frames[0][0].clamp(max=2047), None)])

    tok = _Codes0(synth.make_codec_state_dict(0), device=dev, max_seconds=17.0, max_batch=1)
    torch.manual_seed(0)
    wav = torch.randn(args.prompt_frames * 320) * 0.1
    text = synth.random_prompt(a, args.lx, 1, seed=1)[0][0]
    cfg = dict(top_k=args.top_k, top_p=1.0, temperature=1.0, stop_repetition=3, kvcache=1, codec_sr=50,
               silence_tokens=[1388, 1898, 131], sample_batch_size=1)
    chain = OneSampleChain(eng, tok, a.n_codebooks, dev)
    best = None
    for _ in range(3):
        r = chain.run(text, wav, cfg)
        if best is None or r.seconds["total"] < best.seconds["total"]:
            best = r
    gen_frames = int(best.new_codes.shape[-1])
    gen_s, sec = gen_frames / 50.0, best.seconds
    return {"prompt_frames": int(best.prompt_codes.shape[1]), "gen_frames": gen_frames, "encode_ms": round(sec["encode"] * 1e3, 2),
            "model_ms": round(sec["model"] * 1e3, 2), "decode_concat_and_gen_ms": round(sec["decode"] * 1e3, 2),
            "total_ms": round(sec["total"] * 1e3, 2), "rtf_end_to_end": round(sec["total"] / gen_s, 4),
            "rtf_model_only": round(sec["model"] / gen_s, 4),
            "codec_tokens_per_sec_end_to_end": round(4 * gen_frames / sec["total"], 1)}


def step_alg_bytes(a, dtype, B, s_mean):
    """Algorithmic HBM bytes of one decode step: every weight once + K/V read and write of each of the B sequences."""
    d, L, K = a.d_model, a.num_decoder_layers, a.n_codebooks
    V, P = a.audio_vocab_size + a.n_special, a.audio_vocab_size // 2
    esz = 2 if dtype == "bf16" else 4
    w_bytes = esz * (L * (12 * d * d + 13 * d) + 2 * d + K * (d * P + P + P * V + V))
    return w_bytes + B * esz * 2 * L * d * (s_mean + 1)


class Workload:
    """One benchmark workload on its own engine: BASELINE configs 1-5 differ only in these arguments.  `call(seed)` is one whole
    `inference_tts` / `inference_tts_multi` / `inference` call (prompt build + prefill + every decode step + un-shift) and returns
    the generated frames of every utterance."""

    def __init__(self, preset, mode, batch, lx, prompt_frames, top_k, dtype, dev, use_graph=True, rank=0, world=1, sd=None, best_of=1, lx_min=None):
        from voicecraft_amd import synth
        from voicecraft_amd.engine import VoiceCraftEngine
        self.preset, self.mode, self.B, self.lx, self.top_k, self.dtype, self.dev = preset, mode, batch, lx, top_k, dtype, dev
        self.a = a = synth.make_args(preset)
        self.K = K = a.n_codebooks
        self.sd = sd if sd is not None else synth.make_state_dict(a, seed=0, perturb=False, mute_eos=True, fast=True)
        self.edit = mode == "edit"
        self.best_of = int(best_of)      # > 1: inference_tts_batch(batch_size = best_of) of ONE utterance (models/voicecraft.py:1156)
        assert self.best_of == 1 or (batch == 1 and mode == "tts")
        self.span = None
        if self.edit:      # 16 s utterance (10 frames per phoneme), the middle eighth masked: generation runs to the
            assert batch == 1, "editing is single-utterance (models/voicecraft.py:607)"      # reference's length cap y_len > 10*Lx
            prompt_frames = 10 * lx
            self.span = (prompt_frames * 3 // 8, prompt_frames * 4 // 8)        # [300,400) of 800 frames
        self.prompt_frames = prompt_frames
        span = self.span
        # generated frames: TTS 10 Lx - T (the length cap); editing: the cap minus the rearranged prompt's columns
        # (two shifted pieces of K extra columns each, two mask placeholders, the end token and the start column)
        self.Tg = 10 * lx - prompt_frames if not self.edit else 10 * lx - (prompt_frames - (span[1] - span[0]) + 2 * K + 4) + 1
        self.eng = VoiceCraftEngine(a, self.sd, device=dev, dtype=dtype, max_seqs=max(1, batch, self.best_of),
                                    max_positions=max(1024, lx + prompt_frames + self.Tg + 64), use_graph=use_graph)
        # utterance u of this rank = global utterance (u * world + rank); seed = 1 + global index
        # lx_min: a RAGGED batch - utterance u has lx_min .. lx phonemes (evenly spread), so the reference's length cap (10 frames per
        # phoneme) ends the sequences at different steps: generated lengths spread 10 lx_min - T .. 10 lx - T
        self.lxs = [lx if (lx_min is None or batch == 1) else lx_min + (lx - lx_min) * u // (batch - 1) for u in range(batch)]
        self.ragged = lx_min is not None and batch > 1
        self.prompts = [synth.random_prompt(a, self.lxs[u], prompt_frames, seed=1 + (u * world + rank)) for u in range(batch)]
        self.xs = [p[0].to(dev) for p in self.prompts]
        self.xls = [p[1].to(dev) for p in self.prompts]
        self.ys = [p[2].to(dev) for p in self.prompts]
        self.knobs = dict(top_k=top_k, top_p=1.0, temperature=1.0, stop_repetition=3)

    def call(self, seed):
        eng, K = self.eng, self.K
        if self.edit:
            mi = torch.tensor([[list(self.span)]], dtype=torch.int64)
            res = eng.inference(self.xs[0], self.xls[0], self.ys[0], mi, silence_tokens=[1388, 1898, 131], _seed=seed, **self.knobs)
            return [None], (int(res.shape[2]) - (self.prompt_frames - (self.span[1] - self.span[0]))) * K
        if self.best_of > 1:      # the reference's front-ends run this mode (gradio_app.py:506 sample_batch_size = 3): the kept sample's frames count
            res, gen = eng.inference_tts_batch(self.xs[0], self.xls[0], self.ys[0], kvcache=1, batch_size=self.best_of,
                                               silence_tokens=[1388, 1898, 131], _seed=seed, **self.knobs)
            gens = [gen]
        elif self.B == 1:
            res, gen = eng.inference_tts(self.xs[0], self.xls[0], self.ys[0], kvcache=1, silence_tokens=[1388, 1898, 131], _seed=seed, **self.knobs)
            gens = [gen]
        else:
            outs = eng.inference_tts_multi([x[0] for x in self.xs], [y[0] for y in self.ys], silence_tokens=[1388, 1898, 131], _seed=seed, **self.knobs)
            gens = [o[1] for o in outs]
        return gens, sum(int(g.shape[2]) * K for g in gens)

    def label(self, use_graph=True):
        if self.edit:
            return (f"{self.preset} speech editing, Lx={self.lx}, {self.prompt_frames}-frame utterance, span "
                    f"{self.span} re-generated (~{self.Tg} frames, reference length cap), top_k={self.top_k}")
        if self.best_of > 1:
            return (f"{self.preset} TTS best-of-{self.best_of} (inference_tts_batch: {self.best_of} samples of one utterance decoded together, one kept), Lx={self.lx}, "
                    f"{self.prompt_frames} prompt frames -> {self.Tg} generated frames, top_k={self.top_k}, hipGraph={'on' if use_graph else 'off'}")
        if self.ragged:
            return (f"{self.preset} TTS, RAGGED batch {self.B}/GPU, Lx={self.lxs[0]}..{self.lxs[-1]}, {self.prompt_frames} prompt frames -> "
                    f"{10 * self.lxs[0] - self.prompt_frames}..{10 * self.lxs[-1] - self.prompt_frames} generated frames per utterance (the reference's length cap), "
                    f"top_k={self.top_k}, hipGraph={'on' if use_graph else 'off'}")
        return (f"{self.preset} TTS, batch {self.B}/GPU, Lx={self.lx}, {self.prompt_frames} prompt frames -> "
                f"{self.Tg} generated frames ({(self.prompt_frames + self.Tg) // 50} s total), top_k={self.top_k}, hipGraph={'on' if use_graph else 'off'}")

    def s_mean(self):
        return self.lx + self.prompt_frames + 1 + self.Tg / 2


def configs_block(args, dev, sd830):
    """The OTHER BASELINE configurations, each on its own engine after the headline's timed region: 1 warm-up + 2 timed whole calls
    (host wall around each call, device idle on both sides), so that the driver's single bench line shows every configuration it names
    (missing #3).  C3 is the headline itself; C5 is quoted as its per-GPU share (8 utterances of the 64)."""
    specs = [
        ("C2", "BASELINE config 2", dict(preset="giga330M", mode="tts", batch=1, lx=80, prompt_frames=150, top_k=40)),
        ("C1", "BASELINE config 1's workload on the GPU (the config itself is the CPU reference run: cpu_baseline / BASELINE.md section 2)",
         dict(preset="giga330M", mode="tts", batch=1, lx=40, prompt_frames=150, top_k=1)),
        ("C4", "BASELINE config 4", dict(preset="giga830M", mode="edit", batch=1, lx=80, prompt_frames=150, top_k=40)),
        ("C5_per_gpu_share", "BASELINE config 5: 64 utterances over 8 GPUs = 8 per GPU, this GPU's share",
         dict(preset="giga830M", mode="tts", batch=8, lx=80, prompt_frames=150, top_k=40)),
        ("C5_all_64_on_one_gpu", "BASELINE config 5's 64 utterances decoded together on ONE GPU (64-row steps: the wide-decode kernels of round 6)",
         dict(preset="giga830M", mode="tts", batch=64, lx=80, prompt_frames=150, top_k=40)),
        ("utterances_32_per_gpu", "32 utterances per GPU (32-row steps)", dict(preset="giga830M", mode="tts", batch=32, lx=80, prompt_frames=150, top_k=40)),
        ("C3_best_of_3", "BASELINE config 3's utterance through inference_tts_batch(batch_size=3), the mode the reference's front-ends run "
         "(gradio_app.py:506, inference_tts.ipynb): three samples decoded together, the kept one's frames counted",
         dict(preset="giga830M", mode="tts", batch=1, lx=80, prompt_frames=150, top_k=40, best_of=3)),
    ]
    out = {}
    sd_cache = {"giga830M": sd830} if sd830 is not None else {}
    for key, what, kw in specs:
        try:
            wl = Workload(dtype=args.dtype, dev=dev, use_graph=not args.no_graph, sd=sd_cache.get(kw["preset"]), **kw)
            sd_cache[kw["preset"]] = wl.sd
            wl.call(100)
            torch.cuda.synchronize()
            tok = 0
            wall = dec = pre = 0.0
            steps = 0
            n_calls = 2
            for i in range(n_calls):
                t0 = time.perf_counter()
                tok += wl.call(1000 + i)[1]
                torch.cuda.synchronize()
                wall += time.perf_counter() - t0
                tm = wl.eng.last_timing_ms()
                dec += tm["decode_ms"]; pre += tm["prefill_ms"]; steps += wl.eng.last_steps
            dstep = dec / max(1, steps)
            sb = step_alg_bytes(wl.a, args.dtype, max(wl.B, wl.best_of), wl.s_mean())
            ab = None
            rows = max(wl.B, wl.best_of)
            if wl.B > 16:      # the wide-decode kernels of round 6 against the weight-stationary kernel of rounds 2-5, in process (3 pairs)
                try:
                    ab = ab_block(wl.eng, lambda seed: wl.call(seed), "wide_gemm=0:1", 3)
                    ab = {k: ab[k] for k in ("knob", "A", "B", "A_ms_median", "B_ms_median", "median_delta_pct", "spread_pct")}
                except Exception as e:      # reporting only
                    ab = {"error": str(e)}
            elif 2 <= rows <= 16:      # the byte-halving forms of round 6, each against its off state, in process (3 pairs); up to 6 rows also the paired QKV consumer
                ab = []
                for spec in ("att_p16=0:1", "hq=0:1") + (("qkv_p8=1:2",) if rows <= 6 else ()):
                    try:
                        r = ab_block(wl.eng, lambda seed: wl.call(seed), spec, 3)
                        ab.append({k: r[k] for k in ("knob", "A", "B", "A_ms_median", "B_ms_median", "median_delta_pct", "spread_pct")})
                    except Exception as e:      # reporting only
                        ab.append({"knob": spec, "error": str(e)})
            out[key] = {"config": what, "workload": wl.label(not args.no_graph), "value": round(tok / wall, 1), "unit": "codec-tokens/s",
                        "calls": n_calls, "ms_per_call": round(wall / n_calls * 1e3, 2), "prefill_ms": round(pre / n_calls, 2),
                        "decode_ms_per_step": round(dstep, 4), "rtf": round(wall / (tok / wl.K / 50.0), 4),
                        "hbm_frac_in_loop": round(sb / (dstep * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if dstep > 0 else None}
            if ab is not None:
                out[key]["ab"] = ab
            del wl
        except Exception as e:      # reporting only: never lose the headline to it
            out[key] = {"config": what, "error": str(e)}
    return out


def ragged_block(args, dev, sd830):
    """Utterances of DIFFERENT lengths decoded together (ragged T_g is the one source of imbalance inside a batch): Lx spread
    2x, so the sequences retire between 250 and 650 generated frames.  Timed twice on one engine: the batch re-packed onto narrower
    steps as sequences retire (option shrink = 1, the default) and at its fixed starting width (shrink = 0, rounds 1-5)."""
    out = {}
    for B in (8, 64):
        key = f"ragged_{B}"
        try:
            wl = Workload("giga830M", "tts", B, 80, 150, 40, args.dtype, dev, use_graph=not args.no_graph, sd=sd830, lx_min=40)
            r = {"workload": wl.label(not args.no_graph)}
            legs = (("shrink", 1), ("fixed_width", 0))
            for name, val in legs:            # both states' graphs captured and warm before anything is timed
                wl.eng.set_option("shrink", val)
                wl.call(100)
            torch.cuda.synchronize()
            acc = {name: [0, 0.0, 0.0, 0, 0] for name, _ in legs}       # tokens, wall, decode ms, steps, re-packs
            for i in range(2):                # interleaved pairs (A B, B A): drift between the legs cancels, as in ab_block
                for name, val in (legs if i % 2 == 0 else legs[::-1]):
                    wl.eng.set_option("shrink", val)
                    t0 = time.perf_counter()
                    tok = wl.call(1000 + i)[1]
                    torch.cuda.synchronize()
                    a_ = acc[name]
                    a_[0] += tok; a_[1] += time.perf_counter() - t0
                    a_[2] += wl.eng.last_timing_ms()["decode_ms"]; a_[3] += wl.eng.last_steps
                    a_[4] += int(wl.eng.debug_read("host_ms", (8,), torch.float64)[6])
            for name, _ in legs:
                tok, wall, dec, steps, repacks = acc[name]
                r[name] = {"value": round(tok / wall, 1), "unit": "codec-tokens/s", "ms_per_call": round(wall / 2 * 1e3, 2),
                           "decode_ms": round(dec / 2, 2), "steps": steps // 2, "repacks_per_call": repacks // 2}
            wl.eng.set_option("shrink", 1)
            r["gain_pct"] = round(100.0 * (r["shrink"]["value"] / r["fixed_width"]["value"] - 1.0), 2)
            out[key] = r
            del wl
        except Exception as e:      # reporting only
            out[key] = {"error": str(e)}
    return out


def ab_block(eng, one_step, spec, pairs):
    """In-process A/B of ONE engine option (vc_set_option): the same engine, the same process, the same box.  Whole calls
    are timed in interleaved pairs (A B, B A, A B ...: drift cancels), each arm's captured graph already warm; the figure
    compared is the device-timed decode loop per step taken.  `median_delta_pct` = median over pairs of (B - A) / A;
    `spread_pct` = 1.4826 x the median absolute deviation of the per-pair deltas (the noise one pair carries; `half_range_pct` =
    half their range, which a single disturbed call inflates).  An option only earns a default when |median_delta_pct| >
    spread_pct in a driver-run line."""
    knob, vals = spec.split("=", 1)
    va, vb = vals.split(":", 1)
    prior = option_value(eng.options(), knob)      # the value in force before the A/B (an env preset, or the default): restored afterwards

    def timed(v, seed):
        eng.set_option(knob, v)
        one_step(seed)
        return eng.last_timing_ms()["decode_ms"] / max(1, eng.last_steps)
    for v in (va, vb, va, vb):        # capture + warm both states
        timed(v, 7)
    a_ms, b_ms, deltas = [], [], []
    for i in range(pairs):
        order = (va, vb) if i % 2 == 0 else (vb, va)
        t = {v: timed(v, 2000 + i) for v in order}
        a_ms.append(t[va]); b_ms.append(t[vb])
        deltas.append((t[vb] - t[va]) / t[va] * 100.0)
    eng.set_option(knob, prior if prior is not None else vb)
    med = lambda xs: sorted(xs)[len(xs) // 2]
    m = med(deltas)
    mad = 1.4826 * med([abs(d - m) for d in deltas])          # robust sigma of one pair's delta: a call hit by a host hiccup does not define it
    half_range = (max(deltas) - min(deltas)) / 2.0
    return {"knob": knob, "A": va, "B": vb, "pairs": pairs, "metric": "decode ms per step taken (device events around the loop)",
            "A_ms_median": round(med(a_ms), 5), "B_ms_median": round(med(b_ms), 5),
            "median_delta_pct": round(m, 3), "spread_pct": round(mad, 3), "half_range_pct": round(half_range, 3),
            "deltas_pct": [round(d, 3) for d in deltas],
            "verdict": ("B faster" if m < 0 else "B slower") + (" beyond the spread" if abs(m) > max(mad, 0.05) else " (inside the spread: not shown)")}


def cpu_only(args):
    """`--cpu-baseline-only`: the CPU leg alone (no GPU, no engine).  With VC_REFERENCE_ROOT set (build container) the
    unmodified reference is timed on the same windows; the measured ratio is also merged into
    profiles/cpu_reference_vs_port.json, which the GPU box's bench line quotes."""
    from voicecraft_amd import synth
    a = synth.make_args(args.preset)
    sd = synth.make_state_dict(a, seed=0, perturb=False, mute_eos=True, fast=True)
    x, xl, y = synth.random_prompt(a, args.lx, args.prompt_frames, seed=1)
    out = cpu_baseline(args, sd, a, x, xl, y)
    key = f"{args.preset}/lx{args.lx}/t{args.prompt_frames}/k{args.top_k}"
    print(json.dumps({"config": key, "cpu_baseline": out}), flush=True)
    ref = out.get("reference") or {}
    if ref.get("value"):
        path = os.path.join(ROOT, "profiles", "cpu_reference_vs_port.json")
        try:
            with open(path) as f:
                j = json.load(f)
        except Exception:
            j = {}
        j[key] = {"key": key, "cores": out["cores"], "port": out["value"], "reference": ref["value"],
                  "ratio": ref["port_speed_over_reference"], "port_ms_per_step": out["ms_per_step"], "reference_ms_per_step": ref["ms_per_step"]}
        j.setdefault("default", key)
        with open(path, "w") as f:
            json.dump(j, f, indent=1, sort_keys=True)


def spawn_command(gpus, argv, port):
    """`python bench.py --gpus N` outside a launcher: the command that starts the N ranks (one process per GPU, rendezvous on
    127.0.0.1) with this very argument list - the form the driver itself uses for N > 1."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.join(ROOT, "bench.py"), *argv]


def spawn_ranks(args):
    """--gpus N > 1 with no WORLD_SIZE in the environment: start the N ranks here (a plain `python bench.py --gpus 8`
    used to measure ONE GPU and print n_gpus 1).  Fails loudly when the box shows fewer than N devices - unless
    VC_RANKS_SHARE_DEVICE=1 (every rank drives cuda:0: the single-GPU test of the N > 1 path, which needs --dist-backend gloo)."""
    import socket
    import subprocess
    share = os.environ.get("VC_RANKS_SHARE_DEVICE", "0") == "1"
    have = torch.cuda.device_count()
    if not share and have < args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible (set VC_RANKS_SHARE_DEVICE=1 and "
                 "--dist-backend gloo to run every rank on cuda:0)")
    if share and args.dist_backend == "nccl":
        sys.exit("bench.py: VC_RANKS_SHARE_DEVICE=1 needs --dist-backend gloo (RCCL refuses two ranks on one device)")
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    return subprocess.call(spawn_command(args.gpus, sys.argv[1:], port), env=env, cwd=ROOT)


def main():
    args = parse()
    if args.cpu_baseline_only:
        return cpu_only(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # VC_RANKS_SHARE_DEVICE=1: every rank drives cuda:0 (the single-GPU test of this N > 1 code path; RCCL refuses two ranks on
    # one device, so that run takes --dist-backend gloo).  The job is otherwise identical: sharding, barriers, reductions, gather.
    share = os.environ.get("VC_RANKS_SHARE_DEVICE", "0") == "1"
    dev = torch.device("cuda", 0 if share else local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            assert not share, "RCCL needs one GPU per rank: use --dist-backend gloo with VC_RANKS_SHARE_DEVICE=1"
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    else:
        dist = None
    cdev = dev if (dist is None or args.dist_backend == "nccl") else torch.device("cpu")      # where the reductions' tensors live
    n_gpus = world
    assert args.gpus == n_gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from voicecraft_amd import dist as vdist
    box = box_block(dev) if (rank == 0 and world == 1) else None
    wl = Workload(args.preset, args.mode, args.batch, args.lx, args.prompt_frames, args.top_k, args.dtype, dev,
                  use_graph=not args.no_graph, rank=rank, world=world)
    a, sd, eng, K, B, edit, span, Tg, prompts = wl.a, wl.sd, wl.eng, wl.K, wl.B, wl.edit, wl.span, wl.Tg, wl.prompts
    args.prompt_frames = wl.prompt_frames          # (editing: the whole 16 s utterance is the prompt)

    def one_step(seed):
        gens, n_tok = wl.call(seed)
        if not edit and (dist is not None or args.dump):   # the single collective of the job: gather every rank's token block
            tg0 = time.perf_counter()
            gathered[:] = vdist.gather_token_blocks([g[0] for g in gens], Tg + 8, n_slots=B, K=K, device=dev)
            gather_s[0] += time.perf_counter() - tg0      # host wall incl. the .to(int32) staging and the unpacking (it synchronises)
        return n_tok

    gathered = []
    gather_s = [0.0]

    for w in range(args.warmup):
        one_step(100 + w)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tokens = 0
    dec_ms = pre_ms = 0.0
    steps_run = steps_launched = 0
    gather_s[0] = 0.0
    for s in range(args.steps):
        tokens += one_step(1000 + s)
        tm = eng.last_timing_ms()
        dec_ms += tm["decode_ms"]; pre_ms += tm["prefill_ms"]; steps_run += eng.last_steps
        steps_launched += int(eng.debug_read("host_ms", (8,), torch.float64)[5])
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        tot = torch.tensor([tokens], dtype=torch.int64, device=cdev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        tokens = int(tot.item())

    out = None
    if rank == 0 and args.dump and gathered:
        import numpy as np
        merged = vdist.merge_in_utterance_order(gathered)
        np.savez(args.dump, **{f"u{u}": m.cpu().numpy() for u, m in enumerate(merged)})
    if rank == 0:
        frames = tokens / K
        value = tokens / dt
        step_bytes = step_alg_bytes(a, args.dtype, B, wl.s_mean())      # weights (once per step) + KV read + KV write of each of the B sequences
        # dominant kernel = the largest share of a step's kernel time.  Every decode kernel runs once per layer, so it is the one with the
        # longest launch: each is timed in isolation here (layers rotate: cold caches), the longest becomes the `roofline` object - the
        # FFN up-projection since round 5 (profiles/r05d_rocprof_kernel_stats.txt: 23.4 % against 20.4 % for the down-projection, which
        # was the dominant one through round 4), and `in_situ` carries the committed rocprof average of the same kernel
        mb_rows = min(B, 16)       # the kernel microbenchmarks drive the <=16-row decode kernels
        c0 = eng.launch_counts()
        step_ms, _ = eng.bench_kernel("step", n_rows=mb_rows, iters=8)
        kernels, kraw = {}, {}
        for kn in ("qkv", "attn", "oproj", "ffn1", "ffn2", "qkv_hot", "oproj_hot", "ffn1_hot", "ffn2_hot"):
            # (a short untimed run first, then the better of two: the first microbenchmark after the step used to catch the chip in
            # transition - the same kernel read 6.3 / 6.7 / 7.8 us on three boxes while its in-situ average stayed at 6.4-6.5)
            eng.bench_kernel(kn, n_rows=mb_rows, iters=8)
            ms_, by_ = min(eng.bench_kernel(kn, n_rows=mb_rows, iters=64), eng.bench_kernel(kn, n_rows=mb_rows, iters=64))
            kraw[kn] = (ms_, by_)
        for kn in list(kraw):      # a second sweep, the better of the two kept: the first kernel timed after the whole-step microbenchmark has
            ms_, by_ = min(kraw[kn], eng.bench_kernel(kn, n_rows=mb_rows, iters=64))      # read 7.4-8.8 us for a launch that takes 6.5 in situ (r06b / r06d)
            kraw[kn] = (ms_, by_)
            kernels[kn] = {"avg_us": round(ms_ * 1e3, 2), "GB/s": round(by_ / (ms_ * 1e-3) / 1e9, 1)}
        c1 = eng.launch_counts()
        fr1_form = c1["row_gemm_fr1"] > c0["row_gemm_fr1"]      # which forms the microbenchmarks (= the step) really launched
        fr_form = c1["rows_gemm_fr"] > c0["rows_gemm_fr"]
        # ... decided by the committed in-situ trace of this very configuration where there is one (what a reader of profiles/ computes),
        # by the isolated times otherwise
        ins = {k: in_situ(k, args) for k in ("ffn1", "ffn2", "qkv")}
        if all(v and v.get("avg_us") for v in ins.values()):
            dom = max(ins, key=lambda k: ins[k]["avg_us"])
        else:
            dom = max(("ffn1", "ffn2", "qkv"), key=lambda k: kraw[k][0])
        k_ms, k_bytes = kraw[dom]
        what = {"ffn1": "FFN up-projection", "ffn2": "FFN down-projection", "qkv": "QKV projection"}[dom]
        if fr_form:
            form = {"ffn2": "rows_gemm_fr_k<plain> (finished rows: 8-channel tiles over the whole K)", "ffn1": "rows_gemm_k<LayerNorm fold of finished rows, ReLU>",
                    "qkv": "rows_gemm_k<LayerNorm fold of finished rows, QKV>"}[dom] + ("" if B <= 16 else f"; microbenchmarked at 16 rows - this run's {B}-row steps use the wide-decode kernel rows_gemm_mt_k")
        elif fr1_form:
            form = {"ffn2": "row_gemm_fr1_k<plain, residual> (finished row: 8-channel tiles over the whole K, two k-tiles per MFMA fragment)",
                    "ffn1": "rows_gemm_k<LayerNorm fold of h + 2 slabs, ReLU> (16-channel tiles)",
                    "qkv": "row_gemm_fr1_k<LayerNorm fold, QKV> (8-channel tiles, two k-tiles per MFMA fragment)"}[dom]
        else:
            form = {"ffn2": "rows_gemm_k<plain, split-K slabs>", "ffn1": "rows_gemm_k<LayerNorm fold, ReLU>", "qkv": "rows_gemm_k<LayerNorm fold, QKV>"}[dom]
        # `achieved` / `frac`: the IN-SITU figure (every launch of the decode loop under rocprofv3, caches as the step leaves them) when the
        # committed trace belongs to this configuration and this build; otherwise the isolated microbenchmark of this run, which comes out
        # ~3-5 % higher (a quiet chip) and always rides along as `isolated_*`
        iso_gbs = k_bytes / (k_ms * 1e-3) / 1e9
        ins_dom = in_situ(dom, args)
        tr = pmc_traffic(dom, args)
        if ins_dom and ins_dom.get("frac"):
            ach_us, measured = ins_dom["avg_us"], f"in situ: rocprofv3 --kernel-trace average over {ins_dom['calls']} launches of the decode loop, committed trace of this build ({ins_dom['source']})"
        else:
            ach_us, measured = k_ms * 1e3, "isolated: the kernel back to back over rotating layers, HIP events on the launch stream (vc_bench_kernel); no in-situ trace of this build and configuration is on file"
        ach_gbs = k_bytes / (ach_us * 1e-6) / 1e9
        roof = {"bound": "hbm", "kernel": f"{form} ({what}; the longest of the step's per-layer launches)",
                "achieved": round(ach_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach_gbs / HBM_PEAK_GBS, 4), "traffic": tr["bytes"] if tr else None,
                "traffic_source": ({k: tr[k] for k in ("source", "box", "lib_stamp")} if tr else None),
                "bytes_per_launch": k_bytes, "avg_launch_us": round(ach_us, 2),
                "measured": measured,
                "isolated_frac": round(iso_gbs / HBM_PEAK_GBS, 4), "isolated_avg_launch_us": round(k_ms * 1e3, 2), "isolated_achieved": round(iso_gbs, 1),
                "in_situ": ins_dom, "lib_stamp": lib_stamp(),
                "per_layer_launches": {k: {"isolated_us": kernels[k]["avg_us"], "in_situ": in_situ(k, args)} for k in ("qkv", "attn", "oproj", "ffn1", "ffn2")}}
        # the prefill is GEMM-shaped: MFMA rooflines of its widest block GEMM (FFN up-projection) at the run's own pass
        # size and at a full 512-row pass, and of the MFMA tile attention
        own_rows = min(512, B * ((args.lx + (args.prompt_frames + 1 if not edit else args.prompt_frames - (span[1] - span[0]) + 2 * K + 4) + 15) // 16 * 16))
        pk = MFMA_PEAK_TF[args.dtype]

        def mfma_obj(which, rows, label):
            ms_, fl_ = eng.bench_kernel(which, n_rows=rows, iters=32)
            return {"bound": "mfma", "kernel": label, "achieved": round(fl_ / (ms_ * 1e-3) / 1e12, 1), "peak": pk, "unit": "TFLOP/s",
                    "frac": round(fl_ / (ms_ * 1e-3) / 1e12 / pk, 4), "traffic": None, "flops_per_launch": fl_, "avg_launch_us": round(ms_ * 1e3, 2)}
        mfma = mfma_obj("pf_ffn1", 512, "rows_gemm_blk_k<ReLU> (prefill FFN up-projection, 512 rows)")
        mfma["at_run_rows"] = mfma_obj("pf_ffn1", own_rows, f"rows_gemm_blk_k<ReLU> (prefill FFN up-projection, this run's {own_rows}-row pass)")
        mfma["at_2048_rows"] = mfma_obj("pf_ffn1", 2048, "rows_gemm_big_k<ReLU> (prefill FFN up-projection, a 2048-row stream: 256 x 256 tiles, LDS-DMA)")
        mfma["attention"] = mfma_obj("pf_attn", 512, "tile_attn_k (prefill attention, 512 causal rows of one sequence, all heads)")
        try:      # the second kernel (prompts of >= 768 rows): as many rows as this engine's cache holds, 1024 by default
            rows_long = min(2048, eng.max_positions) // 64 * 64
            if rows_long >= 768:
                mfma["attention"]["long_prompt"] = mfma_obj("pf_attn", rows_long, f"tile_attn64_k (prefill attention, {rows_long} causal rows of one sequence, all heads: "
                                                            "64 query rows per workgroup, P in registers)")
        except Exception as e:   # reporting only
            mfma["attention"]["long_prompt"] = {"error": str(e)}
        # per step TAKEN: the timed region also covers the (graph-rounded) tail of replayed no-op steps after the last sequence
        # retired, so this slightly OVERstates the step (never understates it); the launched count is kept as an annotation
        dec_step_ms = dec_ms / max(1, steps_run)
        out = {
            "metric": "codec_tokens_per_sec", "value": round(value, 1), "unit": "codec-tokens/s", "n_gpus": n_gpus,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": wl.label(not args.no_graph),
                       "utterances_per_step": B * n_gpus, "parallelism": f"dp{n_gpus} (utterance-sharded, one all_gather)"},
            "rtf": round(dt / (frames / 50.0), 4),
            "decode_ms_per_token_step": round(dec_step_ms, 4), "prefill_ms": round(pre_ms / args.steps, 2),
            "decode_steps": {"taken": steps_run // max(1, args.steps), "launched": steps_launched // max(1, args.steps)},
            "decode_step": {"alg_bytes": int(step_bytes), "isolated_step_ms": round(step_ms, 4),
                            "hbm_frac_in_loop": round(step_bytes / (dec_step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if dec_step_ms > 0 else None},
            "roofline": roof, "kernels": kernels,
        }
        out["prefill_roofline"] = mfma
        # launch-shape options in force (prefetch roles of latency-bound launches, finished-row form, ...): additional work
        # inside the timed region where they add any, nothing skipped
        out["config"]["engine_options"] = options_object(eng.options())
        if dist is not None:
            out["collective"] = {"backend": dist.get_backend(), "world": dist.get_world_size(), "op": "all_gather of int32 [B,K,T+1] token blocks",
                                 "gather_ms": round(gather_s[0] / args.steps * 1e3, 3), "ranks_share_one_device": share}
        if box is not None:
            out["box"] = box
            if B == 1 and not edit:
                try:
                    out["sampler"] = sampler_block(wl, box)
                except Exception as e:   # reporting only
                    out["sampler"] = {"error": str(e)}
        if n_gpus == 1 and B == 1 and not edit and not args.no_codec:
            try:
                out["one_sample"] = one_sample_block(eng, a, dev, args)
            except Exception as e:   # reporting only
                out["one_sample"] = {"error": str(e)}
        ab = args.ab
        more = []
        if ab == "auto":       # every default-ON feature has to show its gain in the line the driver records: the largest one in `ab`,
            # the other one-row forms of round 5 in `ab_more` (fewer pairs).  (The prefetch roles of rounds 3-5 left the tree in round 6.)
            ab = ("fr_one=0:1" if B == 1 else "finished_rows=0:16" if B <= 16 else "wide_gemm=0:1")
            if B == 1:
                more = ["qkv_p8=0:1"]
        if n_gpus == 1 and ab and ab != "none":
            try:
                out["ab"] = ab_block(eng, one_step, ab, max(3, args.ab_pairs))
            except Exception as e:   # reporting only
                out["ab"] = {"error": str(e)}
            out["ab_more"] = []
            for spec in more:
                try:
                    r = ab_block(eng, one_step, spec, max(3, min(5, args.ab_pairs)))
                    r.pop("metric", None)
                    out["ab_more"].append(r)
                except Exception as e:   # reporting only
                    out["ab_more"].append({"knob": spec, "error": str(e)})
        if n_gpus == 1 and B == 1 and not edit and args.preset == "giga830M" and not args.no_configs:
            # the other BASELINE configurations on their own engines (after the headline's timed region; ~20 s)
            del eng, wl
            out["configs"] = configs_block(args, dev, sd)
            out["ragged"] = ragged_block(args, dev, sd)
        if n_gpus == 1 and not args.no_codec:
            try:
                out["codec"] = codec_block(dev)
            except Exception as e:   # reporting only
                out["codec"] = {"error": str(e)}
        if n_gpus == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args, sd, a, prompts[0][0], prompts[0][1], prompts[0][2])
            except Exception as e:   # the baseline is reporting only; never lose the GPU number to it
                out["cpu_baseline"] = {"value": None, "unit": "codec-tokens/s", "cores": torch.get_num_threads(),
                                       "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
