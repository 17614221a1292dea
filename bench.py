#!/usr/bin/env python
"""bench.py — codec-tokens/s of the VoiceCraft TTS decode path on MI355X (BASELINE.json metric).

A "step" is one whole `inference_tts` call (prompt build + prefill + every decode step + un-shift)
on the synthetic 16 s workload of BASELINE config 3: giga830M shape, bf16, batch 1 per GPU,
top_k=40, Lx=80 phonemes, 150 prompt frames -> 650 generated frames (the reference's own length
cap ends generation).  codec tokens = K * generated frames.

  python bench.py --gpus 1 --steps 3 --warmup 1
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
    p.add_argument("--dump", default=None, help="rank 0 writes the token blocks gathered in the last step to this .npz (tests)")
    return p.parse_args()


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC pass (rocprofv3 --pmc FETCH_SIZE in its own
    run, x2-corrected for gfx950 as the microarchitecture guide prescribes); None when no pass is on file
    or the preset/dtype differs from the one profiled."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            return int(json.load(f)["kernels"][kernel]["fetch_bytes_per_launch"])
    except Exception:
        return None


def cpu_baseline(args, sd, a, x, x_lens, y):
    """The oracle (a port of the reference's CPU path: same ATen ops incl. the per-step KV torch.cat) timed on this
    box's host cores over a bounded sample of the same workload.  The reference's cost per step GROWS with the
    context (O(S) cache copies, BASELINE.md §2: 43 -> 27 tok/s), so both ends of the run are sampled: the prefill and the
    first n steps of the real run, and n steps at the END of the run's context (a second call whose prompt is as long
    as the real run's context n steps before its end); the whole run is priced with the mean of the two per-step costs."""
    from oracle.voicecraft_oracle import VoiceCraftOracle
    from voicecraft_amd import synth
    orc = VoiceCraftOracle(a, sd)
    torch.manual_seed(0)
    torch.set_num_threads(max(1, min(args.cpu_threads, os.cpu_count() or 1)))
    K = a.n_codebooks
    n = max(8, args.cpu_steps // 2)
    total_steps = 10 * args.lx - args.prompt_frames + K
    kn = dict(top_k=args.top_k, top_p=1.0, temperature=1.0, stop_repetition=3, kvcache=1)
    t0 = time.perf_counter()
    orc.inference_tts(x, x_lens, y, max_steps=1, **kn)
    t_prefill = time.perf_counter() - t0
    t0 = time.perf_counter()
    orc.inference_tts(x, x_lens, y, max_steps=n + 1, **kn)
    t_head = (time.perf_counter() - t0 - t_prefill) / n
    # the last n steps: same text, a prompt of (prompt + generated - n) frames -> context = the real run's, n steps early
    late_T = args.prompt_frames + max(0, total_steps - K - n)
    _, _, y_late = synth.random_prompt(a, args.lx, late_T, seed=7)
    t0 = time.perf_counter()
    orc.inference_tts(x, x_lens, y_late, max_steps=1, **kn)
    t_pre_late = time.perf_counter() - t0
    t0 = time.perf_counter()
    orc.inference_tts(x, x_lens, y_late, max_steps=n + 1, **kn)
    t_tail = (time.perf_counter() - t0 - t_pre_late) / n
    est = t_prefill + total_steps * 0.5 * (t_head + t_tail)
    tokens = K * (10 * args.lx - args.prompt_frames)
    return {
        "value": round(tokens / est, 2), "unit": "codec-tokens/s", "cores": torch.get_num_threads(), "kind": "port",
        "sample": (f"{args.preset} fp32, Lx={args.lx}, {args.prompt_frames} prompt frames: prefill {t_prefill:.2f} s, first {n} decode steps "
                   f"{t_head * 1e3:.0f} ms/step, {n} steps at the end of the run's context ({args.lx + late_T + 1} positions) "
                   f"{t_tail * 1e3:.0f} ms/step; whole run of {total_steps} steps priced at their mean = {est:.1f} s. "
                   "The oracle is a port (1.3-1.5x faster than the unmodified reference on the build box, VERDICT r01); "
                   "the reference itself measured 27.1 tok/s on 8 cores (BASELINE.md §2)"),
    }


def codec_block(dev):
    """EnCodec encode / decode of 16 s of audio (synthetic weights at the VoiceCraft codec shape, fp32 MFMA), SURVEY §8d
    'report codec encode/decode separately'.  The dominant kernel is the LSTM wavefront step, which re-reads the 50 MB of
    fp32 recurrence weights per step out of L2 / Infinity Cache."""
    from voicecraft_amd import synth
    from voicecraft_amd.codec import AudioTokenizer
    tok = AudioTokenizer(synth.make_codec_state_dict(0), device=dev, max_seconds=16.5, max_batch=1)
    torch.manual_seed(0)
    wav = (torch.randn(1, 1, 16 * 16000) * 0.1).to(dev)
    codes = tok.encode(wav)[0][0]
    enc = []
    for _ in range(3):
        tok.encode(wav)
        enc.append(tok.last_ms())
    lstm_ms, lstm_bytes = tok.last_lstm_ms()
    T = int(codes.shape[2])
    dec = []
    for _ in range(3):
        tok.decode([(codes, None)])
        dec.append(tok.last_ms())
    e_ms, d_ms = min(enc), min(dec)
    step_us = lstm_ms * 1e3 / (T + 1)
    return {"audio_s": 16.0, "frames": T, "encode_ms": round(e_ms, 2), "decode_ms": round(d_ms, 2),
            "rtf_encode": round(e_ms / 16e3, 6), "rtf_decode": round(d_ms / 16e3, 6), "dtype": "f32", "parity": "unpinned (audiocraft not vendored)",
            "roofline": {"bound": "hbm", "kernel": "lstm_wave_k (one recurrence step of both LSTM layers)",
                         "achieved": round(lstm_bytes / (step_us * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(lstm_bytes / (step_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                         "bytes_per_launch": lstm_bytes, "avg_launch_us": round(step_us, 2),
                         "note": "weights are cache-resident (50 MB < 256 MB Infinity Cache): priced against HBM as the conservative bound"}}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist = None
    n_gpus = world
    assert args.gpus == n_gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from voicecraft_amd import dist as vdist
    from voicecraft_amd import synth
    from voicecraft_amd.engine import VoiceCraftEngine
    a = synth.make_args(args.preset)
    K = a.n_codebooks
    sd = synth.make_state_dict(a, seed=0, perturb=False, mute_eos=True, fast=True)
    edit = args.mode == "edit"
    if edit:       # 16 s utterance (10 frames per phoneme), the middle quarter masked: generation runs to the
        assert args.batch == 1, "editing is single-utterance (models/voicecraft.py:607)"   # reference's length cap y_len > 10*Lx
        args.prompt_frames = 10 * args.lx
        span = (args.prompt_frames * 3 // 8, args.prompt_frames * 5 // 8)
    Tg = 10 * args.lx - args.prompt_frames if not edit else span[1] - span[0]
    eng = VoiceCraftEngine(a, sd, device=dev, dtype=args.dtype, max_seqs=max(1, args.batch),
                           max_positions=max(1024, args.lx + args.prompt_frames + Tg + 64), use_graph=not args.no_graph)
    # utterance u of this rank = global utterance (u * world + rank); seed = 1 + global index
    B = args.batch
    prompts = [synth.random_prompt(a, args.lx, args.prompt_frames, seed=1 + (u * world + rank)) for u in range(B)]
    xs = [p[0].to(dev) for p in prompts]
    xls = [p[1].to(dev) for p in prompts]
    ys = [p[2].to(dev) for p in prompts]
    knobs = dict(top_k=args.top_k, top_p=1.0, temperature=1.0, stop_repetition=3)

    def one_step(seed):
        if edit:
            mi = torch.tensor([[list(span)]], dtype=torch.int64)
            res = eng.inference(xs[0], xls[0], ys[0], mi, silence_tokens=[1388, 1898, 131], _seed=seed, **knobs)
            return (int(res.shape[2]) - (args.prompt_frames - (span[1] - span[0]))) * K
        if B == 1:
            res, gen = eng.inference_tts(xs[0], xls[0], ys[0], kvcache=1, silence_tokens=[1388, 1898, 131], _seed=seed, **knobs)
            gens = [gen]
        else:
            outs = eng.inference_tts_multi([x[0] for x in xs], [y[0] for y in ys], silence_tokens=[1388, 1898, 131], _seed=seed, **knobs)
            gens = [o[1] for o in outs]
        n_tok = sum(int(g.shape[2]) * K for g in gens)
        if dist is not None or args.dump:   # the single collective of the job: gather every rank's token block
            gathered[:] = vdist.gather_token_blocks([g[0] for g in gens], Tg + 8, n_slots=B, K=K, device=dev)
        return n_tok

    gathered = []

    for w in range(args.warmup):
        one_step(100 + w)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tokens = 0
    dec_ms = pre_ms = 0.0
    steps_run = 0
    for s in range(args.steps):
        tokens += one_step(1000 + s)
        tm = eng.last_timing_ms()
        dec_ms += tm["decode_ms"]; pre_ms += tm["prefill_ms"]; steps_run += eng.last_steps
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        tot = torch.tensor([tokens], dtype=torch.int64, device=dev)
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
        d, L = a.d_model, a.num_decoder_layers
        V, P = a.audio_vocab_size + a.n_special, a.audio_vocab_size // 2
        esz = 2 if args.dtype == "bf16" else 4
        w_bytes = esz * (L * (12 * d * d + 13 * d) + 2 * d + K * (d * P + P + P * V + V))
        s_mean = args.lx + args.prompt_frames + 1 + Tg / 2
        step_bytes = w_bytes + B * esz * 2 * L * d * (s_mean + 1)        # weights (once per step) + KV read + KV write of each of the B sequences
        # dominant kernel: the FFN up-projection rows-GEMM (LayerNorm prologue, ReLU epilogue)
        mb_rows = min(B, 16)       # the kernel microbenchmarks drive the <=16-row decode kernels
        k_ms, k_bytes = eng.bench_kernel("ffn1", n_rows=mb_rows, iters=64)
        step_ms, _ = eng.bench_kernel("step", n_rows=mb_rows, iters=8)
        kernels = {}
        for kn in ("qkv", "attn", "oproj", "ffn1", "ffn2", "qkv_hot", "oproj_hot", "ffn1_hot", "ffn2_hot"):
            ms_, by_ = eng.bench_kernel(kn, n_rows=mb_rows, iters=64)
            kernels[kn] = {"avg_us": round(ms_ * 1e3, 2), "GB/s": round(by_ / (ms_ * 1e-3) / 1e9, 1)}
        roof = {"bound": "hbm", "kernel": "rows_gemm_k<LN,ReLU> (FFN up-projection)",
                "achieved": round(k_bytes / (k_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(k_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": pmc_traffic("ffn1") if (args.preset == "giga830M" and args.dtype == "bf16") else None,
                "bytes_per_launch": k_bytes, "avg_launch_us": round(k_ms * 1e3, 2)}
        mfma = None
        if edit:       # the prefill of the editing call is GEMM-shaped: the MFMA roofline of its widest block GEMM
            pf_rows = 512
            pf_ms, pf_flops = eng.bench_kernel("pf_ffn1", n_rows=pf_rows, iters=32)
            mfma = {"bound": "mfma", "kernel": f"rows_gemm_blk_k<ReLU> (prefill FFN up-projection, {pf_rows} rows)",
                    "achieved": round(pf_flops / (pf_ms * 1e-3) / 1e12, 1), "peak": MFMA_PEAK_TF[args.dtype], "unit": "TFLOP/s",
                    "frac": round(pf_flops / (pf_ms * 1e-3) / 1e12 / MFMA_PEAK_TF[args.dtype], 4), "traffic": None,
                    "flops_per_launch": pf_flops, "avg_launch_us": round(pf_ms * 1e3, 2)}
        dec_step_ms = dec_ms / max(1, steps_run)
        out = {
            "metric": "codec_tokens_per_sec", "value": round(value, 1), "unit": "codec-tokens/s", "n_gpus": n_gpus,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": (f"{args.preset} TTS, batch {B}/GPU, Lx={args.lx}, {args.prompt_frames} prompt frames -> "
                                    f"{Tg} generated frames (16 s total), top_k={args.top_k}, hipGraph={'off' if args.no_graph else 'on'}")
                                   if not edit else
                                   (f"{args.preset} speech editing, Lx={args.lx}, {args.prompt_frames}-frame utterance, span "
                                    f"{span} re-generated (~{Tg} frames, reference length cap), top_k={args.top_k}"),
                       "utterances_per_step": B * n_gpus, "parallelism": f"dp{n_gpus} (utterance-sharded, one all_gather)"},
            "rtf": round(dt / (frames / 50.0), 4),
            "decode_ms_per_token_step": round(dec_step_ms, 4), "prefill_ms": round(pre_ms / args.steps, 2),
            "decode_step": {"alg_bytes": int(step_bytes), "isolated_step_ms": round(step_ms, 4),
                            "hbm_frac_in_loop": round(step_bytes / (dec_step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if dec_step_ms > 0 else None},
            "roofline": roof, "kernels": kernels,
        }
        if mfma is not None:
            out["prefill_roofline"] = mfma
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
