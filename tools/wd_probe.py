#!/usr/bin/env python
"""The linear layers of a WIDE decode step (17..64 rows), isolated: rows_gemm_wd_k (option wide_gemm = 1, round 6) against the
weight-stationary rows_gemm_mt_k of rounds 2-5 (wide_gemm = 0), per matrix and row count, with the algorithmic
bytes per launch and the fraction of the 8 TB/s HBM peak; plus the step's LayerNorm launch and its attention launch.
  python tools/wd_probe.py [preset] > profiles/r06_wd_probe.log"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from voicecraft_amd import synth
from voicecraft_amd.engine import VoiceCraftEngine

preset = sys.argv[1] if len(sys.argv) > 1 else "giga830M"
a = synth.make_args(preset)
sd = synth.make_state_dict(a, seed=0, perturb=False, fast=True)
eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype="bf16", max_seqs=64, max_positions=1024)
out = {"preset": preset}
for rows in (32, 48, 64):
    for which in ("wd_ffn1", "wd_ffn2", "wd_qkv", "wd_oproj"):
        r = {}
        for name, opts in (("wd", {"wide_gemm": 1}), ("mt2", {"wide_gemm": 0})):
            for k, v in opts.items():
                eng.set_option(k, v)
            eng.bench_kernel(which, n_rows=rows, iters=16)
            ms, by = min(eng.bench_kernel(which, n_rows=rows, iters=128), eng.bench_kernel(which, n_rows=rows, iters=128))
            r[name] = {"us": round(ms * 1e3, 2), "GB/s": round(by / (ms * 1e-3) / 1e9, 1), "hbm_frac": round(by / (ms * 1e-3) / 8e12, 4)}
        r["delta_pct"] = round(100.0 * (r["wd"]["us"] / r["mt2"]["us"] - 1.0), 1)
        out[f"{which}@{rows}"] = r
    eng.set_option("wide_gemm", 1)
    # X through the wave-private LDS stage (wd_stage = 1, the default) against fragments straight from L2 (0)
    for which in ("wd_ffn1", "wd_ffn2", "wd_qkv", "wd_oproj"):
        r = {}
        for o in (1, 0, 1):
            eng.set_option("wd_stage", o)
            eng.bench_kernel(which, n_rows=rows, iters=16)
            ms, by = min(eng.bench_kernel(which, n_rows=rows, iters=128), eng.bench_kernel(which, n_rows=rows, iters=128))
            r.setdefault(f"stage{o}", []).append(round(ms * 1e3, 2))
        out[f"{which}@{rows} staged / direct"] = r
    eng.set_option("wd_stage", 1)
    for which in ("wd_ln", "wd_attn"):
        eng.bench_kernel(which, n_rows=rows, iters=16)
        ms, by = min(eng.bench_kernel(which, n_rows=rows, iters=128), eng.bench_kernel(which, n_rows=rows, iters=128))
        out[f"{which}@{rows}"] = {"us": round(ms * 1e3, 2), "GB/s": round(by / (ms * 1e-3) / 1e9, 1)}
for k, v in out.items():
    print(k, json.dumps(v) if isinstance(v, dict) else v)
