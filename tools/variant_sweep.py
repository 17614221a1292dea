"""Microbenchmarks of the decode kernels for one engine library (VC_ENGINE_LIB), one line per kernel.
usage: VC_ENGINE_LIB=voicecraft_amd/libvcengine_<variant>.so python tools/variant_sweep.py [rows]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from voicecraft_amd import synth
from voicecraft_amd.engine import VoiceCraftEngine
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1
a = synth.make_args("giga830M")
sd = synth.make_state_dict(a, seed=0, perturb=False, fast=True)
eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype="bf16", max_seqs=max(1, rows), max_positions=1024)
tag = os.path.basename(os.environ.get("VC_ENGINE_LIB", "libvcengine.so"))
out = []
for kn in ("qkv", "attn", "oproj", "ffn1", "ffn2", "step"):
    ms, _ = eng.bench_kernel(kn, n_rows=rows, iters=64 if kn != "step" else 16)
    out.append(f"{kn} {ms * 1e3:.2f}us")
print(tag, f"rows={rows}", " | ".join(out), flush=True)
