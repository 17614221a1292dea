"""Prefill attention at several pass sizes; VC_TILE_ATTN=k[,min_rows] picks the kernel (1 = tile_attn_k: 16 query rows per wave; 2 = tile_attn64_k:
64 query rows per workgroup, from min_rows prompt rows - the option `tile_attn` of include/vc_engine.h).
usage: python tools/pf_attn_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from voicecraft_amd import synth
from voicecraft_amd.engine import VoiceCraftEngine
a = synth.make_args("giga830M")
sd = synth.make_state_dict(a, seed=0, perturb=False, fast=True)
eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype="bf16", max_seqs=1, max_positions=2304)
form = os.environ.get("VC_TILE_ATTN", "default")
for rows in (240, 512, 800, 1024, 2048):
    ms, fl = eng.bench_kernel("pf_attn", n_rows=rows, iters=32)
    print(f"[pf_attn] form {form} rows {rows:5d}: {ms * 1e3:7.2f} us  {fl / (ms * 1e-3) / 1e12:7.1f} TFLOP/s", flush=True)
