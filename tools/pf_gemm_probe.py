"""Prefill FFN up-projection at several pass sizes: TFLOP/s of the block GEMM the engine picks (128 x 128 up to 512 rows,
256 x 256 LDS-DMA tiles beyond).  usage: python tools/pf_gemm_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from voicecraft_amd import synth
from voicecraft_amd.engine import VoiceCraftEngine
a = synth.make_args("giga830M")
sd = synth.make_state_dict(a, seed=0, perturb=False, fast=True)
eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype="bf16", max_seqs=1, max_positions=2304)
for rows in (240, 512, 768, 800, 1024, 1280, 1536, 1920, 2048):
    ms, fl = eng.bench_kernel("pf_ffn1", n_rows=rows, iters=32)
    print(f"[pf_gemm] {'big off' if os.environ.get('VC_NO_BIG_GEMM') else ('no 256x128' if os.environ.get('VC_NO_BIG128') else 'default')} rows {rows:5d}: {ms * 1e3:7.2f} us  {fl / (ms * 1e-3) / 1e12:7.1f} TFLOP/s  ({fl / (ms * 1e-3) / 1e12 / 2500:.3f} of 2.5 PF)", flush=True)
