"""Chained vs stream-ordered decode pass (vc_common.h "chained launches"): the 5L+2 launches of one
forward pass + heads, giga830M bf16, one row; then a full TTS call with VC_CHAIN=1 checked against VC_CHAIN=0."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from voicecraft_amd import synth
from voicecraft_amd.engine import VoiceCraftEngine
a = synth.make_args("giga830M")
sd = synth.make_state_dict(a, seed=0, perturb=False, mute_eos=True, fast=True)
eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype="bf16", max_seqs=1, max_positions=1024)
for rep in range(2):
    ms0, _ = eng.bench_kernel("step", n_rows=1, iters=8)
    try:
        ms1, _ = eng.bench_kernel("step_chain", n_rows=1, iters=8)
    except Exception as e:
        print("step_chain failed:", e, flush=True)
        continue
    print(f"forward pass + heads (82 launches): stream order {ms0*1e3:.1f} us, chained {ms1*1e3:.1f} us", flush=True)
