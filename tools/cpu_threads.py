"""Times the oracle (CPU port of the reference path) at several intra-op thread counts on this
box, to pick an honest CPU baseline configuration.  Run on the GPU box's host."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from voicecraft_amd import synth
from oracle.voicecraft_oracle import VoiceCraftOracle

a = synth.make_args("giga830M")
sd = synth.make_state_dict(a, seed=0, perturb=False, fast=True)
x, xl, y = synth.random_prompt(a, 80, 150, seed=1)
orc = VoiceCraftOracle(a, sd)
print("cpu_count", os.cpu_count(), flush=True)
for nt in [int(v) for v in (sys.argv[1:] or ["8", "16", "32", "64"])]:
    torch.set_num_threads(nt)
    t0 = time.perf_counter()
    orc.inference_tts(x, xl, y, top_k=40, stop_repetition=3, max_steps=1)
    t1 = time.perf_counter()
    orc.inference_tts(x, xl, y, top_k=40, stop_repetition=3, max_steps=13)
    t2 = time.perf_counter()
    per_step = ((t2 - t1) - (t1 - t0)) / 12
    print(f"threads={nt}: prefill+1 step {t1 - t0:.2f}s, decode {per_step * 1e3:.1f} ms/step -> {4 / per_step:.1f} codec-tok/s", flush=True)
