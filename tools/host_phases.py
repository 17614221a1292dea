"""Host wall-clock phases of one inference_tts call (giga830M bf16, the bench workload): where the time
outside the decode kernels goes (graph capture / instantiate / replay loop / python)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from voicecraft_amd import synth
from voicecraft_amd.engine import VoiceCraftEngine
a = synth.make_args("giga830M")
sd = synth.make_state_dict(a, seed=0, perturb=False, mute_eos=True, fast=True)
eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype="bf16", max_seqs=1, max_positions=1024)
x, xl, y = synth.random_prompt(a, 80, 150, seed=1)
x, xl, y = x.cuda(), xl.cuda(), y.cuda()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res, gen = eng.inference_tts(x, xl, y, top_k=40, stop_repetition=3, silence_tokens=[1388, 1898, 131], _seed=rep)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    h = eng.debug_read("host_ms", (8,), dtype=torch.float64).numpy()
    tm = eng.last_timing_ms()
    print(f"call {dt:.2f} ms | events: prefill {tm['prefill_ms']:.2f} decode {tm['decode_ms']:.2f} | host: capture {h[1]:.2f} "
          f"instantiate {h[2]:.2f} replay loop {h[3]:.2f} destroy {h[4]:.2f} | steps {eng.last_steps}", flush=True)
