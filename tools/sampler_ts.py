"""Reads the shader-clock stamps of the sampler kernel (VC_SAMPLER_TS=1) after a short run."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["VC_SAMPLER_TS"] = "1"
from voicecraft_amd import synth
from voicecraft_amd.engine import VoiceCraftEngine
a = synth.make_args("giga830M")
sd = synth.make_state_dict(a, seed=0, perturb=False, fast=True)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype="bf16", max_seqs=B, max_positions=1024)
x, xl, y = synth.random_prompt(a, 20, 150, seed=1)
prompts = [synth.random_prompt(a, 20, 150, seed=1 + u) for u in range(B)]
for rep in range(2):
    if B == 1:
        eng.inference_tts(x.cuda(), xl.cuda(), y.cuda(), top_k=40, _seed=1)
    else:
        eng.inference_tts_multi([p[0][0].cuda() for p in prompts], [p[2][0].cuda() for p in prompts], top_k=40, _seed=1)
    ts = eng.debug_read("kernel_ts", (64,), dtype=torch.int64).numpy()
    blk = ts[16:32].reshape(8, 2)
    wall = ts[32:48].reshape(8, 2)[:B]          # 100 MHz chip-wide counter: entry / exit of every block
    w0 = wall[:, 0].min()
    print("wall clock, us after the first block's entry: entries", [round((int(v) - int(w0)) / 100.0, 2) for v in wall[:, 0]],
          "exits", [round((int(v) - int(w0)) / 100.0, 2) for v in wall[:, 1]], flush=True)
    print("per-block (entry -> TS0 of block 0 is", int(ts[0] - blk[0, 0]), "clk); in-kernel clocks per block:", [int(e - s) for s, e in blk[:B]], flush=True)
    acc = ts[48:59]
    if acc[8] > 0:
        nm = ["state parked + row in LDS", "edits + arg-max", "filter + draw", "cond + sync", "advance (thread 0)", "next-row embedding", "store state"]
        print(f"MEAN over {int(acc[8])} steps of this and earlier calls, block 0 (shader clocks):", {n: int(v / acc[8]) for n, v in zip(nm, acc[:7])},
              "entry->exit", int(acc[7] / acc[8]), "| last block entry->exit", int(acc[9] / max(1, acc[10])), flush=True)
    idx = [0, 1, 2, 5, 6, 7, 8, 9]          # the stamps the kernel sets (vc_tokens.hip VC_TS)
    d = np.diff(ts[idx])
    names = ["state parked + row in LDS", "edits + arg-max", "temperature/top-k/softmax/top-p/draw", "cond + sync", "advance (thread 0)",
             "next-row embedding", "store state"]
    print(f"sampler stamps, sequence 0 of {B} (shader clocks, last step):", {n: int(v) for n, v in zip(names, d)}, "total", int(ts[9] - ts[0]), flush=True)
