"""The prefill QKV projection on the 12-channel and on the 16-channel image of the matrix (option qkv16), isolated:
python tools/qkv16_probe.py [--preset giga830M] -> one JSON line per row count: microseconds and the fraction of the bf16 MFMA peak."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from voicecraft_amd import synth
from voicecraft_amd.engine import VoiceCraftEngine

p = argparse.ArgumentParser()
p.add_argument("--preset", default="giga830M")
p.add_argument("--rows", default="240,512,800,1280,2048")
args = p.parse_args()
a = synth.make_args(args.preset)
sd = synth.make_state_dict(a, seed=0, perturb=False, mute_eos=True, fast=True)
eng = VoiceCraftEngine(a, sd, device=torch.device("cuda", 0), dtype="bf16", max_seqs=1, max_positions=2304)
for rows in [int(r) for r in args.rows.split(",")]:
    out = {"rows": rows}
    for q16 in (0, 1, 0, 1):
        eng.set_option("qkv16", q16)
        ms, flops = eng.bench_kernel("pf_qkv", n_rows=rows, iters=48)
        k = f"q16={q16}"
        out[k] = min(out.get(k, 1e9), round(ms * 1e3, 2))
    out["delta_pct"] = round(100.0 * (out["q16=1"] / out["q16=0"] - 1.0), 2)
    out["mfma_frac_q16"] = round(flops / (out["q16=1"] * 1e-6) / 2.5e15, 4)
    print(json.dumps(out), flush=True)
