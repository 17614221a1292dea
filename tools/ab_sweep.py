"""Several in-process A/Bs (bench.ab_block) on ONE engine: python tools/ab_sweep.py [--preset giga830M] [--batch 1] [--mode tts]
[--pairs 7] knob=A:B [knob=A:B ...] [--set knob=value ...].  Prints one JSON object per spec.  Saves the engine creation and the
import per A/B that separate bench.py runs would pay (GPU lease minutes)."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from voicecraft_amd import synth
from voicecraft_amd.engine import VoiceCraftEngine

p = argparse.ArgumentParser()
p.add_argument("specs", nargs="*")
p.add_argument("--preset", default="giga830M")
p.add_argument("--batch", type=int, default=1)
p.add_argument("--lx", type=int, default=80)
p.add_argument("--prompt-frames", type=int, default=150)
p.add_argument("--top-k", type=int, default=40)
p.add_argument("--pairs", type=int, default=7)
p.add_argument("--set", action="append", default=[], help="knob=value applied before the sweep")
p.add_argument("--kernels", action="store_true", help="print the isolated kernel microbenchmarks after every --set / at the end")
args = p.parse_args()
a = synth.make_args(args.preset)
sd = synth.make_state_dict(a, seed=0, perturb=False, mute_eos=True, fast=True)
dev = torch.device("cuda", 0)
B = args.batch
eng = VoiceCraftEngine(a, sd, device=dev, dtype="bf16", max_seqs=B, max_positions=1024)
prompts = [synth.random_prompt(a, args.lx, args.prompt_frames, seed=1 + u) for u in range(B)]
xs = [q[0].to(dev) for q in prompts]; xls = [q[1].to(dev) for q in prompts]; ys = [q[2].to(dev) for q in prompts]
kn = dict(top_k=args.top_k, top_p=1.0, temperature=1.0, stop_repetition=3)


def one_step(seed):
    if B == 1:
        eng.inference_tts(xs[0], xls[0], ys[0], kvcache=1, silence_tokens=[1388, 1898, 131], _seed=seed, **kn)
    else:
        eng.inference_tts_multi([x[0] for x in xs], [y[0] for y in ys], silence_tokens=[1388, 1898, 131], _seed=seed, **kn)


def kernels():
    out = {}
    for k in ("qkv", "attn", "oproj", "ffn1", "ffn2", "step"):
        ms, by = eng.bench_kernel(k, n_rows=min(B, 16), iters=64 if k != "step" else 8)
        out[k] = round(ms * 1e3, 2)
    return out


for kv in args.set:
    k, v = kv.split("=", 1)
    eng.set_option(k, v)
print(json.dumps({"preset": args.preset, "batch": B, "options": eng.options()}), flush=True)
if args.kernels:
    print(json.dumps({"kernels_us": kernels()}), flush=True)
for spec in args.specs:
    try:
        r = bench.ab_block(eng, one_step, spec, args.pairs)
        r.pop("deltas_pct", None)
    except Exception as e:
        r = {"spec": spec, "error": str(e)}
    print(json.dumps(r), flush=True)
    if args.kernels:
        print(json.dumps({"options": eng.options(), "kernels_us": kernels()}), flush=True)
