"""Does a weight matrix that weight_prefetch_k has just streamed make the launch that reads it run 'hot'?
Times, per matrix of a layer: the cold launch, the launch on cache-resident weights (_hot), the prefetch alone and
prefetch + launch back to back on one stream (so launch-after-prefetch = the difference).
usage: python tools/pf_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from voicecraft_amd import synth
from voicecraft_amd.engine import VoiceCraftEngine
a = synth.make_args("giga830M")
sd = synth.make_state_dict(a, seed=0, perturb=False, fast=True)
eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype="bf16", max_seqs=1, max_positions=1024)
for kn in ("qkv", "oproj", "ffn1", "ffn2"):
    r = {}
    for suf in ("", "_hot", "_pfonly", "_pf"):
        ms, _ = eng.bench_kernel(kn + suf, n_rows=1, iters=64)
        r[suf] = ms * 1e3
    print(f"[pf] {kn}: cold {r['']:.2f} us | hot {r['_hot']:.2f} us | prefetch alone {r['_pfonly']:.2f} us | prefetch + launch {r['_pf']:.2f} us "
          f"=> launch after prefetch {r['_pf'] - r['_pfonly']:.2f} us", flush=True)
