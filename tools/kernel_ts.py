"""In-kernel shader-clock stamps of the decode kernels (diagnostic build: python voicecraft_amd/build.py --ts).

Runs each kernel's microbenchmark against libvcengine_ts.so and prints, for the first and the last
workgroup of the last launch, the clock deltas between the stages of the kernel:
  rows-GEMM : entry | loads+burst issued | prologue math done | block sync | weights consumed | K-reduce sync | epilogue
  attention : entry | pos known | loads issued | q arrived | K/V consumed | wave merge | block sync | end
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("VC_ENGINE_LIB", os.path.join(ROOT, "voicecraft_amd", "libvcengine_ts.so"))
import numpy as np, torch
from voicecraft_amd import synth
from voicecraft_amd.engine import VoiceCraftEngine

GEMM = ["issue(loads+burst out, active known)", "prologue math", "block sync", "weights consumed", "K-reduce sync", "epilogue"]
ATTN = ["pos known", "loads issued", "q arrived+scaled", "K/V consumed", "wave merge", "block sync", "final+store"]
PRESET = sys.argv[1] if len(sys.argv) > 1 else "giga830M"
ROWS = int(sys.argv[2]) if len(sys.argv) > 2 else 1
a = synth.make_args(PRESET)
print(f"# {PRESET}, {ROWS} row(s)")
sd = synth.make_state_dict(a, seed=0, perturb=False, fast=True)
eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype="bf16", max_seqs=max(1, ROWS), max_positions=1024)
for kn in ("qkv", "qkv_hot", "attn", "oproj", "ffn1", "ffn2", "ffn1_hot", "ffn2_hot"):
    ms, _ = eng.bench_kernel(kn, n_rows=ROWS, iters=32)
    ts = eng.debug_read("kernel_ts", (32,), dtype=torch.int64).numpy()
    names = ATTN if kn.startswith("attn") else GEMM
    for blk, t in (("first", ts[:16]), ("last", ts[16:])):
        n = len(names) + 1
        d = np.diff(t[:n])
        print(f"{kn:9s} {blk:5s} wg: avg launch {ms*1e3:6.2f} us | in-kernel {int(t[n-1]-t[0]):6d} clk | " +
              " | ".join(f"{nm} {int(v)}" for nm, v in zip(names, d)), flush=True)
    print(f"{kn:9s} last wg entered {int(ts[16]-ts[0])} clk after the first", flush=True)
