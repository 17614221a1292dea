"""Where does the prefill block GEMM's time go?  Times the 512-row FFN up-projection (rows_gemm_blk_k) with the
diagnostic mask VC_BLK_DBG: 0 = product, 1 = weights re-read from chunk 0 (L1/L2 hits), 2 = X re-read from chunk 0,
3 = both.  usage: python tools/blk_probe.py [rows ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from voicecraft_amd import synth
from voicecraft_amd.engine import VoiceCraftEngine
rows_list = [int(r) for r in sys.argv[1:]] or [512, 256, 128]
a = synth.make_args("giga830M")
sd = synth.make_state_dict(a, seed=0, perturb=False, fast=True)
eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype="bf16", max_seqs=1, max_positions=1024)
for rows in rows_list:
    out = []
    for dbg in (0, 1, 2, 3):
        os.environ["VC_BLK_DBG"] = str(dbg)
        ms, fl = eng.bench_kernel("pf_ffn1", n_rows=rows, iters=32)
        out.append(f"dbg{dbg} {ms * 1e3:.2f}us ({fl / (ms * 1e-3) / 1e12:.0f} TF/s)")
    print(f"form={os.environ.get('VC_BLK_FORM', '1')} rows={rows}", " | ".join(out), flush=True)
os.environ["VC_BLK_DBG"] = "0"
