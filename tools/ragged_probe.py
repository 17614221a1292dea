#!/usr/bin/env python3
"""Per-call figures of bench.py's ragged block (shrinking batch against the fixed width), interleaved legs: decode ms, steps, re-packs."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    wl = bench.Workload("giga830M", "tts", B, 80, 150, 40, "bf16", "cuda:0", use_graph=True, lx_min=40)
    for v in (1, 0):
        wl.eng.set_option("shrink", v); wl.call(100); torch.cuda.synchronize()
    for i in range(6):
        for v in (1, 0) if i % 2 == 0 else (0, 1):
            wl.eng.set_option("shrink", v)
            t0 = time.perf_counter()
            tok = wl.call(1000 + i)[1]
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            tm = wl.eng.last_timing_ms()
            hm = wl.eng.debug_read("host_ms", (8,), torch.float64)
            print(json.dumps({"shrink": v, "call": i, "tok": int(tok), "wall_ms": round(wall * 1e3, 2), "decode_ms": round(tm["decode_ms"], 2), "prefill_ms": round(tm["prefill_ms"], 2),
                              "steps": wl.eng.last_steps, "repacks": int(hm[6]), "host_loop_ms": round(float(hm[3]), 2), "capture_ms": round(float(hm[1] + hm[2]), 2)}), flush=True)

if __name__ == "__main__":
    main()
