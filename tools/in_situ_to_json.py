"""profiles/in_situ.json from a tools/prof_decode.sh kernel-stats summary: the IN-SITU average duration of the decode step's kernels
(rocprofv3 --kernel-trace over a whole `python bench.py` run, every launch of the decode loop - rotating layers, cold caches), which
bench.py quotes next to its isolated microbenchmark as `roofline.in_situ` when its own run matches the tag.
usage: python tools/in_situ_to_json.py <kernel stats .txt> <out .json> [preset dtype batch lx prompt_frames mode]"""
import json, os, re, socket, sys
src, dst = sys.argv[1], sys.argv[2]
cfg = dict(zip(("preset", "dtype", "batch", "lx", "prompt_frames", "mode"), sys.argv[3:9]))
cfg = {"preset": cfg.get("preset", "giga830M"), "dtype": cfg.get("dtype", "bf16"), "batch": int(cfg.get("batch", 1)),
       "lx": int(cfg.get("lx", 80)), "prompt_frames": int(cfg.get("prompt_frames", 150)), "mode": cfg.get("mode", "tts")}
d = {"giga830M": 2048, "giga330M": 1024}.get(cfg["preset"], 2048)
alg = {"ffn1": 4 * d * d * 2 + d * 4 + 4 * d * 2, "ffn2": 4 * d * d * 2 + 4 * d * 2 + d * 4, "qkv": 3 * d * d * 2 + d * 4 + 3 * d * 2,
       "oproj": d * d * 2 + d * 4 * 2}
# one-row step: the form each matrix runs in (first pattern that matches a line wins; most calls wins among equal names)
pat = {"ffn2": [r"row_gemm_fr1_k<bf16_t, \d+, \w+, 8, 1, 5>", r"rows_gemm_k<bf16_t, \d+, 1, 1,"], "qkv": [r"row_gemm_fr1_k<bf16_t, \d+, \w+, 4, 0, 0>", r"rows_gemm_k<bf16_t, \d+, 0, 0,"],
       "ffn1": [r"rows_gemm_k<bf16_t, \d+, 0, 2,"], "oproj": [r"rows_gemm_k<bf16_t, \d+, 2, 1,"], "attn": [r"rows_attn_k<bf16_t"],
       "heads1": [r"rows_gemm_k<bf16_t, \d+, 0, 3,"], "heads2": [r"rows_gemm_k<bf16_t, \d+, 1, 4,"], "sampler": [r"sample_fused_k"]}
rows = []
for line in open(src):
    m = re.match(r"(\S.*?)\s+\((\d+), (\d+), (\d+)\)\s+(\d+)\s+(\d+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)", line)
    if m:
        rows.append({"name": m.group(1).strip(), "grid": [int(m.group(i)) for i in (2, 3, 4)], "calls": int(m.group(7)),
                     "total_ms": float(m.group(8)), "pct": float(m.group(9)), "avg_us": float(m.group(10))})
out = {"source": f"{src} (rocprofv3 --kernel-trace --stats over `python bench.py --steps 2 --warmup 1`, tools/prof_decode.sh)", "config": cfg, "kernels": {}}
for k, ps in pat.items():
    for p in ps:
        hit = [r for r in rows if re.match(p, r["name"])]
        if hit:
            calls = sum(r["calls"] for r in hit)
            tot = sum(r["total_ms"] for r in hit)
            e = {"name": hit[0]["name"], "calls": calls, "avg_us": round(tot * 1e3 / calls, 3), "pct_of_kernel_time": round(sum(r["pct"] for r in hit), 2)}
            if k in alg:
                e["algorithmic_bytes"] = alg[k]
                e["frac_of_8TBs"] = round(alg[k] / (e["avg_us"] * 1e-6) / 8e12, 4)
            out["kernels"][k] = e
            break
try:      # the library the pass was taken on (digest of its sources, voicecraft_amd/build.py) and the box: bench.py quotes the figures only for the same build
    out["lib_stamp"] = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "voicecraft_amd", ".build_stamp")).read()[:16]
except Exception:
    out["lib_stamp"] = None
out["box"] = socket.gethostname()
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
