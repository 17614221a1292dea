"""profiles/pmc_traffic.json from a tools/prof_pmc.sh FETCH_SIZE summary: HBM bytes per launch of the decode GEMMs (x2
correction of gfx950's FETCH_SIZE tally for wide coalesced streams, MI355X_MICROARCH.md §HBM), tagged with the bench
configuration the pass was taken on - bench.py reports `roofline.traffic` only when its own run matches that tag.
usage: python tools/pmc_to_json.py <pmc summary .txt> <out .json> [preset dtype batch lx prompt_frames mode]"""
import json, os, re, socket, sys
src, dst = sys.argv[1], sys.argv[2]
cfg = dict(zip(("preset", "dtype", "batch", "lx", "prompt_frames", "mode"), sys.argv[3:9]))
cfg = {"preset": cfg.get("preset", "giga830M"), "dtype": cfg.get("dtype", "bf16"), "batch": int(cfg.get("batch", 1)),
       "lx": int(cfg.get("lx", 80)), "prompt_frames": int(cfg.get("prompt_frames", 150)), "mode": cfg.get("mode", "tts")}
d = 2048
alg = {"ffn1": 4 * d * d * 2 + d * 4 + 4 * d * 2, "ffn2": 4 * d * d * 2 + 4 * d * 2 + d * 4, "qkv": 3 * d * d * 2 + d * 4 + 3 * d * 2,
       "oproj": d * d * 2 + d * 4 * 2}
# first pattern that matches wins (the forms a one-row step launches since round 5, then the round-4 forms)
pat = {"ffn1": [r"rows_gemm_k<bf16_t, 16, 0, 2,"], "ffn2": [r"row_gemm_fr1_k<bf16_t, 16, true, 8, 1, 5>", r"rows_gemm_k<bf16_t, 16, 1, 1,"],
       "qkv": [r"row_gemm_fr1_k<bf16_t, 8, true, 4, 0, 0>", r"rows_gemm_k<bf16_t, 16, 0, 0,"], "oproj": [r"rows_gemm_k<bf16_t, 8, 2, 1,"]}
out = {"source": f"{src} (rocprofv3 --pmc FETCH_SIZE --kernel-trace, own pass of `python bench.py --steps 1 --warmup 0`; FETCH_SIZE[KB] * 1024 * 2)",
       "config": cfg, "kernels": {}}
lines = [l for l in open(src) if "FETCH_SIZE" in l]
for k, ps in pat.items():
    for p in ps:
        hit = [l for l in lines if l.startswith(p)]
        if hit:
            calls = tot = 0
            for l in hit:
                f = l.split("FETCH_SIZE")[1].split()
                calls += int(f[0]); tot += int(f[0]) * float(f[1])
            out["kernels"][k] = {"name": p, "calls": calls, "fetch_bytes_per_launch": int(round(2 * tot / calls * 1024)), "algorithmic_bytes": alg[k]}
            break
try:      # the library the pass was taken on (digest of its sources, voicecraft_amd/build.py) and the box: bench.py quotes the figures only for the same build
    out["lib_stamp"] = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "voicecraft_amd", ".build_stamp")).read()[:16]
except Exception:
    out["lib_stamp"] = None
out["box"] = socket.gethostname()
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
