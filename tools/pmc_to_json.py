"""profiles/pmc_traffic.json from a tools/prof_pmc.sh FETCH_SIZE summary: HBM bytes per launch of the decode GEMMs (x2
correction of gfx950's FETCH_SIZE tally for wide coalesced streams, MI355X_MICROARCH.md §HBM), tagged with the bench
configuration the pass was taken on - bench.py reports `roofline.traffic` only when its own run matches that tag.
usage: python tools/pmc_to_json.py <pmc summary .txt> <out .json> [preset dtype batch lx prompt_frames mode]"""
import json, re, sys
src, dst = sys.argv[1], sys.argv[2]
cfg = dict(zip(("preset", "dtype", "batch", "lx", "prompt_frames", "mode"), sys.argv[3:9]))
cfg = {"preset": cfg.get("preset", "giga830M"), "dtype": cfg.get("dtype", "bf16"), "batch": int(cfg.get("batch", 1)),
       "lx": int(cfg.get("lx", 80)), "prompt_frames": int(cfg.get("prompt_frames", 150)), "mode": cfg.get("mode", "tts")}
d = 2048
alg = {"ffn1": 4 * d * d * 2 + d * 4 + 4 * d * 2, "ffn2": 4 * d * d * 2 + 4 * d * 2 + d * 4, "qkv": 3 * d * d * 2 + d * 4 + 3 * d * 2,
       "oproj": d * d * 2 + d * 4 * 2}
pat = {"ffn1": r"rows_gemm_k<bf16_t, 16, 0, 2,", "ffn2": r"rows_gemm_k<bf16_t, 16, 1, 1,", "qkv": r"rows_gemm_k<bf16_t, 16, 0, 0,",
       "oproj": r"rows_gemm_k<bf16_t, 8, 2, 1,"}      # (template tail: tiles per workgroup, non-temporal)
out = {"source": f"{src} (rocprofv3 --pmc FETCH_SIZE --kernel-trace, own pass of `python bench.py --steps 1 --warmup 0`; FETCH_SIZE[KB] * 1024 * 2)",
       "config": cfg, "kernels": {}}
for line in open(src):
    for k, p in pat.items():
        if line.startswith(p) and "FETCH_SIZE" in line:
            f = line.split("FETCH_SIZE")[1].split()
            out["kernels"][k] = {"name": p, "calls": int(f[0]), "fetch_bytes_per_launch": int(round(2 * float(f[1]) * 1024)),
                                 "algorithmic_bytes": alg[k]}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
