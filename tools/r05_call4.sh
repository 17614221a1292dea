#!/bin/bash
# round 5, GPU call 4: paired QKV kernel (qkv_p8) - parity, A/B; new defaults (attention role off, gemm_pf on); default bench line;
# rocprofv3 kernel trace + FETCH_SIZE pass of the default command; box probe
set -u
export TMPDIR=/tmp
O=gpurun_out
echo "== probe"; timeout 200 python tools/box_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/r05d_box_probe.log
echo "== parity of the one-row forms"; timeout 400 python -m pytest tests/test_gpu_one_row.py -x -q 2>&1 | tail -4
echo "== giga830M"; timeout 400 python tools/ab_sweep.py --kernels qkv_p8=0:1 fr_one=0:1 gemm_pf=0:128,-1,0 2>&1 | grep -v amdgpu.ids | tee $O/r05d_ab_830M.log
echo "== default bench line"; timeout 900 python bench.py 2>/dev/null | tail -1 > $O/r05d_bench.json.log; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05d_bench.json.log").read())
print({k:d[k] for k in ("value","ms_per_step","decode_ms_per_token_step","prefill_ms")}, d["roofline"]["frac"], d["roofline"]["kernel"][:40], d["decode_step"])
print("ab", {k:d["ab"].get(k) for k in ("knob","A_ms_median","B_ms_median","median_delta_pct","spread_pct")})
for r in d.get("ab_more", []): print("  ", {k:r.get(k) for k in ("knob","A","B","median_delta_pct","spread_pct","error")})
print("box", d.get("box"), "sampler", d.get("sampler"))
print({k:(v.get("value"), v.get("decode_ms_per_step"), v.get("hbm_frac_in_loop"), v.get("error")) for k,v in d.get("configs",{}).items()})
print(d["kernels"])
PY
echo "== rocprof kernel trace"; bash tools/prof_decode.sh r05d --no-codec --ab none --no-configs; head -14 $O/r05d_rocprof_kernel_stats.txt
python tools/in_situ_to_json.py $O/r05d_rocprof_kernel_stats.txt $O/in_situ.json | head -30
echo "== FETCH_SIZE pass"; bash tools/prof_pmc.sh r05d_fetch FETCH_SIZE --ab none --no-configs; head -12 $O/r05d_fetch_pmc.txt
python tools/pmc_to_json.py $O/r05d_fetch_pmc.txt $O/pmc_traffic.json
