#!/bin/bash
# The whole-state check of a round: GPU suite, default bench line, rocprofv3 kernel trace + FETCH_SIZE pass of
# the same command, in-kernel stamps (twin library), leftover A/Bs (giga330M finished rows, attention split counts), 8-row kernel trace
# (SHORT=1 stops after the FETCH_SIZE pass)
set -u
export TMPDIR=/tmp
O=gpurun_out
TAG=${TAG:-chk}      # file-name prefix of everything written under gpurun_out/
export TAG
echo "== probe"; timeout 200 python tools/box_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_box_probe.log
echo "== whole GPU suite"; timeout 700 python -m pytest tests -m gpu -q 2>&1 | tail -25 | tee $O/${TAG}_pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/${TAG}_smoke.log
echo "== default bench line"; timeout 900 python bench.py 2>/dev/null | tail -1 > $O/${TAG}_bench.json.log; python - <<'PY'
import json
d=json.loads(open("gpurun_out/" + __import__("os").environ.get("TAG", "chk") + "_bench.json.log").read())
print({k:d[k] for k in ("value","ms_per_step","decode_ms_per_token_step","prefill_ms")}, d["roofline"]["frac"], d["roofline"]["kernel"][:60], d["roofline"]["in_situ"], d["decode_step"])
print("ab", {k:d["ab"].get(k) for k in ("knob","A_ms_median","B_ms_median","median_delta_pct","spread_pct")})
for r in d.get("ab_more", []): print("  ", {k:r.get(k) for k in ("knob","A","B","median_delta_pct","spread_pct","error")})
print("box", d.get("box"), "sampler", d.get("sampler"))
print({k:(v.get("value"), v.get("decode_ms_per_step"), v.get("hbm_frac_in_loop"), v.get("error")) for k,v in d.get("configs",{}).items()})
print(d["kernels"])
PY
echo "== rocprof kernel trace"; bash tools/prof_decode.sh ${TAG} --no-codec --ab none --no-configs; head -14 $O/${TAG}_rocprof_kernel_stats.txt
python tools/in_situ_to_json.py $O/${TAG}_rocprof_kernel_stats.txt $O/in_situ.json > /dev/null
echo "== FETCH_SIZE pass"; bash tools/prof_pmc.sh ${TAG}_fetch FETCH_SIZE --ab none --no-configs; head -12 $O/${TAG}_fetch_pmc.txt
python tools/pmc_to_json.py $O/${TAG}_fetch_pmc.txt $O/pmc_traffic.json > /dev/null
if [ "${SHORT:-0}" = "1" ]; then exit 0; fi      # SHORT=1: suite + smoke + bench + the two profiles only
echo "== in-kernel stamps"; timeout 200 python tools/kernel_ts.py giga830M 1 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_kernel_stamps_giga830M.log
echo "== giga330M"; timeout 300 python tools/ab_sweep.py --preset giga330M fr_one=0:1 qkv_p8=0:1 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_ab_330M.log
echo "== 8 rows"; timeout 300 python tools/ab_sweep.py --batch 8 fr_pair=0:1 finished_rows=0:16 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_ab_b8.log
echo "== wide steps"; timeout 300 python tools/wd_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_wd_probe.log
bash tools/prof_decode.sh ${TAG}_b8 --batch 8 --no-codec --ab none; head -12 $O/${TAG}_b8_rocprof_kernel_stats.txt
