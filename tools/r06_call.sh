set -u
export TMPDIR=/tmp
O=gpurun_out
export TAG=r06d
SHORT=1 bash tools/round_check.sh
echo "== 64 rows traced"; bash tools/prof_decode.sh r06d_b64 --batch 64 --no-codec --ab none --no-configs; head -16 $O/r06d_b64_rocprof_kernel_stats.txt
echo "== 32 rows traced"; bash tools/prof_decode.sh r06d_b32 --batch 32 --no-codec --ab none --no-configs; head -14 $O/r06d_b32_rocprof_kernel_stats.txt
echo "== 8 rows traced"; bash tools/prof_decode.sh r06d_b8 --batch 8 --no-codec --ab none --no-configs; head -12 $O/r06d_b8_rocprof_kernel_stats.txt
for B in 16 32 64; do
echo "== bench batch $B"; timeout 600 python bench.py --batch $B --steps 2 --warmup 1 --no-codec --no-configs --no-cpu-baseline 2>/dev/null | tail -1 > $O/r06d_bench_batch$B.json.log
python - <<PY
import json
d=json.loads(open("gpurun_out/r06d_bench_batch$B.json.log").read())
print({k:d[k] for k in ("value","ms_per_step","decode_ms_per_token_step","prefill_ms")}, d["decode_step"])
print("ab", {k:d["ab"].get(k) for k in ("knob","A_ms_median","B_ms_median","median_delta_pct","spread_pct","error")})
PY
done
