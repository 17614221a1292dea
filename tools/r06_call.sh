set -u
export TMPDIR=/tmp
O=gpurun_out
START=$(date +%s)
TAG=r06k SHORT=1 bash tools/round_check.sh
echo "== round_check took $(( $(date +%s) - START )) s"
for B in 8 16 32 64; do
  timeout 600 python bench.py --batch $B --no-codec 2>/dev/null | tail -1 > $O/r06k_bench_batch$B.json.log
  python - <<PY
import json
d=json.loads(open("gpurun_out/r06k_bench_batch$B.json.log").read())
print($B, {k:d[k] for k in ("value","ms_per_step","decode_ms_per_token_step","prefill_ms")}, d["decode_step"], "ab", {k:d.get("ab",{}).get(k) for k in ("knob","median_delta_pct","spread_pct")})
PY
done
echo "== default bench wall"; ( time python bench.py > /dev/null 2>&1 ) 2>&1 | grep real
