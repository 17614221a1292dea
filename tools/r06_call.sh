set -u
export TMPDIR=/tmp
O=gpurun_out
( time python bench.py 2>/dev/null | tail -1 > $O/r06n_bench.json.log ) 2>&1 | grep real
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06n_bench.json.log").read())
print({k:d[k] for k in ("value","ms_per_step","decode_ms_per_token_step","prefill_ms")})
r=d["roofline"]; print({k:r.get(k) for k in ("frac","achieved","traffic","traffic_source","isolated_frac","avg_launch_us","measured")}); print(r.get("in_situ"))
print(json.dumps(d["ragged"])[:1500])
PY
