set -u
export TMPDIR=/tmp
O=gpurun_out
( time python bench.py 2>/dev/null | tail -1 > $O/r06t_bench.json.log ) 2>&1 | grep real
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06t_bench.json.log").read())
print({k:d[k] for k in ("value","ms_per_step","decode_ms_per_token_step","prefill_ms")})
r=d["roofline"]; print({k:r.get(k) for k in ("frac","achieved","traffic","isolated_frac","avg_launch_us","lib_stamp")}); print(r.get("in_situ")); print(r.get("traffic_source"))
print({k:(v.get("value"), v.get("decode_ms_per_step")) for k,v in d["configs"].items()})
print({k:v.get("gain_pct") for k,v in d["ragged"].items()}, d["cpu_baseline"]["value"], d["cpu_baseline"]["kind"])
PY
