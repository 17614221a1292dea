export TMPDIR=/tmp
for pf in 0 4 8 2; do
  VC_PREFETCH=$pf timeout 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-codec > gpurun_out/r03f_bench_pf$pf.json.log 2>gpurun_out/r03f.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r03f_bench_pf$pf.json.log").read().strip().splitlines()[-1])
print("VC_PREFETCH=$pf", {k:d[k] for k in ("value","ms_per_step","decode_ms_per_token_step","prefill_ms")})
PY
done
VC_PREFETCH=4 timeout 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-codec --batch 8 > gpurun_out/r03f_bench_pf4_b8.json.log 2>gpurun_out/r03f.err; python - <<PY
import json
d=json.loads(open("gpurun_out/r03f_bench_pf4_b8.json.log").read().strip().splitlines()[-1])
print("batch8 pf4", {k:d[k] for k in ("value","ms_per_step","decode_ms_per_token_step","prefill_ms")})
PY
VC_PREFETCH=0 timeout 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-codec --batch 8 > gpurun_out/r03f_bench_pf0_b8.json.log 2>gpurun_out/r03f.err; python - <<PY
import json
d=json.loads(open("gpurun_out/r03f_bench_pf0_b8.json.log").read().strip().splitlines()[-1])
print("batch8 pf0", {k:d[k] for k in ("value","ms_per_step","decode_ms_per_token_step","prefill_ms")})
PY
