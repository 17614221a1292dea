set -u
export TMPDIR=/tmp
O=gpurun_out
echo "== new tests first"; timeout 1200 python -m pytest tests/test_gpu_options.py tests/test_gpu_scale.py tests/test_gpu_multi.py -m gpu -q -x 2>&1 | tail -25 | tee $O/r06a_pytest_new.log
echo "== wd probe"; timeout 300 python tools/wd_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/r06a_wd_probe.log
for B in 64 32; do
echo "== bench batch $B"; timeout 600 python bench.py --batch $B --steps 2 --warmup 1 --no-codec --no-configs --no-cpu-baseline --ab wide_gemm=0:1 2>/dev/null | tail -1 > $O/r06a_bench_batch$B.json.log
python - <<PY
import json
d=json.loads(open("gpurun_out/r06a_bench_batch$B.json.log").read())
print({k:d[k] for k in ("value","ms_per_step","decode_ms_per_token_step","prefill_ms")}, d["decode_step"])
print("ab", {k:d["ab"].get(k) for k in ("knob","A_ms_median","B_ms_median","median_delta_pct","spread_pct","error")})
PY
done
echo "== rest of the suite"; timeout 1200 python -m pytest tests -m gpu -q --deselect tests/test_gpu_options.py --deselect tests/test_gpu_scale.py --deselect tests/test_gpu_multi.py 2>&1 | tail -8 | tee $O/r06a_pytest_rest.log
