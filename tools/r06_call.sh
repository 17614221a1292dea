set -u
export TMPDIR=/tmp
O=gpurun_out
echo "== wd probe"; timeout 400 python tools/wd_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/r06c_wd_probe.log
echo "== wide tests"; timeout 900 python -m pytest tests/test_gpu_options.py tests/test_gpu_scale.py -m gpu -q -x --durations=12 -k "retire or 33_to_64 or sixty_four or thirty_two or qkv16 or options_do_not" 2>&1 | tail -25 | tee $O/r06c_pytest_wide.log
for B in 64 32; do
echo "== bench batch $B"; timeout 600 python bench.py --batch $B --steps 2 --warmup 1 --no-codec --no-configs --no-cpu-baseline --ab wd_order=0:6 2>/dev/null | tail -1 > $O/r06c_bench_batch$B.json.log
python - <<PY
import json
d=json.loads(open("gpurun_out/r06c_bench_batch$B.json.log").read())
print({k:d[k] for k in ("value","ms_per_step","decode_ms_per_token_step","prefill_ms")}, d["decode_step"])
print("ab", {k:d["ab"].get(k) for k in ("knob","A_ms_median","B_ms_median","median_delta_pct","spread_pct","error")})
PY
done
echo "== whole suite"; timeout 1500 python -m pytest tests -m gpu -q --durations=15 2>&1 | tail -30 | tee $O/r06c_pytest_gpu.log
