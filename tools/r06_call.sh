set -u
export TMPDIR=/tmp
O=gpurun_out
echo "== new tests (durations)"; timeout 1500 python -m pytest tests/test_gpu_options.py tests/test_gpu_scale.py -m gpu -q -x --durations=30 -k "retire or 33_to_64 or sixty_four or best_of_n_at or thirty_two or qkv16" 2>&1 | tail -45 | tee $O/r06b_pytest_new.log
echo "== wd probe"; timeout 400 python tools/wd_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/r06b_wd_probe.log
echo "== default bench line"; timeout 1200 python bench.py 2>/dev/null | tail -1 > $O/r06b_bench.json.log; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06b_bench.json.log").read())
print({k:d[k] for k in ("value","ms_per_step","decode_ms_per_token_step","prefill_ms")}, d["roofline"]["frac"], d["roofline"].get("isolated_frac"), d["roofline"]["measured"][:40], d["decode_step"])
print("ab", {k:d["ab"].get(k) for k in ("knob","A_ms_median","B_ms_median","median_delta_pct","spread_pct")})
for r in d.get("ab_more", []): print("  ", {k:r.get(k) for k in ("knob","A","B","median_delta_pct","spread_pct","error")})
print({k:(v.get("value"), v.get("decode_ms_per_step"), v.get("hbm_frac_in_loop"), v.get("error")) for k,v in d.get("configs",{}).items()})
print(json.dumps(d.get("ragged"), indent=1))
PY
