export TMPDIR=/tmp
timeout 300 python tools/pf_gemm_probe.py 2>&1 | grep pf_gemm
for cfg in "--mode edit" "--batch 8" ""; do
timeout 200 python bench.py $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-codec 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['value'], 'step', d['decode_ms_per_token_step'], 'prefill', d['prefill_ms'], d.get('prefill_roofline',{}).get('frac'), d.get('prefill_roofline',{}).get('at_2048_rows',{}).get('frac'))"
done
timeout 900 python -m pytest tests/test_gpu_scale.py tests/test_gpu_model.py tests/test_gpu_forward.py -x -q -m gpu 2>&1 | tail -5
