#!/bin/bash
# First GPU call of a round (from the repo root on the GPU box; ~8 minutes):
#   gpurun --timeout 1200 -- 'bash tools/next_round.sh > gpurun_out/next_round.log 2>&1; tail -60 gpurun_out/next_round.log'
# Re-establishes the state the previous round ended in (round 3: 178 GPU tests green; batch 1 6.6-6.7 k tok/s, 0.587-0.598 ms per
# step; 8 rows 0.82-0.90 ms box to box; C4 prefill 3.3-3.45 ms; C5 prefill 5.3 ms; FFN-up probe 0.27 / 0.44 of the bf16 peak at
# 512 / 2048 rows; tile attention 14.7 / 24.9 / 89.5 us at 512 / 800 / 2048 rows) and re-measures the two prefetch switches.
set -u
export TMPDIR=/tmp
echo "== GPU suite"
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4
echo "== default bench line"
timeout 300 python bench.py --steps 5 --warmup 2 2>/dev/null | tail -c 3000
echo
echo "== piggyback prefetch, batch 1 (off / default)"
bash tools/apf_sweep.sh 0 8,0,32 0 8,0,32
echo "== piggyback prefetch on the LayerNorm launches, 8 rows (off / default)"
bash tools/lpf_sweep.sh 0 248,24,24 0 248,24,24
echo "== prefill GEMM and attention by pass size"
timeout 200 python tools/pf_gemm_probe.py 2>&1 | grep pf_gemm
timeout 200 python tools/pf_attn_probe.py 2>&1 | grep pf_attn
echo "== editing (C4) and 32 rows"
for cfg in "--mode edit" "--batch 32"; do
  timeout 300 python bench.py $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-codec 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['value'], 'step', d['decode_ms_per_token_step'], 'prefill', d['prefill_ms'])"
done
