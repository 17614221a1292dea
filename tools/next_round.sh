#!/bin/bash
# First GPU call of a round (from the repo root on the GPU box; ~12 minutes):
#   gpurun --timeout 1500 -- 'bash tools/next_round.sh > gpurun_out/next_round.log 2>&1; tail -60 gpurun_out/next_round.log'
# Re-establishes the state round 4 ended in: 200 GPU tests (199 pass, the 2-GPU RCCL one skips); batch 1 6.8-7.2 k codec-tok/s by
# box (decode step 0.552-0.582 ms); 8 / 16 / 32 / 64 utterances per GPU 42.9 k / 70.0 k / 90.0 k / 119 k; giga330M 7.4 k; editing
# 5.5-5.7 k (800-row prefill 3.4 ms).  Every default-on launch-shape feature carries its in-process A/B in the line (`ab`).
# 200 GPU tests since the second prefill attention kernel (tests/test_gpu_attn64.py prints both kernels' times at 512 / 800 / 2048 rows:
# 14.3 / 24.6 / 86.7 us against 15.5 / 21.0 / 52.5 us).
set -u
export TMPDIR=/tmp
TAG=next bash tools/r04_final.sh
echo "== the round-4 A/Bs, one each (in-process, interleaved pairs)"
B="--steps 3 --warmup 1 --no-cpu-baseline --no-codec --ab-pairs 7"
for ab in nt=28:63 attn_pf=0:8,0,-1 attn_pf_cut=0,0:400,700,128; do
  timeout 300 python bench.py $B --ab $ab 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch 1', d['ab'])"
done
timeout 300 python bench.py --batch 8 $B --ab attn_nt=0:2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch 8', d['ab'])"
