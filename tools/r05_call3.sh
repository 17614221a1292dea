#!/bin/bash
# round 5, GPU call 3: prefetch roles re-measured on the trailing-slice layout (second box); box probe + sampler stamps
set -u
export TMPDIR=/tmp
O=gpurun_out
echo "== probe"; timeout 200 python tools/box_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/r05c_box_probe.log
echo "== quick parity"; timeout 300 python -m pytest tests/test_gpu_one_row.py tests/test_gpu_options.py -x -q 2>&1 | tail -3
echo "== giga830M"; timeout 400 python tools/ab_sweep.py --kernels attn_pf=0:8,0,-1 gemm_pf=0:128,16,0 gemm_pf=0:128,24,0 gemm_pf=0:64,16,0 2>&1 | grep -v amdgpu.ids | tee $O/r05c_ab_830M.log
echo "== giga830M, attention role off"; timeout 400 python tools/ab_sweep.py --set attn_pf=0 gemm_pf=0:128,16,0 gemm_pf=0:128,32,0 attn_fast=0:1 2>&1 | grep -v amdgpu.ids | tee $O/r05c_ab_830M_apf0.log
echo "== giga330M"; timeout 400 python tools/ab_sweep.py --preset giga330M attn_pf=0:8,0,-1 gemm_pf=0:128,32,0 gemm_pf=0:128,16,0 2>&1 | grep -v amdgpu.ids | tee $O/r05c_ab_330M.log
echo "== giga830M editing"; timeout 300 python bench.py --mode edit --steps 3 --warmup 1 --no-cpu-baseline --no-codec --ab attn_pf=0:8,0,-1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['decode_ms_per_token_step'], d['prefill_ms'], d['ab'])" | tee $O/r05c_edit.log
