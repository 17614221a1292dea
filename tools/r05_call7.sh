#!/bin/bash
# round 5, GPU call 7: lean FFN-up kernel (ffn1_lean) and lean first K/V batch of the attention (attn_fast 3 = the general addressing):
# parity, in-process A/Bs, stamps
set -u
export TMPDIR=/tmp
O=gpurun_out
echo "== parity"; timeout 700 python -m pytest tests/test_gpu_one_row.py tests/test_gpu_model.py -x -q 2>&1 | tail -4
echo "== giga830M"; timeout 400 python tools/ab_sweep.py --kernels ffn1_lean=0:1 attn_fast=3:1 2>&1 | grep -v amdgpu.ids | tee $O/r05g_ab_830M.log
echo "== giga330M"; timeout 400 python tools/ab_sweep.py --preset giga330M --kernels ffn1_lean=0:1 attn_fast=3:1 2>&1 | grep -v amdgpu.ids | tee $O/r05g_ab_330M.log
echo "== batch 8"; timeout 300 python tools/ab_sweep.py --batch 8 attn_fast=3:1 2>&1 | grep -v amdgpu.ids | tee $O/r05g_ab_b8.log
echo "== in-kernel stamps"; timeout 200 python tools/kernel_ts.py giga830M 1 2>&1 | grep -v amdgpu.ids | grep -v hot | tee $O/r05g_kernel_stamps_giga830M.log
