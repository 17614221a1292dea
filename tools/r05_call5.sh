#!/bin/bash
# round 5, GPU call 5: paired producer of 2..8-row steps (fr_pair) - parity + A/B at 8 / 4 rows; QKV 4 vs 8 waves; FFN-up tiles under
# the out-projection launch; giga330M with the finished-row forms forced
set -u
export TMPDIR=/tmp
O=gpurun_out
echo "== parity"; timeout 600 python -m pytest tests/test_gpu_options.py tests/test_gpu_one_row.py "tests/test_gpu_scale.py" -x -q 2>&1 | tail -5
echo "== giga830M batch 8"; timeout 400 python tools/ab_sweep.py --batch 8 --kernels fr_pair=0:1 2>&1 | grep -v amdgpu.ids | tee $O/r05e_ab_b8.log
echo "== giga830M batch 4"; timeout 300 python tools/ab_sweep.py --batch 4 fr_pair=0:1 2>&1 | grep -v amdgpu.ids | tee $O/r05e_ab_b4.log
echo "== giga830M batch 1"; timeout 400 python tools/ab_sweep.py qkv_p8=1:2 gemm_pf=128,-1,0,0:128,-1,0,16 gemm_pf=128,-1,0,0:128,0,0,16 gemm_pf=128,-1,0,0:256,-1,0,0 2>&1 | grep -v amdgpu.ids | tee $O/r05e_ab_830M.log
echo "== giga330M"; timeout 400 python tools/ab_sweep.py --preset giga330M --kernels fr_one=1:2 gemm_pf=128,-1,0,0:128,-1,0,16 2>&1 | grep -v amdgpu.ids | tee $O/r05e_ab_330M.log
