#!/bin/bash
# round 5, GPU call 1: the one-row forms (fr_one / ln_trim / attn_fast) - parity first, then in-process A/Bs on giga830M and giga330M
set -u
export TMPDIR=/tmp
O=gpurun_out
echo "== new tests"; timeout 400 python -m pytest tests/test_gpu_one_row.py -x -q 2>&1 | tail -8
echo "== giga830M one-row A/Bs"; timeout 300 python tools/ab_sweep.py --kernels --set fr_one=0 --set ln_trim=0 --set attn_fast=0 ln_trim=0:1 fr_one=0:1 attn_fast=0:1 2>&1 | grep -v amdgpu.ids | tee $O/r05a_ab_830M.log
echo "== giga330M one-row A/Bs"; timeout 300 python tools/ab_sweep.py --preset giga330M --kernels --set fr_one=0 --set ln_trim=0 --set attn_fast=0 ln_trim=0:1 fr_one=0:1 attn_fast=0:1 2>&1 | grep -v amdgpu.ids | tee $O/r05a_ab_330M.log
echo "== whole GPU suite"; timeout 500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $O/r05a_pytest_gpu.log
