#!/bin/bash
# round 5, GPU call 2: attention with one scalar batch / role from blockIdx; GEMM-hosted prefetch roles (gemm_pf); the default bench line with `configs`
set -u
export TMPDIR=/tmp
O=gpurun_out
echo "== quick parity"; timeout 500 python -m pytest tests/test_gpu_one_row.py tests/test_gpu_options.py tests/test_gpu_attn64.py -x -q 2>&1 | tail -6
echo "== giga830M"; timeout 400 python tools/ab_sweep.py --kernels gemm_pf=0:128,32,0 gemm_pf=0:256,32,0 gemm_pf=0:256,64,0 gemm_pf=0:128,16,0 attn_pf=0:8,0,-1 2>&1 | grep -v amdgpu.ids | tee $O/r05b_ab_830M.log
echo "== giga330M"; timeout 400 python tools/ab_sweep.py --preset giga330M --kernels gemm_pf=0:128,32,0 gemm_pf=0:128,64,0 gemm_pf=0:128,0,24 gemm_pf=0:128,64,24 fr_one=1:2 attn_pf=0:8,0,-1 2>&1 | grep -v amdgpu.ids | tee $O/r05b_ab_330M.log
echo "== default bench line"; timeout 600 python bench.py 2>/dev/null | tail -1 > $O/r05b_bench.json.log; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05b_bench.json.log").read())
print({k:d[k] for k in ("value","ms_per_step","decode_ms_per_token_step","prefill_ms")}, d["roofline"]["frac"], d["decode_step"], d.get("ab"))
print(json.dumps(d.get("configs"), indent=1))
print(d["kernels"])
PY
