#!/bin/bash
# Round 4, first GPU call (from the repo root on the GPU box):
#   gpurun --timeout 1500 -- 'bash tools/r04_call1.sh > gpurun_out/r04a_call1.log 2>&1; tail -40 gpurun_out/r04a_call1.log'
# GPU suite (incl. the giga330M, finished-row, options and single-GPU two-rank tests), the default line with the in-process
# A/B of the attention-launch prefetch, giga330M lines (C2 and C1), the 8-row step with the A/B of the finished-row form,
# and a kernel trace of the 8-row step.
set -u
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
echo "== GPU suite"; date
timeout 1100 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 | tee $O/r04a_pytest_gpu.log
echo "== default line + A/B of the attention-launch prefetch"; date
timeout 400 python bench.py --steps 5 --warmup 2 --ab attn_pf=0:8,0,32 --ab-pairs 9 2>$O/r04a_bench.err | tail -1 > $O/r04a_bench.json.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04a_bench.json.log"))
print({k:d[k] for k in ("value","ms_per_step","decode_ms_per_token_step","prefill_ms")}, d.get("ab"), d["roofline"]["frac"], d["decode_step"])
PY
echo "== giga330M: C2 (batch 1, 16 s) and C1 (Lx 40, greedy, 150 -> 250)"; date
timeout 400 python bench.py --preset giga330M --steps 5 --warmup 2 --no-codec 2>>$O/r04a_bench.err | tail -1 > $O/r04a_bench_presetgiga330M.json.log
timeout 400 python bench.py --preset giga330M --lx 40 --top-k 1 --steps 5 --warmup 2 --no-codec 2>>$O/r04a_bench.err | tail -1 > $O/r04a_bench_presetgiga330M_c1.json.log
python - <<'PY'
import json
for f in ("r04a_bench_presetgiga330M","r04a_bench_presetgiga330M_c1"):
    d=json.load(open(f"gpurun_out/{f}.json.log"))
    print(f, {k:d[k] for k in ("value","ms_per_step","decode_ms_per_token_step","prefill_ms")}, d["roofline"]["frac"], d["decode_step"], (d.get("cpu_baseline") or {}).get("value"))
PY
echo "== 8 / 4 / 2 rows per step: finished-row form off : on"; date
for b in 8 4 2; do
  timeout 300 python bench.py --batch $b --steps 3 --warmup 1 --no-cpu-baseline --no-codec --ab finished_rows=0:8 --ab-pairs 7 2>>$O/r04a_bench.err | tail -1 > $O/r04a_bench_batch$b.json.log
  python - <<PY
import json
d=json.load(open("gpurun_out/r04a_bench_batch$b.json.log"))
print("batch $b", {k:d[k] for k in ("value","ms_per_step","decode_ms_per_token_step","prefill_ms")}, d.get("ab"), d["kernels"])
PY
done
echo "== kernel trace, 8 rows (finished-row form)"; date
bash tools/prof_decode.sh r04a_b8 --batch 8 --no-codec
head -30 $O/r04a_b8_rocprof_kernel_stats.txt
echo "== kernel trace, 8 rows, form off"
VC_FINISHED_ROWS=0 bash tools/prof_decode.sh r04a_b8_off --batch 8 --no-codec
head -16 $O/r04a_b8_off_rocprof_kernel_stats.txt
date
