#!/bin/bash
# Round 4, sixth GPU call: whole suite on the 2..16-row finished-row form, its A/B at 12 and 16 rows, prefetch length at giga330M.
set -u
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
show() { python - "$1" <<'PY'
import json, sys
d=json.load(open(sys.argv[1]))
ab=d.get("ab") or {}
print(sys.argv[1].split("/")[-1], d["decode_ms_per_token_step"], d["value"], "|", ab.get("knob"), ab.get("A"), "->", ab.get("B"), "A", ab.get("A_ms_median"), "B", ab.get("B_ms_median"), "delta", ab.get("median_delta_pct"), "+-", ab.get("spread_pct"), "|", d["config"]["engine_options"])
PY
}
echo "== GPU suite"; date
timeout 1100 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 | tee $O/r04f_pytest_gpu.log
B="--steps 3 --warmup 1 --no-cpu-baseline --no-codec --ab-pairs 7"
echo "== 12 / 16 rows: finished-row form off : on"; date
for b in 16 12 9; do
  timeout 300 python bench.py --batch $b $B 2>>$O/r04f.err | tail -1 > $O/r04f_bench_batch$b.json.log; show $O/r04f_bench_batch$b.json.log
done
echo "== giga330M: prefetch length (default = half a tile = 16 KB)"; date
for v in 8,0,8 8,0,24 4,0,16 16,0,16; do
  timeout 300 python bench.py --preset giga330M $B --ab attn_pf=8,0,16:$v 2>>$O/r04f.err | tail -1 > $O/r04f_bench_330_apf_$v.json.log; show $O/r04f_bench_330_apf_$v.json.log
done
timeout 300 python bench.py --preset giga330M $B 2>>$O/r04f.err | tail -1 > $O/r04f_bench_330.json.log; show $O/r04f_bench_330.json.log
date
