#!/bin/bash
# Round 4, tenth GPU call: the A-arms stay correct (parity subsets under the non-default option presets), and where the
# finished-row form should stop splitting the attention.
set -u
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
echo "== parity under non-default presets"; date
VC_FINISHED_ROWS=0 timeout 500 python -m pytest tests/test_gpu_model.py tests/test_gpu_scale.py -m gpu -q -x -k "batch or wide or batched or c5_share or multi_utterance or best_of or edit_3span or replay" 2>&1 | tail -2
VC_NT=0 VC_ATTN_NT=1 VC_ATTN_PF=0 timeout 500 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "bf16_teacher or greedy_tokens or giga830M_logits or batched" 2>&1 | tail -2
VC_ATTN_PF_CUT=0,0 VC_LNW_TILES=1 VC_ATTN_NT=0 timeout 500 python -m pytest tests/test_gpu_model.py tests/test_gpu_scale.py -m gpu -q -x -k "long_context or c5_share or batched or giga830M_long" 2>&1 | tail -2
show() { python - "$1" <<'PY'
import json, sys
d=json.load(open(sys.argv[1]))
ab=d.get("ab") or {}
print(sys.argv[1].split("/")[-1], "step", d["decode_ms_per_token_step"], "| ab", ab.get("knob"), ab.get("A"), "->", ab.get("B"), ab.get("A_ms_median"), ab.get("B_ms_median"), "delta", ab.get("median_delta_pct"), "+-", ab.get("spread_pct"))
PY
}
B="--steps 3 --warmup 1 --no-cpu-baseline --no-codec --ab-pairs 7"
echo "== finished-row form: split attention up to 8 rows : unsplit from 5 / from 7 rows"; date
timeout 300 python bench.py --batch 8 $B --ab fr_split_rows=8:4 2>>$O/r04j.err | tail -1 > $O/r04j_b8_split8_4.json.log; show $O/r04j_b8_split8_4.json.log
timeout 300 python bench.py --batch 6 $B --ab fr_split_rows=8:4 2>>$O/r04j.err | tail -1 > $O/r04j_b6_split8_4.json.log; show $O/r04j_b6_split8_4.json.log
timeout 300 python bench.py --batch 8 $B --ab lnw_tiles=2:1 2>>$O/r04j.err | tail -1 > $O/r04j_b8_lnw2_1.json.log; show $O/r04j_b8_lnw2_1.json.log
echo "== batch 1: cut positions"; date
for c in 450,750,128 400,700,0 350,650,128; do
  timeout 300 python bench.py $B --ab attn_pf_cut=400,700,128:$c 2>>$O/r04j.err | tail -1 > $O/r04j_cut_$c.json.log; show $O/r04j_cut_$c.json.log
done
date
