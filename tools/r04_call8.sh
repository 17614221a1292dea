#!/bin/bash
# Round 4, eighth GPU call: where along the context does the attention-launch prefetch pay?  Off : on at three context ranges and on
# the editing workload; then the per-graph cut policy (half length from p1, none from p2) against the uncut default.
set -u
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
show() { python - "$1" <<'PY'
import json, sys
d=json.load(open(sys.argv[1]))
ab=d.get("ab") or {}
print(sys.argv[1].split("/")[-1], "step", d["decode_ms_per_token_step"], "| ab", ab.get("knob"), ab.get("A"), "->", ab.get("B"), ab.get("A_ms_median"), ab.get("B_ms_median"), "delta", ab.get("median_delta_pct"), "+-", ab.get("spread_pct"), "hr", ab.get("half_range_pct"))
PY
}
timeout 300 python -m pytest tests/test_gpu_options.py tests/test_gpu_model.py -m gpu -q -x -k "options_do_not or greedy_tokens or graph_equals" 2>&1 | tail -2
B="--steps 3 --warmup 1 --no-cpu-baseline --no-codec --ab-pairs 7"
echo "== off : on by context range (positions at the first .. last step)"; date
timeout 200 python bench.py $B --lx 20 --prompt-frames 20 --ab attn_pf=0:8,0,-1 2>>$O/r04h.err | tail -1 > $O/r04h_ctx_41_221.json.log; show $O/r04h_ctx_41_221.json.log
timeout 200 python bench.py $B --lx 40 --prompt-frames 100 --ab attn_pf=0:8,0,-1 2>>$O/r04h.err | tail -1 > $O/r04h_ctx_141_441.json.log; show $O/r04h_ctx_141_441.json.log
timeout 200 python bench.py $B --lx 60 --prompt-frames 400 --ab attn_pf=0:8,0,-1 2>>$O/r04h.err | tail -1 > $O/r04h_ctx_461_661.json.log; show $O/r04h_ctx_461_661.json.log
timeout 200 python bench.py $B --lx 80 --prompt-frames 650 --ab attn_pf=0:8,0,-1 2>>$O/r04h.err | tail -1 > $O/r04h_ctx_731_884.json.log; show $O/r04h_ctx_731_884.json.log
timeout 200 python bench.py $B --mode edit --ab attn_pf=0:8,0,-1 2>>$O/r04h.err | tail -1 > $O/r04h_edit.json.log; show $O/r04h_edit.json.log
timeout 200 python bench.py $B --ab attn_pf=0:8,0,-1 2>>$O/r04h.err | tail -1 > $O/r04h_tts.json.log; show $O/r04h_tts.json.log
echo "== the cut policy against the uncut default: TTS, editing"; date
for c in 500,800 400,700 600,900 300,600; do
  timeout 200 python bench.py $B --ab attn_pf_cut=0,0:$c 2>>$O/r04h.err | tail -1 > $O/r04h_tts_cut_$c.json.log; show $O/r04h_tts_cut_$c.json.log
done
for c in 500,800 400,700; do
  timeout 200 python bench.py $B --mode edit --ab attn_pf_cut=0,0:$c 2>>$O/r04h.err | tail -1 > $O/r04h_edit_cut_$c.json.log; show $O/r04h_edit_cut_$c.json.log
done
date
