#!/bin/bash
# Round 4, third GPU call: the non-temporal hint per matrix (A = the mix the compiler used to leave: FFN + heads-1 only; B = every
# matrix), the start delay of the attention launch's prefetch workgroups, and the state after removing the sampler prefetch role.
set -u
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
echo "== quick parity subset"; date
timeout 600 python -m pytest tests/test_gpu_options.py tests/test_gpu_model.py -m gpu -q -x -k "options or finished or bf16_teacher or greedy_tokens or giga830M_logits" 2>&1 | tail -4
show() { python - "$1" <<'PY'
import json, sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1], {k:d[k] for k in ("value","ms_per_step","decode_ms_per_token_step","prefill_ms")}, d.get("ab"), d["config"]["engine_options"])
PY
}
B="--steps 3 --warmup 1 --no-cpu-baseline --no-codec --ab-pairs 7"
echo "== batch 1: non-temporal hint, legacy mix : every matrix; none : every matrix"; date
timeout 300 python bench.py $B --ab nt=28:63 2>>$O/r04c.err | tail -1 > $O/r04c_bench_nt28_63.json.log; show $O/r04c_bench_nt28_63.json.log
timeout 300 python bench.py $B --ab nt=0:63 2>>$O/r04c.err | tail -1 > $O/r04c_bench_nt0_63.json.log; show $O/r04c_bench_nt0_63.json.log
for m in 29 30 60; do      # + QKV only, + out-projection only, + heads-2 only
  timeout 300 python bench.py $B --ab nt=28:$m 2>>$O/r04c.err | tail -1 > $O/r04c_bench_nt28_$m.json.log; show $O/r04c_bench_nt28_$m.json.log
done
echo "== batch 1: prefetch workgroups start late"; date
for dly in 4 8 16; do
  timeout 300 python bench.py $B --ab attn_pf=8,0,32:8,0,32,$dly 2>>$O/r04c.err | tail -1 > $O/r04c_bench_apf_delay$dly.json.log; show $O/r04c_bench_apf_delay$dly.json.log
done
timeout 300 python bench.py $B --ab attn_pf=8,0,32:8,0,48,8 2>>$O/r04c.err | tail -1 > $O/r04c_bench_apf_48_delay8.json.log; show $O/r04c_bench_apf_48_delay8.json.log
echo "== 8 rows and giga330M"; date
timeout 300 python bench.py --batch 8 $B --ab nt=28:63 2>>$O/r04c.err | tail -1 > $O/r04c_bench_batch8_nt.json.log; show $O/r04c_bench_batch8_nt.json.log
timeout 300 python bench.py --preset giga330M $B --ab nt=28:63 2>>$O/r04c.err | tail -1 > $O/r04c_bench_330_nt.json.log; show $O/r04c_bench_330_nt.json.log
echo "== kernel trace, batch 1"; date
bash tools/prof_decode.sh r04c --no-codec --ab none
head -12 $O/r04c_rocprof_kernel_stats.txt
date
