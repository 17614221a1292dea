#!/bin/bash
# Round 4, fourth GPU call: whole suite on the template-NT build; leave-one-out of the per-matrix hint; K/V hint of the decode
# attention; the wide-decode kernel with / without the hint (twin library); the sampler's time at batch 1 (stamps + trace histogram).
set -u
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
echo "== GPU suite"; date
timeout 1100 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | tee $O/r04d_pytest_gpu.log
show() { python - "$1" <<'PY'
import json, sys
d=json.load(open(sys.argv[1]))
ab=d.get("ab") or {}
print(sys.argv[1].split("/")[-1], d["decode_ms_per_token_step"], d["value"], "|", ab.get("knob"), ab.get("A"), "->", ab.get("B"), "A", ab.get("A_ms_median"), "B", ab.get("B_ms_median"), "delta", ab.get("median_delta_pct"), "+-", ab.get("spread_pct"), "|", d["config"]["engine_options"])
PY
}
B="--steps 3 --warmup 1 --no-cpu-baseline --no-codec --ab-pairs 7"
echo "== batch 1: leave one matrix out of the hint (A = all, B = all but one)"; date
for m in 62 61 59 55 47 31; do
  timeout 300 python bench.py $B --ab nt=63:$m 2>>$O/r04d.err | tail -1 > $O/r04d_bench_nt63_$m.json.log; show $O/r04d_bench_nt63_$m.json.log
done
echo "== K/V hint of the decode attention: batch 1, 8, 32"; date
timeout 300 python bench.py $B --ab attn_nt=0:1 2>>$O/r04d.err | tail -1 > $O/r04d_bench_attn_nt.json.log; show $O/r04d_bench_attn_nt.json.log
timeout 300 python bench.py --batch 8 $B --ab attn_nt=0:1 2>>$O/r04d.err | tail -1 > $O/r04d_bench_batch8_attn_nt.json.log; show $O/r04d_bench_batch8_attn_nt.json.log
timeout 300 python bench.py --batch 32 $B --ab attn_nt=0:1 2>>$O/r04d.err | tail -1 > $O/r04d_bench_batch32_attn_nt.json.log; show $O/r04d_bench_batch32_attn_nt.json.log
echo "== 32 rows: wide-decode kernel without : with the hint (twin library, alternating processes)"; date
for rep in 1 2 3; do
  for lib in libvcengine_mtnt0.so libvcengine.so; do
    VC_ENGINE_LIB=$PWD/voicecraft_amd/$lib timeout 300 python bench.py --batch 32 --steps 3 --warmup 1 --no-cpu-baseline --no-codec --ab none 2>>$O/r04d.err | tail -1 > $O/r04d_tmp.json
    python - $lib <<'PY'
import json, sys
d=json.load(open("gpurun_out/r04d_tmp.json")); print("batch 32", sys.argv[1], d["decode_ms_per_token_step"], d["value"])
PY
  done
done 2>&1 | tee $O/r04d_batch32_mt_nt_twin.log
echo "== sampler at batch 1: in-kernel stamps, then the trace histogram"; date
timeout 200 python tools/sampler_ts.py 1 2>&1 | tail -8 | tee $O/r04d_sampler_stamps_b1.log
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_r04d -o kt -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-codec --ab none > $OLDPWD/$O/r04d_bench_under_rocprof.log 2>&1 )
db=$(find /tmp/prof_r04d -name "*.db" | head -1)
python tools/rocprof_summary.py $db $O/r04d_rocprof_kernel_stats.txt; head -11 $O/r04d_rocprof_kernel_stats.txt
python tools/rocprof_hist.py $db sample_fused_k 8 4 | tee $O/r04d_sampler_hist.txt
python tools/rocprof_hist.py $db "rows_gemm_k<bf16_t, 16, 0, 3" 8 4 | tee $O/r04d_heads1_hist.txt
date
