#!/bin/bash
# Round 4, reference lines of the final build (from the repo root on the GPU box):
#   gpurun --timeout 1500 -- 'bash tools/r04_final.sh > gpurun_out/${TAG:-r04i}_final.log 2>&1; tail -60 gpurun_out/${TAG:-r04i}_final.log'
set -u
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
show() { python - "$1" <<'PY'
import json, sys
d=json.load(open(sys.argv[1]))
ab=d.get("ab") or {}
print(sys.argv[1].split("/")[-1], "tok/s", d["value"], "ms/call", d["ms_per_step"], "step", d["decode_ms_per_token_step"], "prefill", d["prefill_ms"], "roofline", d["roofline"]["frac"], "step-frac", d["decode_step"]["hbm_frac_in_loop"],
      "| ab", ab.get("knob"), ab.get("A"), "->", ab.get("B"), ab.get("A_ms_median"), ab.get("B_ms_median"), ab.get("median_delta_pct"), "+-", ab.get("spread_pct"), "| cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
}
echo "== GPU suite"; date
timeout 1100 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | tee $O/${TAG:-r04i}_pytest_gpu.log
echo "== smoke"; date
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/${TAG:-r04i}_smoke.log
echo "== default line (exactly what the driver runs)"; date
timeout 500 python bench.py 2>$O/${TAG:-r04i}_bench.err | tail -1 > $O/${TAG:-r04i}_bench.json.log; show $O/${TAG:-r04i}_bench.json.log
timeout 500 python bench.py --steps 5 --warmup 2 2>>$O/${TAG:-r04i}_bench.err | tail -1 > $O/${TAG:-r04i}_bench_steps5.json.log; show $O/${TAG:-r04i}_bench_steps5.json.log
echo "== giga330M: C2, C1"; date
timeout 400 python bench.py --preset giga330M --steps 5 --warmup 2 --no-codec 2>>$O/${TAG:-r04i}_bench.err | tail -1 > $O/${TAG:-r04i}_bench_presetgiga330M.json.log; show $O/${TAG:-r04i}_bench_presetgiga330M.json.log
timeout 400 python bench.py --preset giga330M --lx 40 --top-k 1 --steps 5 --warmup 2 --no-codec 2>>$O/${TAG:-r04i}_bench.err | tail -1 > $O/${TAG:-r04i}_bench_presetgiga330M_c1.json.log; show $O/${TAG:-r04i}_bench_presetgiga330M_c1.json.log
echo "== editing (C4), 8 / 16 / 32 / 64 utterances per GPU"; date
timeout 300 python bench.py --mode edit --steps 3 --warmup 1 --no-cpu-baseline --no-codec 2>>$O/${TAG:-r04i}_bench.err | tail -1 > $O/${TAG:-r04i}_bench_modeedit.json.log; show $O/${TAG:-r04i}_bench_modeedit.json.log
for b in 8 16 32 64; do
  timeout 400 python bench.py --batch $b --steps 3 --warmup 1 --no-cpu-baseline --no-codec 2>>$O/${TAG:-r04i}_bench.err | tail -1 > $O/${TAG:-r04i}_bench_batch$b.json.log; show $O/${TAG:-r04i}_bench_batch$b.json.log
done
echo "== kernel traces: batch 1, 8 rows"; date
bash tools/prof_decode.sh ${TAG:-r04i} --no-codec --ab none; head -12 $O/${TAG:-r04i}_rocprof_kernel_stats.txt
bash tools/prof_decode.sh ${TAG:-r04i}_b8 --batch 8 --no-codec --ab none; head -12 $O/${TAG:-r04i}_b8_rocprof_kernel_stats.txt
echo "== FETCH_SIZE pass, batch 1 (pmc_traffic.json)"; date
bash tools/prof_pmc.sh ${TAG:-r04i}_fetch "FETCH_SIZE" --ab none; head -10 $O/${TAG:-r04i}_fetch_pmc.txt
python tools/pmc_to_json.py $O/${TAG:-r04i}_fetch_pmc.txt $O/pmc_traffic.json
date
