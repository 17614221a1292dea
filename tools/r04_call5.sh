#!/bin/bash
# Round 4, fifth GPU call: the sampler with its argument blocks in LDS (A/B at 1 and 8 sequences + stamps), the attention-launch
# prefetch re-swept now that the out-projection no longer evicts what it fetched, then the round's reference lines.
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
echo "== quick parity subset (sampler variant + attention hint)"; date
VC_SAMPLER_LDS=1 timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_sampler.py -m gpu -q -x -k "greedy_tokens or replay or best_of or sampled or seeded or multi_utterance" 2>&1 | tail -3
B="--steps 3 --warmup 1 --no-cpu-baseline --no-codec --ab-pairs 7"
echo "== sampler: argument blocks in LDS, off : on"; date
timeout 300 python bench.py $B --ab sampler_lds=0:1 2>>$O/r04e.err | tail -1 > $O/r04e_bench_sampler_lds.json.log; show $O/r04e_bench_sampler_lds.json.log
timeout 300 python bench.py --batch 8 $B --ab sampler_lds=0:1 2>>$O/r04e.err | tail -1 > $O/r04e_bench_batch8_sampler_lds.json.log; show $O/r04e_bench_batch8_sampler_lds.json.log
VC_SAMPLER_LDS=1 timeout 200 python tools/sampler_ts.py 1 2>&1 | tail -4 | tee $O/r04e_sampler_stamps_b1_lds.log
echo "== attention-launch prefetch, re-swept"; date
for v in 8,0,40 8,0,48 8,0,64 8,16,32 16,0,32 8,0,24; do
  timeout 300 python bench.py $B --ab attn_pf=8,0,32:$v 2>>$O/r04e.err | tail -1 > $O/r04e_bench_apf_$v.json.log; show $O/r04e_bench_apf_$v.json.log
done
echo "== giga330M: attention-launch prefetch variants"; date
for v in 8,0,16 8,0,48; do
  timeout 300 python bench.py --preset giga330M $B --ab attn_pf=8,0,32:$v 2>>$O/r04e.err | tail -1 > $O/r04e_bench_330_apf_$v.json.log; show $O/r04e_bench_330_apf_$v.json.log
done
date
