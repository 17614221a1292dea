#!/bin/bash
# Round 4, second GPU call: suite on the two-tile consumers, A/Bs (lnw_tiles at 8 rows, samp_pf at batch 1), in-kernel stamps
# at giga330M / giga830M, a batch-1 kernel trace, and the counter passes (L2 hit / miss of the FFN-up launch with the
# attention-launch prefetch on / off; MFMA busy of the 256 x 256 prefill GEMM).
set -u
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
echo "== GPU suite"; date
timeout 1100 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 | tee $O/r04b_pytest_gpu.log
show() { python - "$1" <<'PY'
import json, sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1], {k:d[k] for k in ("value","ms_per_step","decode_ms_per_token_step","prefill_ms")}, d.get("ab"), d["config"]["engine_options"])
PY
}
echo "== 8 rows: one : two tiles per consumer workgroup"; date
timeout 300 python bench.py --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-codec --ab lnw_tiles=1:2 --ab-pairs 7 2>>$O/r04b_bench.err | tail -1 > $O/r04b_bench_batch8.json.log; show $O/r04b_bench_batch8.json.log
timeout 300 python bench.py --batch 4 --steps 3 --warmup 1 --no-cpu-baseline --no-codec --ab lnw_tiles=1:2 --ab-pairs 5 2>>$O/r04b_bench.err | tail -1 > $O/r04b_bench_batch4.json.log; show $O/r04b_bench_batch4.json.log
echo "== batch 1: sampler-launch prefetch off : on"; date
for v in 248,24 248,40 504,24; do
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-codec --ab samp_pf=0:$v --ab-pairs 7 2>>$O/r04b_bench.err | tail -1 > $O/r04b_bench_spf_$v.json.log; show $O/r04b_bench_spf_$v.json.log
done
echo "== batch 1, giga330M: sampler-launch prefetch"; date
timeout 300 python bench.py --preset giga330M --steps 3 --warmup 1 --no-cpu-baseline --no-codec --ab samp_pf=0:248,24 --ab-pairs 7 2>>$O/r04b_bench.err | tail -1 > $O/r04b_bench_330_spf.json.log; show $O/r04b_bench_330_spf.json.log
timeout 300 python bench.py --preset giga330M --steps 3 --warmup 1 --no-cpu-baseline --no-codec --ab attn_pf=0:8,0,32 --ab-pairs 7 2>>$O/r04b_bench.err | tail -1 > $O/r04b_bench_330_apf.json.log; show $O/r04b_bench_330_apf.json.log
echo "== in-kernel stamps"; date
timeout 200 python tools/kernel_ts.py giga330M 1 > $O/r04b_kernel_stamps_giga330M.log 2>&1; grep -v "^#" $O/r04b_kernel_stamps_giga330M.log | grep "first\|last  wg" | cut -c1-260
timeout 200 python tools/kernel_ts.py giga830M 1 > $O/r04b_kernel_stamps_giga830M.log 2>&1; grep "last  wg" $O/r04b_kernel_stamps_giga830M.log | cut -c1-260
echo "== kernel trace, batch 1"; date
bash tools/prof_decode.sh r04b --no-codec
head -14 $O/r04b_rocprof_kernel_stats.txt
echo "== counters: L2 hit / miss with the attention-launch prefetch on, then off"; date
bash tools/prof_pmc.sh r04b_tcc_on "TCC_HIT_sum TCC_MISS_sum"
head -12 $O/r04b_tcc_on_pmc.txt
VC_ATTN_PF=0 bash tools/prof_pmc.sh r04b_tcc_off "TCC_HIT_sum TCC_MISS_sum"
head -12 $O/r04b_tcc_off_pmc.txt
echo "== counters: MFMA busy, 8 utterances (the run that launches rows_gemm_big_k)"; date
bash tools/prof_pmc.sh r04b_big_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" --batch 8
grep "big_k\|blk_k\|tile_attn" $O/r04b_big_mfma_pmc.txt | head -20
bash tools/prof_pmc.sh r04b_big_fetch "FETCH_SIZE" --batch 8
grep "big_k\|rows_gemm_fr\|rows_gemm_k<bf16_t, 16, 3" $O/r04b_big_fetch_pmc.txt | head -20
date
