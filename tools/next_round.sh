#!/bin/bash
# First GPU call of a round (from the repo root on the GPU box; ~11 minutes of box time):
#   gpurun --timeout 1800 -- 'bash tools/next_round.sh > gpurun_out/next_round.log 2>&1; tail -80 gpurun_out/next_round.log'
# Re-establishes the state round 6 ended in (tools/round_check.sh does the work; SHORT=1 stops after the FETCH_SIZE pass):
#   227 GPU tests (226 pass, the 2-GPU RCCL one skips; 6.5 min); smoke(); default bench line 7.6-7.7 k codec-tok/s (decode step 0.514-0.517 ms,
#   whole step 0.42 of the HBM roofline; `configs`: giga330M 7.8 k / 7.7 k, editing 6.3 k, 8 utterances 45.1 k, best-of-3 6.4 k, 32 / 64 utterances
#   116-117 k / 163 k; `ragged`: +4 % / +10 % from the shrinking batch);
#   rocprofv3 of the same command: FFN-up 7.6 + FFN-down 7.5 + QKV 6.5 + out-projection 4.9 + attention 4.8 = 31.3 us per layer; FETCH_SIZE x2
#   within 1.3 % of the algorithmic bytes on the three weight streams.
#   Wide steps: python bench.py --batch 32|64 --no-codec --no-configs --no-cpu-baseline -> 116.7 k / 163.0 k tok/s (1.058 / 1.499 ms per step);
#   isolated kernels of a wide step: python tools/wd_probe.py (64 rows: FFN-up 9.7, FFN-down 9.9, QKV 9.1, out-projection 4.5 us).
# Every default-on launch-shape form carries its in-process A/B in the bench line (`ab`: fr_one at one row, finished_rows at 2..16 rows, wide_gemm
# above; `ab_more`: qkv_p8).  Several A/Bs on one engine: python tools/ab_sweep.py [--preset P] [--batch B] knob=A:B ...
# After a profile run on the FINAL library: cp gpurun_out/in_situ.json gpurun_out/pmc_traffic.json profiles/ (bench.py quotes them only for
# the build whose source digest they carry).
set -u
export TMPDIR=/tmp
bash tools/round_check.sh
