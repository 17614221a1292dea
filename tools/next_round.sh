#!/bin/bash
# First GPU call of a round (from the repo root on the GPU box; ~9 minutes of box time):
#   gpurun --timeout 1800 -- 'bash tools/next_round.sh > gpurun_out/next_round.log 2>&1; tail -80 gpurun_out/next_round.log'
# Re-establishes the state round 5 ended in (tools/round_check.sh does the work; SHORT=1 stops after the FETCH_SIZE pass):
#   215 GPU tests (214 pass, the 2-GPU RCCL one skips); smoke(); default bench line 7.5-7.7 k codec-tok/s (decode step 0.514-0.524 ms, whole
#   step 0.41-0.42 of the HBM roofline; `configs`: giga330M 7.4-7.9 k / 7.55-7.75 k, editing 6.2-6.4 k, 8 utterances 43.5-45.2 k);
#   rocprofv3 of the same command: FFN-up 7.7 + FFN-down 6.7-6.9 + QKV 6.3-6.5 + out-projection 5.0 + attention 4.8-4.9 = 30.7 us per
#   layer; FETCH_SIZE x2 within 1.3 % of the algorithmic bytes on the three weight streams; in-kernel stamps of the one-row kernels (twin
#   library: python voicecraft_amd/build.py --ts BEFORE the call - the .so travels with the snapshot); the leftover A/Bs.
#   Wide steps: python bench.py --batch 32|64 --no-codec --no-configs --no-cpu-baseline -> 95-96 k / 128-129 k tok/s (1.29-1.30 / 1.91-1.93 ms).
# Every default-on launch-shape form carries its in-process A/B in the bench line (`ab`: fr_one; `ab_more`: qkv_p8, ln_trim, gemm_pf and
# the default-off attn_pf; many-row options: --ab qkv16=0:1 / wide_heads=0:1 with --batch 32).  Several A/Bs on one engine: python tools/ab_sweep.py [--preset P] [--batch B] knob=A:B ...
set -u
export TMPDIR=/tmp
bash tools/round_check.sh
