#!/bin/bash
# First GPU call of a round (from the repo root on the GPU box; ~6 minutes):
#   gpurun --timeout 900 -- 'bash tools/next_round.sh > gpurun_out/next_round.log 2>&1; tail -40 gpurun_out/next_round.log'
# Re-establishes the state the previous round ended in: full GPU suite, the default bench line, and the two opt-in paths.
set -u
export TMPDIR=/tmp
echo "== GPU suite"
timeout 700 python -m pytest tests -m gpu -q 2>&1 | tail -4
echo "== default bench line"
timeout 200 python bench.py --steps 5 --warmup 2 2>/dev/null | tail -c 2500
echo "== stream engine (VC_STREAM=1): parity + phase stamps at giga830M"
timeout 200 python tools/stream_probe.py giga830M 16 80 150 2>&1 | grep -v "layer [0-9]*: rel" | tail -6
echo "== prefill GEMM by pass size"
timeout 200 python tools/pf_gemm_probe.py 2>&1 | grep pf_gemm
