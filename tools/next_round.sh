#!/bin/bash
# First GPU call of the next round (from the repo root on the GPU box; ~3 minutes):
#   gpurun --timeout 400 -- 'bash tools/next_round.sh > gpurun_out/next_round.log 2>&1; tail -40 gpurun_out/next_round.log'
# Validates what round 2 wrote after its GPU budget was spent, each step under its own timeout.
set -u
export TMPDIR=/tmp
echo "== f4: vc_eval_forward against the reference fixtures (opt-in tests)"
VC_TEST_EXPERIMENTAL=1 timeout 150 python -m pytest tests/test_gpu_forward.py -x -q 2>&1 | tail -5
echo "== persistent LSTM, both forms, against the wavefront path"
timeout 90 python tools/lstm_probe.py 1 2>&1 | tail -10
VC_TEST_EXPERIMENTAL=1 timeout 60 python -m pytest tests/test_gpu_codec.py -x -q -k experimental 2>&1 | tail -3
echo "== attention split count of a single-row decode step (VC_ATTN_BLOCKS1 = blocks per row)"
for b in 16 32 64 128; do   # = 1, 2, 4, 8 splits at 16 heads (VC_MAX_NSPLIT caps at 8)
  VC_ATTN_BLOCKS1=$b timeout 60 python tools/variant_sweep.py 1 2>&1 | tail -1 | sed "s/^/blocks1=$b  /"
done
