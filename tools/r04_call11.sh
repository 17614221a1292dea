#!/bin/bash
# Round 4, eleventh GPU call: the slab form (finished rows off) through every several-row parity test, by name.
set -u
export TMPDIR=/tmp
VC_FINISHED_ROWS=0 timeout 700 python -m pytest tests/test_gpu_model.py tests/test_gpu_scale.py tests/test_gpu_forward.py -m gpu -q -k "batch or wide or batched or c5_share or eight_utterances or multi_utterance or best_of or edit_3span or replay or thirty_two or forward" 2>&1 | tail -12
