"""CPU oracle for the VoiceCraft decode path — TEST INFRASTRUCTURE ONLY.

This package restates, on the CPU, the algorithm of the reference's hot path
(jasonppy/VoiceCraft: models/voicecraft.py, models/codebooks_patterns.py, models/modules/*)
so that the HIP engine in `voicecraft_amd/` can be checked on a GPU box where `/root/reference`
does not exist.  Integer work (delayed pattern, un-shift, sequence rearrangement) is numpy / pure
Python; the floating-point Transformer is torch-CPU fp32, issuing the same ATen ops in the same
order and shapes as the reference so that its greedy trajectories are bit-identical to it.

Pinning: `tests/golden/*.npz` were produced by `oracle/gen_golden.py`, which imports the real
reference from `/root/reference` in the build container; `tests/test_oracle_golden.py` checks this
oracle against them.  The EnCodec restatement (oracle/encodec_oracle.py) has no reference output
to pin against (audiocraft is not vendored): parity unpinned at that boundary.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import this
package.  Nothing under `voicecraft_amd/` does.
"""
