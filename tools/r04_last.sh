#!/bin/bash
# Round 4, last GPU call: the whole suite, smoke and the driver's bench command on the final library.
set -u
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
date
timeout 1100 python -m pytest tests -m gpu -q 2>&1 | tail -5 | tee $O/r04l_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/r04l_smoke.log
timeout 500 python bench.py 2>$O/r04l_bench.err | tail -1 > $O/r04l_bench.json.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04l_bench.json.log"))
print({k:d[k] for k in ("value","ms_per_step","decode_ms_per_token_step","prefill_ms","rtf")}, d["roofline"]["frac"], d["decode_step"]["hbm_frac_in_loop"], d["ab"]["median_delta_pct"], d["ab"]["spread_pct"], d["cpu_baseline"]["value"], d["config"]["engine_options"])
PY
date
