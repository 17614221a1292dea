#!/bin/bash
# round-end check of the final library: GPU suite, smoke, A/B of the LayerNorm-launch prefetch at 32 rows, the driver's bench command
tag=${1:-r03j}
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > $out/${tag}_pytest_gpu.log; tail -2 $out/${tag}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
run() { VC_LN_PF=$1 timeout 200 python bench.py --batch 32 --steps 2 --warmup 1 --no-cpu-baseline --no-codec 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[lpf32] VC_LN_PF=$1', d['value'], 'step', d['decode_ms_per_token_step'])"; }
(run 0; run 248,24,24; run 0; run 248,24,24) 2>&1 | tee $out/${tag}_lpf32_ab.log
timeout 600 python bench.py --steps 10 --warmup 2 2>/dev/null | tail -1 > $out/${tag}_bench.json.log
python -c "import json; d=json.loads(open('$out/${tag}_bench.json.log').read()); print('bench', d['value'], d['decode_ms_per_token_step'], d['prefill_ms'])"
