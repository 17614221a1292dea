#!/bin/bash
# round-end check of the final library: GPU suite, smoke, the driver's bench command, A/B of the LayerNorm-launch prefetch at 8 rows
tag=${1:-r03i}
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > $out/${tag}_pytest_gpu.log; tail -2 $out/${tag}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $out/${tag}_bench.json.log
python -c "import json; d=json.loads(open('$out/${tag}_bench.json.log').read()); print('bench', d['value'], d['decode_ms_per_token_step'], d['prefill_ms'], d['roofline']['frac'], d['roofline']['traffic'])"
bash tools/lpf_sweep.sh 0 248,24,24 0 248,24,24 2>&1 | tee $out/${tag}_lpf_ab.log
timeout 300 python bench.py --mode edit --steps 3 --warmup 1 --no-cpu-baseline --no-codec 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('edit', d['value'], 'step', d['decode_ms_per_token_step'], 'prefill', d['prefill_ms'])"
