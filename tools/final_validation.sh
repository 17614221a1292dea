#!/bin/bash
# Round-end validation on the GPU box (repo root): the full GPU suite, smoke(), the driver's bench command, rocprof kernel
# stats and a FETCH_SIZE pass of the same command, the other bench configurations.  Everything lands in gpurun_out/<tag>_*.
tag=${1:-r03h}
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > $out/${tag}_pytest_gpu.log; tail -2 $out/${tag}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $out/${tag}_bench.json.log
python -c "import json; d=json.loads(open('$out/${tag}_bench.json.log').read()); print('bench', d['value'], d['decode_ms_per_token_step'], d['prefill_ms'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['prefill_roofline']['at_2048_rows']['frac'], d['prefill_roofline']['attention']['avg_launch_us'], d['codec']['encode_ms'], d['codec']['decode_ms'])"
bash tools/prof_decode.sh $tag --no-codec >/dev/null 2>&1; head -12 $out/${tag}_rocprof_kernel_stats.txt
bash tools/prof_pmc.sh ${tag}_fetch FETCH_SIZE >/dev/null 2>&1; python tools/pmc_to_json.py $out/${tag}_fetch_pmc.txt $out/${tag}_pmc_traffic.json | head -30
for cfg in "--batch 8" "--mode edit" "--batch 32"; do
  timeout 300 python bench.py $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-codec 2>/dev/null | tail -1 > $out/${tag}_bench_$(echo $cfg | tr -d ' -').json.log
  python -c "import json; d=json.loads(open('$out/${tag}_bench_$(echo $cfg | tr -d ' -').json.log').read()); print('$cfg', d['value'], 'step', d['decode_ms_per_token_step'], 'prefill', d['prefill_ms'])"
done
