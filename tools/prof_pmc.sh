#!/bin/bash
# usage (GPU box, repo root): bash tools/prof_pmc.sh <tag> "<counters>" [bench args...]
# one rocprofv3 --pmc pass (own run: counters are never combined with trace domains beyond --kernel-trace)
tag=$1; ctr=$2; shift; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out
mkdir -p $out /tmp/pmc_$tag
( cd /tmp && rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_$tag -o pm -- python $OLDPWD/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-codec "$@" > $out/${tag}_bench_under_pmc.log 2>&1 )
db=$(find /tmp/pmc_$tag -name "*.db" | head -1)
python tools/rocprof_pmc_summary.py $db $out/${tag}_pmc.txt
