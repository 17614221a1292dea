#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/prof_decode.sh <tag> [bench args...]
# rocprofv3 kernel trace of a short bench run -> per-kernel summary + a timeline window + in-kernel stamps
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out
mkdir -p $out /tmp/prof_$tag
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o kt -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $out/${tag}_bench_under_rocprof.log 2>&1 )
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
python tools/rocprof_summary.py $db $out/${tag}_rocprof_kernel_stats.txt
python tools/rocprof_timeline.py $db 30000 100 > $out/${tag}_rocprof_timeline.txt
