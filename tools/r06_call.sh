set -u
export TMPDIR=/tmp
O=gpurun_out
for B in 64 32 16 8; do
  bash tools/prof_decode.sh r06p_b$B --batch $B --no-codec --ab none --no-configs 2>/dev/null; head -12 $O/r06p_b${B}_rocprof_kernel_stats.txt
done
