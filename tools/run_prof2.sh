bash tools/prof_decode.sh r03d_edit --mode edit --no-codec >/dev/null 2>&1; head -30 gpurun_out/r03d_edit_rocprof_kernel_stats.txt
bash tools/prof_decode.sh r03d_b8 --batch 8 --no-codec >/dev/null 2>&1; head -30 gpurun_out/r03d_b8_rocprof_kernel_stats.txt
