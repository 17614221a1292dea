set -u
export TMPDIR=/tmp
O=gpurun_out
echo "== attn_pipe A/B"
for B in 8 16 4 12 2; do timeout 300 python tools/ab_sweep.py --batch $B attn_pipe=0:1 2>&1 | grep -v amdgpu.ids | tee -a $O/r06e_ab_attn_pipe.log; done
echo "== tests"; timeout 1200 python -m pytest tests/test_gpu_options.py tests/test_gpu_scale.py tests/test_gpu_model.py -m gpu -q -x 2>&1 | tail -8 | tee $O/r06e_pytest.log
echo "== 8 rows traced"; bash tools/prof_decode.sh r06e_b8 --batch 8 --no-codec --ab none --no-configs; head -10 $O/r06e_b8_rocprof_kernel_stats.txt
echo "== 16 rows traced"; bash tools/prof_decode.sh r06e_b16 --batch 16 --no-codec --ab none --no-configs; head -10 $O/r06e_b16_rocprof_kernel_stats.txt
