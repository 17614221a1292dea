set -u
export TMPDIR=/tmp
O=gpurun_out
echo "== tests"; timeout 1800 python -m pytest tests/test_gpu_options.py tests/test_gpu_scale.py tests/test_gpu_model.py -m gpu -q -x 2>&1 | tail -25 | tee $O/r06i_pytest.log
echo "== paired QKV consumer A/B"
for B in 8 4 2 6; do timeout 300 python tools/ab_sweep.py --batch $B qkv_p8=1:2 2>&1 | grep -v amdgpu.ids | tee -a $O/r06i_ab_qp.log; done
timeout 300 python tools/ab_sweep.py --preset giga330M --batch 8 qkv_p8=1:2 2>&1 | grep -v amdgpu.ids | tee -a $O/r06i_ab_qp.log
echo "== 8 rows traced"; bash tools/prof_decode.sh r06i_b8 --batch 8 --no-codec --ab none --no-configs; head -10 $O/r06i_b8_rocprof_kernel_stats.txt
