set -u
export TMPDIR=/tmp
O=gpurun_out
echo "== att_p16 A/B"
for B in 8 4 2 6; do timeout 300 python tools/ab_sweep.py --batch $B att_p16=0:1 2>&1 | grep -v amdgpu.ids | tee -a $O/r06f_ab_att_p16.log; done
echo "== tests"; timeout 1200 python -m pytest tests/test_gpu_options.py tests/test_gpu_scale.py -m gpu -q -x -k "finished_row or c5_share or eight_utter or options_do_not" 2>&1 | tail -8 | tee $O/r06f_pytest.log
echo "== 8 rows traced"; bash tools/prof_decode.sh r06f_b8 --batch 8 --no-codec --ab none --no-configs; head -10 $O/r06f_b8_rocprof_kernel_stats.txt
