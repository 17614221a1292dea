set -u
export TMPDIR=/tmp
O=gpurun_out
echo "== tests"; timeout 1800 python -m pytest tests/test_gpu_options.py tests/test_gpu_scale.py tests/test_gpu_model.py tests/test_gpu_one_row.py -m gpu -q 2>&1 | tail -25 | tee $O/r06h_pytest.log
echo "== att_p16 one-row A/B"
timeout 300 python tools/ab_sweep.py att_p16=1:2 2>&1 | grep -v amdgpu.ids | tee -a $O/r06h_ab_p16.log
timeout 300 python tools/ab_sweep.py --preset giga330M att_p16=1:2 2>&1 | grep -v amdgpu.ids | tee -a $O/r06h_ab_p16.log
timeout 300 python tools/ab_sweep.py --batch 3 att_p16=1:2 hq=0:1 2>&1 | grep -v amdgpu.ids | tee -a $O/r06h_ab_p16.log
