set -u
export TMPDIR=/tmp
O=gpurun_out
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_one_row.py -m gpu -q -x 2>&1 | tail -12 | tee $O/r06m_pytest.log
echo "== fr_one 1 -> 3"
timeout 300 python tools/ab_sweep.py fr_one=1:3 2>&1 | grep -v amdgpu.ids | tee -a $O/r06m_ab_fr3.log
timeout 300 python tools/ab_sweep.py --preset giga330M fr_one=1:3 2>&1 | grep -v amdgpu.ids | tee -a $O/r06m_ab_fr3.log
