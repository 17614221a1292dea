set -u
export TMPDIR=/tmp
O=gpurun_out
timeout 400 python tools/ragged_probe.py 8 2>&1 | grep -v amdgpu.ids | tee $O/r06l_ragged8.log
timeout 400 python tools/ragged_probe.py 8 2>&1 | grep -v amdgpu.ids | tee -a $O/r06l_ragged8.log
