# two-slot host_live (deterministic re-pack step): decode-path parity files + the ragged probe
set -u
export TMPDIR=/tmp
O=gpurun_out
echo "== tests"; timeout 1500 python -m pytest tests/test_gpu_options.py tests/test_gpu_model.py tests/test_gpu_scale.py -m gpu -q 2>&1 | tail -15 | tee $O/r06r_pytest.log
echo "== ragged probe"; timeout 300 python tools/ragged_probe.py 8 2>&1 | grep -v amdgpu.ids | tee $O/r06r_ragged8.log
