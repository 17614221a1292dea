export TMPDIR=/tmp
for f in 2 3; do VC_TILE_ATTN=$f timeout 200 python tools/pf_attn_probe.py 2>&1 | grep pf_attn; done
timeout 900 python -m pytest tests/test_gpu_scale.py tests/test_gpu_model.py -x -q -m gpu 2>&1 | tail -5
