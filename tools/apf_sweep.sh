# sweep of the piggyback weight prefetch (VC_ATTN_PF=z,wo_kb,w1_kb) on the batch-1 bench
export TMPDIR=/tmp
run() { VC_ATTN_PF=$1 timeout 150 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-codec 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[apf] VC_ATTN_PF=$1', d['value'], 'step', d['decode_ms_per_token_step'])"; }
for v in "$@"; do run $v; done
