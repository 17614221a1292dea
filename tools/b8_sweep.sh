export TMPDIR=/tmp
run() { env "$@" timeout 150 python bench.py --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-codec 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['value'], 'step', d['decode_ms_per_token_step'], 'prefill', d['prefill_ms'])"; }
run A=1
run VC_ATTN_BLOCKS=1024
run VC_ATTN_BLOCKS=256
run VC_LN_SPLIT_ROWS=9
run VC_KSPLIT_F=2
run A=2
