# sweep of the LayerNorm-launch piggyback prefetch (VC_LN_PF=blocks,qkv_kb,w1_kb) on the 8-utterance bench
export TMPDIR=/tmp
run() { VC_LN_PF=$1 timeout 150 python bench.py --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-codec 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[lpf] VC_LN_PF=$1', d['value'], 'step', d['decode_ms_per_token_step'])"; }
for v in "$@"; do run $v; done
