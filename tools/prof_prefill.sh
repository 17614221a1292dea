# rocprofv3 kernel stats of the editing run (C4) and of the 8-utterance run (C5); summaries into gpurun_out/
export TMPDIR=/tmp
R=$PWD
cd /tmp
for cfg in "edit:--mode edit" "b8:--batch 8"; do
  tag=${cfg%%:*}; fl=${cfg#*:}
  rm -rf /tmp/prof_$tag
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o p -- python $R/bench.py $fl --steps 1 --warmup 1 --no-cpu-baseline --no-codec > /tmp/prof_$tag.log 2>&1
  tail -5 /tmp/prof_$tag.log | cut -c1-400
  grep '^{' /tmp/prof_$tag.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag', d['value'], 'step', d['decode_ms_per_token_step'], 'prefill', d['prefill_ms'])"
  f=$(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1); ls -R /tmp/prof_$tag | head
  python - "$f" "$tag" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
print(f"--- {sys.argv[2]}: kernels by total time")
for r in rows[:22]:
    print(f"{r['Name'][:90]:90s} calls {int(r['Calls']):6d} avg {float(r['AverageNs'])/1e3:8.2f} us total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
  cp "$f" $R/gpurun_out/r03d_${tag}_kernel_stats.csv
done
