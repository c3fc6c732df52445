# acoustic-batch sweep of the bench (development aid)
for ab in "$@"; do
  timeout 250 python bench.py --acoustic-batch $ab --acoustic-min-batch $ab --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/ab_$ab.json
  python - "$ab" <<'PY'
import json, sys
d = json.loads(open('gpurun_out/ab_%s.json' % sys.argv[1]).read())
print('acoustic_batch', sys.argv[1], d['value'], d['ms_per_step'], d['stage_seconds_per_step']['flow+hift'])
PY
done
