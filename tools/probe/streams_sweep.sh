# GPU-box probe: steps in flight on the default bench line
for n in 2 3 4 6; do
  timeout -k 10 200 python bench.py --streams $n --steps 30 --warmup 6 --no-cpu-baseline --no-batched-matching --no-next-rows > gpurun_out/tmp_b.json 2>/dev/null || exit 1
  python3 - $n <<'PY'
import json,sys
d=json.loads(open('gpurun_out/tmp_b.json').read().strip().splitlines()[-1])
print('steps in flight', sys.argv[1], 'value', d['value'], 'sustained', d['sustained']['value'])
PY
done
