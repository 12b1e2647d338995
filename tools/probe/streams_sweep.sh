# GPU-box probe: steps in flight on the default bench line (20 timed steps as the driver runs it, then the sustained figure)
for rep in 1 2; do
for n in 3 4 5 6; do
  timeout -k 10 200 python bench.py --streams $n --steps 20 --no-cpu-baseline --no-batched-matching --no-next-rows > gpurun_out/tmp_b.json 2>/dev/null || exit 1
  python3 - $n <<'PY'
import json,sys
d=json.loads(open('gpurun_out/tmp_b.json').read().strip().splitlines()[-1])
print('steps in flight', sys.argv[1], 'value', d['value'], 'sustained', d['sustained']['value'], 'mfma', d['mfma_matcher']['value'], d['mfma_matcher']['sustained'], flush=True)
PY
done
done
