# GPU-box probe: frames per step of the default bench line over the 20 steps the driver times (fill and drain weigh less the longer a step is)
for rep in 1 2; do
for b in 256 384 512 768 1024; do
  timeout -k 10 200 python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-batched-matching --no-next-rows > gpurun_out/tmp_b.json 2>/dev/null || exit 1
  python3 - $b <<'PY'
import json,sys
d=json.loads(open('gpurun_out/tmp_b.json').read().strip().splitlines()[-1])
print('batch', sys.argv[1], 'value', d['value'], 'sustained', d['sustained']['value'], 'mfma', d['mfma_matcher']['value'], d['mfma_matcher']['sustained'], 'k_fast frac', d['roofline']['frac'], flush=True)
PY
done
done
