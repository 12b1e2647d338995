# GPU-box probe: frames per step x steps in flight on the default bench line (sustained rate, 40 steps)
for b in 192 256 384 512; do
  for n in 3 4 6 8; do
    timeout -k 10 200 python bench.py --batch $b --streams $n --steps 40 --warmup 6 --no-cpu-baseline --no-extra-configs --no-batched-matching --no-next-rows > gpurun_out/tmp_b.json 2>/dev/null || exit 1
    python3 - $b $n <<'PY'
import json,sys
d=json.loads(open('gpurun_out/tmp_b.json').read().strip().splitlines()[-1])
print('batch', sys.argv[1], 'in flight', sys.argv[2], 'value', d['value'], 'sustained', d['sustained']['value'] if d.get('sustained') else None, flush=True)
PY
  done
done
