# GPU-box probe: k_blur forked before k_fast (default) against after it (SSVIO_BLUR_AFTER_FAST=1), 3 interleaved runs each
for i in 1 2 3; do
  for v in early late; do
    if [ $v = late ]; then export SSVIO_BLUR_AFTER_FAST=1; else unset SSVIO_BLUR_AFTER_FAST; fi
    timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-batched-matching --no-next-rows > gpurun_out/tmp_b.json 2>/dev/null || exit 1
    python3 - $v <<'PY'
import json,sys
d=json.loads(open('gpurun_out/tmp_b.json').read().strip().splitlines()[-1])
c={c['config'][:10]: c.get('value') for c in d['configs']}
print('fork', sys.argv[1], 'value', d['value'], 'sustained', d['sustained']['value'], 'others', c)
PY
  done
done
