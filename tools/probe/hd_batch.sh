# GPU box: frames per step of the 1080p config (configs[2]) and of the extraction-only figures
for b in 32 64 96; do
  SSVIO_BENCH_1080P_BATCH=$b timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batched-matching --no-next-rows > gpurun_out/tmp_b.json 2>/dev/null || exit 1
  python3 - $b <<'PY'
import json,sys
d=json.loads(open('gpurun_out/tmp_b.json').read().strip().splitlines()[-1]); c=d['configs']
print('1080p batch', sys.argv[1], 'configs[2]', c[2]['value'], 'mfma', c[2]['mfma_matcher']['value'], 'extract-only', c[3]['value'], c[3]['value_1080p'], 'roofline_1080p frac', c[2]['roofline_1080p']['frac'], flush=True)
PY
done
