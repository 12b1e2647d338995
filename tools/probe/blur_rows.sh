# GPU box: rows per k_blur strip on the default line
for rep in 1 2; do
for r in 29 43 57 85; do
  SSVIO_BLUR_ROWS=$r timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --no-batched-matching --no-next-rows > gpurun_out/tmp_b.json 2>/dev/null || exit 1
  python3 - $r <<'PY'
import json,sys
d=json.loads(open('gpurun_out/tmp_b.json').read().strip().splitlines()[-1])
print('blur rows', sys.argv[1], 'value', d['value'], 'k_blur', d['kernels']['k_blur']['avg_us'], flush=True)
PY
done
done
