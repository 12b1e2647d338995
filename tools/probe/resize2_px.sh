# GPU box: the pixel threshold below which two pyramid levels share a launch (k_resize2), on the default line
for rep in 1 2; do
for t in 600000 9000000 13000000 19000000 27000000; do
  SSVIO_RESIZE2_MAX_PX=$t timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra-configs --no-batched-matching --no-next-rows > gpurun_out/tmp_b.json 2>/dev/null || exit 1
  python3 - $t <<'PY'
import json,sys
d=json.loads(open('gpurun_out/tmp_b.json').read().strip().splitlines()[-1])
print('max px', sys.argv[1], 'value', d['value'], 'resize avg', d['kernels']['k_resize'], flush=True)
PY
done
done
