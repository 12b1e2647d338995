# GPU-box probe: the pixel count below which two pyramid levels are built per launch (k_resize2), on the default bench line
for v in 600000 2500000 5000000 10000000 100000000; do
  SSVIO_RESIZE2_MAX_PX=$v timeout -k 10 120 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra-configs --no-batched-matching --no-next-rows > gpurun_out/tmp_r.json 2>/dev/null || exit 1
  python3 - $v <<'PY'
import json,sys
d=json.loads(open('gpurun_out/tmp_r.json').read().strip().splitlines()[-1])
k=d['kernels']
print(sys.argv[1], d['value'], 'resize total us/step', round(k['k_resize']['total_ms']*1e3/d['steps'],1), 'launches/step', k['k_resize']['launches']/d['steps'], 'finalize', k['k_match_finalize']['avg_us'], 'knn2', k['k_hamming_knn2']['avg_us'], 'valu', d['valu']['frac'], d['roofline']['kernel'])
PY
done
