# GPU-box probe: frames per step of the default bench line
for b in 64 96 128 192 256; do
  timeout -k 10 200 python bench.py --batch $b --steps $((2560 / b)) --warmup 5 --no-cpu-baseline --no-batched-matching --no-next-rows > gpurun_out/tmp_b.json 2>/dev/null || exit 1
  python3 - $b <<'PY'
import json,sys
d=json.loads(open('gpurun_out/tmp_b.json').read().strip().splitlines()[-1])
print('batch', sys.argv[1], 'value', d['value'], 'sustained', d['sustained']['value'] if d['sustained'] else None, 'valu', d['valu']['frac'], 'k_fast excl', d['kernels']['k_fast']['avg_us'], 'frac', d['roofline']['frac'], d['roofline']['exclusive_frac'])
PY
done
