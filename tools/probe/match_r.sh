# GPU box: queries per lane of k_hamming_knn2 on the bench line
for rep in 1 2; do
for r in 0 1 2 4; do
  SSVIO_MATCH_R=$r timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --no-batched-matching --no-next-rows > gpurun_out/tmp_b.json 2>/dev/null || exit 1
  python3 - $r <<'PY'
import json,sys
d=json.loads(open('gpurun_out/tmp_b.json').read().strip().splitlines()[-1])
print('R', sys.argv[1], 'value', d['value'], 'knn2', d['kernels']['k_hamming_knn2']['avg_us'], flush=True)
PY
done
done
