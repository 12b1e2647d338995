# GPU-box probe: the streamed configs[1] run under batch size / steps in flight / spare upload buffers
for v in "128 3 1" "128 3 3" "128 4 4" "256 3 3" "256 4 2" "128 6 6" "64 6 6"; do set -- $v
  SSVIO_BENCH_STREAMED_BATCH=$1 SSVIO_BENCH_STREAMED_IN_FLIGHT=$2 SSVIO_BENCH_STREAMED_SPARE_BUFFERS=$3 timeout -k 10 200 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-batched-matching --no-next-rows > gpurun_out/tmp_c.json 2>/dev/null || exit 1
  python3 - $v <<'PY'
import json,sys
d=json.loads(open('gpurun_out/tmp_c.json').read().strip().splitlines()[-1])
c=d['configs'][1]; print('batch %s in flight %s spare %s:' % tuple(sys.argv[1:4]), c['value'], 'frames/s', c['h2d_GBps'], 'GB/s', flush=True)
PY
done
timeout -k 10 100 python tools/probe/h2d_bw.py
