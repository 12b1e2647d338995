#!/bin/bash
# usage (GPU box): tools/gpu_sweep.sh "<bench args>" ...   -- one bench line per argument set
for extra in "$@"; do
  timeout -k 10 200 python -u bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-batched-matching $extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$extra', '| value', d['value'], 'ms/step', d['ms_per_step'], 'dom', d['roofline']['kernel'], d['roofline']['avg_launch_us'], {k: v['avg_us'] for k, v in d['kernels'].items()})" || exit 1
done
