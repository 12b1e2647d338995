#!/bin/bash
# A/B on the GPU box: the bench with and without an environment setting, 3 runs each, interleaved.  usage: tools/probe/ab_env.sh NAME=VALUE [bench args]
kv=$1; shift
for i in 1 2 3; do
  for v in off on; do
    if [ $v = on ]; then export "$kv"; else unset "${kv%%=*}"; fi
    timeout -k 10 200 python bench.py --batch 256 --steps 40 --warmup 5 --no-cpu-baseline --no-batched-matching --no-next-rows "$@" > gpurun_out/tmp_b.json 2>/dev/null || exit 1
    python3 - $v "$kv" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/tmp_b.json').read().strip().splitlines()[-1])
c=d.get("configs") or []
print(sys.argv[2] if sys.argv[1]=="on" else "(unset)", 'value', d['value'], 'sustained', (d.get('sustained') or {}).get('value'), 'mfma', (d.get("mfma_matcher") or {}).get("value"), 'cfg0', c[0]["value"] if c else None, 'streamed', c[1]["value"] if c else None, '1080p', c[2]["value"] if c else None, 'extract-only', (c[3]["value"], c[3]["value_1080p"]) if c else None)
PY
  done
done
