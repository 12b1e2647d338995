#!/bin/bash
# usage (on the GPU box): tools/gpu_check.sh <tag> [pytest-args...]   -- parity tests, then a short bench
set -o pipefail
tag=${1:-run}; shift
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread "$@" > gpurun_out/${tag}_tests.log 2>&1
rc=$?
tail -4 gpurun_out/${tag}_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python -u bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { tail -5 gpurun_out/${tag}_bench.err; exit 1; }
python - <<PY
import json
d=json.loads(open("gpurun_out/${tag}_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "roof", d["roofline"]["kernel"], d["roofline"]["frac"], "valu", d["valu"]["frac"] if d["valu"] else None)
for k,v in d["kernels"].items(): print("  %-18s avg_us %8.2f  launches %4d  GB/s %s"%(k,v["avg_us"],v["launches"],v["GBps"]))
print("sustained", d.get("sustained"))
for c in d.get("configs") or []: print("  cfg", c["value"], c["unit"], c["config"][:60])
for c in d.get("batched_matching") or []: print("  bm", c["ms"], "ms", "%.3e pairs/s" % c["descriptor_pairs_per_s"], c["config"][:50])
PY
for extra in "--sync-steps" "--sync-steps --batch 1 --steps 200"; do
  timeout -k 10 200 python -u bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-batched-matching --no-extra-configs --no-next-rows $extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$extra', 'value', d['value'], 'ms/step', d['ms_per_step'])" || exit 1
done
