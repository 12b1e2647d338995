for b in ${GRID_BATCHES:-96 128 160 192 256}; do for s in ${GRID_STREAMS:-4 6}; do
  v=""; for i in 1 2 3; do v="$v $(timeout -k 10 100 python bench.py --batch $b --streams $s --steps 20 --warmup 5 --no-cpu-baseline --no-batched-matching --no-next-rows --no-extra-configs 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(int(d['value']))")"; done
  echo "batch $b in flight $s: value (K=20) $v"
done; done
