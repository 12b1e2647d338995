# GPU box: shader clock and power while the bench line's steps run (polls rocm-smi beside a 3000-step run)
R=$GRAFT_REPO_ROOT; cd $R
rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|power" | head -4
( timeout -k 5 120 python bench.py --steps 3000 --warmup 5 --no-cpu-baseline --no-extra-configs --no-batched-matching --no-next-rows > gpurun_out/clock_bench.json 2>/dev/null ) &
BP=$!
sleep 14
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|Socket Power\|Average" | tr '\n' ' '; echo
  sleep 0.4
done
wait $BP
python3 -c "
import json; d=json.loads(open('gpurun_out/clock_bench.json').read().strip().splitlines()[-1]); print('value', d['value'], 'ms_per_step', d['ms_per_step'])"
