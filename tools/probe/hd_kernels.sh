# GPU-box probe: the 1080p / 4000-feature / kNN2-ratio configuration as the bench's main workload, to read its exclusive kernel times
timeout -k 10 280 python bench.py --width 1920 --height 1080 --features 4000 --match-mode ratio --batch 32 --steps 40 --warmup 5 --no-cpu-baseline --no-batched-matching --no-next-rows --no-extra-configs > gpurun_out/hd_b.json 2>gpurun_out/hd_b.err || exit 1
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/hd_b.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], 'kp', d.get('keypoints_per_frame'))
t=0
for k,v in d['kernels'].items():
    print(k.ljust(20), v['launches'], v['avg_us'], round(v['total_ms']/d['steps']*1000,1), 'us/step', v.get('GBps'))
    t+=v['total_ms']/d['steps']
print('sum ms/step', t)
print(d.get('valu'))
PY
