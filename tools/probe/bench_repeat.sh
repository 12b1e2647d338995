# GPU-box probe: the default bench command n times (fresh process each): exit status, wall time, value
n=${1:-4}
for i in $(seq 1 $n); do
  SECONDS=0
  timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/rep_$i.json 2> gpurun_out/rep_$i.err
  echo "run $i rc=$? wall=${SECONDS}s $(python3 -c "import json;d=json.loads(open('gpurun_out/rep_$i.json').read().strip().splitlines()[-1]);print(d['value'], d['sustained']['value'])")"
done
