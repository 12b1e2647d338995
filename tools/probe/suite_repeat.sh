# GPU-box probe: the GPU suite n times in a row (fresh process each), to catch flaky failures and hangs
n=${1:-5}
for i in $(seq 1 $n); do
  timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread -p no:cacheprovider > gpurun_out/suite_$i.log 2>&1
  echo "run $i rc=$? $(tail -1 gpurun_out/suite_$i.log)"
done
