# A/B on the GPU box: bench with the tree's library, then with tools/probe/lib_old.bin in its place (3 runs each, interleaved)
cp ssvio_amd/libssvio_gpu.so /tmp/lib_new.so
for i in 1 2 3; do
  for v in new old; do
    if [ $v = new ]; then cp /tmp/lib_new.so ssvio_amd/libssvio_gpu.so; else cp tools/probe/lib_old.bin ssvio_amd/libssvio_gpu.so; fi
    timeout -k 10 200 python bench.py --batch 256 --steps 40 --warmup 5 --no-cpu-baseline --no-batched-matching --no-next-rows > gpurun_out/tmp_b.json 2>/dev/null || exit 1
    python3 - $v <<'PY'
import json,sys
d=json.loads(open('gpurun_out/tmp_b.json').read().strip().splitlines()[-1])
print(sys.argv[1], 'value', d['value'], 'sustained', d['sustained']['value'], 'k_fast excl', d["kernels"]["k_fast"]["avg_us"], "blur", d["kernels"]["k_blur"]["avg_us"], "describe", d["kernels"]["k_describe"]["avg_us"], "resize", d["kernels"]["k_resize"]["avg_us"], "knn2", d["kernels"]["k_hamming_knn2"]["avg_us"], "qt", d["kernels"]["k_quadtree"]["avg_us"], "cfg0", d["configs"][0]["value"], 'level0', d['kernels']['k_copy_gray']['avg_us'], 'in-region', d['roofline']['avg_launch_us'])
PY
  done
done
cp /tmp/lib_new.so ssvio_amd/libssvio_gpu.so
