# A/B on the GPU box at the default batch: bench with the tree's library, then with tools/probe/lib_old.bin in its place (3 runs each, interleaved)
cp ssvio_amd/libssvio_gpu.so /tmp/lib_new.so
for i in 1 2 3; do
  for v in new old; do
    if [ $v = new ]; then cp /tmp/lib_new.so ssvio_amd/libssvio_gpu.so; else cp tools/probe/lib_old.bin ssvio_amd/libssvio_gpu.so; fi
    timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batched-matching --no-next-rows > gpurun_out/tmp_b.json 2>/dev/null || exit 1
    python3 - $v <<'PY'
import json,sys
d=json.loads(open('gpurun_out/tmp_b.json').read().strip().splitlines()[-1])
k=d["kernels"]
print(sys.argv[1], 'value', d['value'], 'sustained', d['sustained']['value'], 'mfma', d['mfma_matcher']['sustained'], 'fast', k["k_fast"]["avg_us"], "blur", k["k_blur"]["avg_us"], "describe", k["k_describe"]["avg_us"], "resize", k["k_resize"]["avg_us"], "knn2", k["k_hamming_knn2"]["avg_us"], "qt", k["k_quadtree"]["avg_us"], 'level0', k['k_copy_gray']['avg_us'], 'cfg0', d["configs"][0]["value"], '1080p', d["configs"][2]["value"], flush=True)
PY
  done
done
cp /tmp/lib_new.so ssvio_amd/libssvio_gpu.so
