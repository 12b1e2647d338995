"""GPU-box probe for profilers: N launches of one matcher kernel on one batched shape, nothing else.
python3 tools/probe/mfma_one.py <valu|mfma> <rows> <npairs> [launches]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import ssvio_amd
kernel, rows, npairs = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 10
rng = np.random.default_rng(1)
nsets = min(npairs + 1, 65)
d = torch.from_numpy(rng.integers(0, 256, (nsets, rows, 32), dtype=np.uint8)).cuda()
c = torch.full((nsets,), rows, dtype=torch.int32).cuda()
pairs = np.array([[(f + 1) % nsets, f % nsets] for f in range(npairs)], np.int32)
fm = ssvio_amd.FeatureMatching(None, {})
fm.set_kernel(kernel)
for _ in range(n): fm.match_batch_dev(d, rows, rows, c, pairs, 0)
fm.sync()
fm.close()
print("done", kernel, rows, npairs, n)
