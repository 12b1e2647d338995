"""GPU-box probe: k_hamming_knn2 (VALU) against k_hamming_knn2_mfma on the batched shapes -- 8000 x 8000 blocks (BASELINE
config 5), 64 x 4000 x 4000 (config 4's shape), 256 pairs of ~1006 rows (the bench line's step).  One process per knob
setting (the knobs are read once): python tools/probe/mfma_probe.py [child-tag]"""
import os, subprocess, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
if len(sys.argv) > 1:
    import numpy as np, torch
    import ssvio_amd
    rng = np.random.default_rng(1)
    shapes = [("8000x8000 x32", 9, 8000, 8000, 32, 0), ("4000x4000 x64", 65, 4000, 4000, 64, 0), ("1006x1006 x256", 256, 1032, 1006, 256, 1)]
    for name, nsets, cap, rows, npairs, ragged in shapes:
        counts = (rng.integers(rows - 10, rows + 6, nsets) if ragged else np.full(nsets, rows)).astype(np.int32)
        d = torch.from_numpy(rng.integers(0, 256, (nsets, cap, 32), dtype=np.uint8)).cuda()
        c = torch.from_numpy(counts).cuda()
        pairs = np.array([[(f + 1) % nsets, f % nsets] for f in range(npairs)], np.int32)
        pe = float((counts[pairs[:, 0]].astype(np.float64) * counts[pairs[:, 1]]).sum())
        for kernel, mode in (("valu", 0), ("mfma", 0), ("mfma", 1)):
            fm = ssvio_amd.FeatureMatching(None, {})
            fm.set_kernel(kernel)
            for _ in range(3): fm.match_batch_dev(d, cap, cap, c, pairs, mode)
            fm.sync(); fm.set_profiling(True)
            n = 20
            for _ in range(n): fm.match_batch_dev(d, cap, cap, c, pairs, mode)
            fm.sync()
            t = fm.kernel_times()
            us = {k: v[0] * 1e3 / v[1] for k, v in t.items()}
            print("%-22s %-16s %-5s mode %d knn2 %9.2f us  finalize(+merge) %7.2f us  -> %.3e pairs/s" % (sys.argv[1], name, kernel, mode, us["k_hamming_knn2"], us["k_match_finalize"],
                  pe / (us["k_hamming_knn2"] * 1e-6)), flush=True)
            fm.close()
else:
    for env in ({}, {"SSVIO_MFMA_G": "3"}, {"SSVIO_MFMA_BLOCKS": "2048"}):
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, __file__, " ".join("%s=%s" % (k.replace("SSVIO_MFMA_", ""), v) for k, v in env.items()) or "default"], env=e, capture_output=True, text=True, timeout=200)
        print(r.stdout.strip() or r.stderr[-600:], flush=True)
