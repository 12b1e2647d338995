"""GPU-box probe: k_hamming_knn2 (the default kernel) by queries per lane (SSVIO_MATCH_R) and workgroup target (SSVIO_MATCH_BLOCKS) on the shapes of the
bench line, configs[2], configs[3] and configs[4].  One process per setting (the knobs are read once): python tools/probe/match_r_sweep.py [child rows pairs mode]"""
import os, subprocess, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
if len(sys.argv) > 1:
    import numpy as np, torch
    import ssvio_amd
    tag, rows, B, mode = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    rng = np.random.default_rng(1)
    cap = rows + 32
    counts = rng.integers(rows - 10, rows + 6, B).astype(np.int32)
    d = torch.from_numpy(rng.integers(0, 256, (B, cap, 32), dtype=np.uint8)).cuda()
    c = torch.from_numpy(counts).cuda()
    pairs = np.array([[f, (f - 1) % B] for f in range(B)], np.int32)
    fm = ssvio_amd.FeatureMatching(None, {})
    for _ in range(3): fm.match_batch_dev(d, cap, cap, c, pairs, mode)
    fm.sync(); fm.set_profiling(True)
    for _ in range(12): fm.match_batch_dev(d, cap, cap, c, pairs, mode)
    fm.sync()
    t = fm.kernel_times()
    pe = float((counts[pairs[:, 0]].astype(np.float64) * counts[pairs[:, 1]]).sum()) * (2 if mode in (2, 4) else 1)
    us = {k: v[0] * 1e3 / v[1] * (2 if (k == "k_hamming_knn2" and mode in (2, 4)) else 1) for k, v in t.items()}     # cross-check: two launches per call
    print("%-34s %5d rows x %4d pairs mode %d  knn2 %8.2f us per call -> %.3e pairs/s" % (tag, rows, B, mode, us["k_hamming_knn2"], pe / (us["k_hamming_knn2"] * 1e-6)), flush=True)
else:
    for rows, B, mode in ((1006, 256, 0), (1006, 768, 0), (1006, 64, 0), (2000, 64, 0), (4000, 32, 1), (4000, 64, 2), (8000, 32, 0)):
        for env in ({}, {"SSVIO_MATCH_R": "1"}, {"SSVIO_MATCH_R": "2"}, {"SSVIO_MATCH_R": "4"}, {"SSVIO_MATCH_R": "1", "SSVIO_MATCH_BLOCKS": "4096"}, {"SSVIO_MATCH_R": "2", "SSVIO_MATCH_BLOCKS": "4096"}):
            e = dict(os.environ); e.update(env)
            r = subprocess.run([sys.executable, __file__, " ".join("%s=%s" % (k[12:], v) for k, v in env.items()) or "default", str(rows), str(B), str(mode)], env=e, capture_output=True, text=True, timeout=120)
            print(r.stdout.strip() or r.stderr[-400:], flush=True)
