"""GPU-box probe: k_hamming_knn2 time for the per-frame batch shape (64 pairs of ~1000 x 1000 rows, set stride 1032) under the
matcher's tuning knobs.  One process per setting (the knobs are read once): python tools/probe/match_sweep.py [child]"""
import os, subprocess, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
if len(sys.argv) > 1:
    import numpy as np, torch
    import ssvio_amd
    from ssvio_amd import synth
    rng = np.random.default_rng(1)
    B, cap = int(os.environ.get("SWEEP_B", "64")), 1032
    counts = rng.integers(996, 1012, B).astype(np.int32)
    d = torch.from_numpy(rng.integers(0, 256, (B, cap, 32), dtype=np.uint8)).cuda()
    c = torch.from_numpy(counts).cuda()
    pairs = np.array([[f, (f - 1) % B] for f in range(B)], np.int32)
    fm = ssvio_amd.FeatureMatching(None, {})
    for _ in range(5): fm.match_batch_dev(d, cap, cap, c, pairs, 0)
    fm.sync(); fm.set_profiling(True)
    for _ in range(50): fm.match_batch_dev(d, cap, cap, c, pairs, 0)
    fm.sync()
    t = fm.kernel_times()
    pe = float((counts[pairs[:, 0]].astype(np.float64) * counts[pairs[:, 1]]).sum())
    us = {k: v[0] * 1e3 / v[1] for k, v in t.items()}
    print("%-40s knn2 %.2f us  finalize(+merge) %.2f us  -> %.3e pairs/s = %.3f of the VALU roof" % (sys.argv[1], us["k_hamming_knn2"], us["k_match_finalize"],
          pe / (us["k_hamming_knn2"] * 1e-6), pe / (us["k_hamming_knn2"] * 1e-6) / 2.42e12))
else:
    for env in ({}, {"SWEEP_B": "128"}, {"SWEEP_B": "256"}, {"SWEEP_B": "16"}, {"SSVIO_MATCH_BLOCKS": "1024"}, {"SSVIO_MATCH_BLOCKS": "4096"}, {"SSVIO_MATCH_R": "4"}, {"SSVIO_MATCH_R": "4", "SSVIO_MATCH_BLOCKS": "1024"},
                {"SSVIO_MATCH_R": "1"}, {"SSVIO_MATCH_R": "1", "SSVIO_MATCH_BLOCKS": "4096"}, {"SSVIO_MATCH_CHUNKROWS": "128"}, {"SSVIO_MATCH_CHUNKROWS": "256"}):
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, __file__, " ".join("%s=%s" % kv for kv in env.items()) or "default"], env=e, capture_output=True, text=True, timeout=120)
        print(r.stdout.strip() or r.stderr[-400:])
