"""BASELINE configs[4] at its stated size on ONE GPU: 4096 key frames x 8000 descriptors, all 8 386 560 unordered pairs through
ssvio_amd.dist.loop_candidates (one native ssvio_loop_candidates_dev walk, MatchByBruteForce per pair-block, device-side
{pair, n_matches, sum_dist} records).  Checks that records 0..2015 equal those of the 64-key-frame subset (the bench line's
configs[4] entry) run on its own.  python tools/config5_full.py [valu|mfma] [nsets] [sb] > json"""
import json, os, sys, threading, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import ssvio_amd
from ssvio_amd import dist as D, _lib as L

kernel = sys.argv[1] if len(sys.argv) > 1 else "mfma"
nsets = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
sb = int(sys.argv[3]) if len(sys.argv) > 3 else 32
rows = 8000
dev = torch.device("cuda:0")
free0, total = torch.cuda.mem_get_info()

# key frames = noisy, permuted copies of one base set (so that MatchByBruteForce keeps a non-trivial list), generated on the device
g = torch.Generator(device=dev); g.manual_seed(5)
base = torch.randint(0, 256, (rows, 32), dtype=torch.uint8, device=dev, generator=g)
desc = torch.empty((nsets, rows, 32), dtype=torch.uint8, device=dev)
w = (2 ** torch.arange(8, device=dev, dtype=torch.int32)).view(1, 1, 1, 8)
for s0 in range(0, nsets, 32):
    n = min(32, nsets - s0)
    perm = torch.stack([torch.randperm(rows, device=dev, generator=g) for _ in range(n)])
    flips = torch.rand((n, rows, 32, 8), device=dev, generator=g) < 0.08
    fresh = torch.rand((n, rows, 1, 1), device=dev, generator=g) < 0.3                  # 30 % of the rows are unrelated to the base
    noise = ((flips | (fresh & (torch.rand((n, rows, 32, 8), device=dev, generator=g) < 0.5))).int() * w).sum(-1).to(torch.uint8)
    desc[s0:s0 + n] = base[perm] ^ noise
    del perm, flips, fresh, noise
cnt = torch.full((nsets,), rows, dtype=torch.int32, device=dev)
torch.cuda.synchronize()

def run(ns):
    fm = ssvio_amd.FeatureMatching(None, {"Threshold": 50, "NNRatio": 0.6})
    fm.set_kernel(kernel)
    blocks = D.DeviceBlocks(fm, desc, rows, rows, cnt, dev)
    D.loop_candidates(blocks, min(ns, 8), 0, 1, sb=sb); torch.cuda.synchronize()        # warm-up: allocations, code load
    stop = threading.Event()
    t0 = time.perf_counter()
    def ticker():
        while not stop.wait(20):
            print("... %d key frames, %.0f s" % (ns, time.perf_counter() - t0), file=sys.stderr, flush=True)
    th = threading.Thread(target=ticker, daemon=True); th.start()
    rec = D.loop_candidates(blocks, ns, 0, 1, sb=sb)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stop.set(); th.join()
    free1, _ = torch.cuda.mem_get_info()
    r = D.records_to_numpy(rec)
    fm.close()
    return r, dt, free0 - free1

sub, dt_sub, _ = run(64)
full, dt, used = run(nsets)
npairs = nsets * (nsets - 1) // 2
assert full["pair"].tolist() == list(range(npairs))
same = bool(np.array_equal(full[:2016].view(np.uint8), sub.view(np.uint8))) if nsets >= 64 else None
print(json.dumps({
    "config": "configs[4] at full size: all %d unordered pairs of %d key frames x %d descriptors on ONE MI355X, MatchByBruteForce per pair-block, "
              "super-blocks of %dx%d key frames, one ssvio_loop_candidates_dev walk (%s kernel), 16-byte records on the device" % (npairs, nsets, rows, sb, sb, kernel),
    "kernel": kernel, "seconds": round(dt, 2), "pair_blocks": npairs, "descriptor_pairs_per_s": npairs * float(rows) * rows / dt,
    "matches": int(full["n_matches"].sum()), "sum_dist": int(full["sum_dist"].sum()),
    "peak_device_bytes_in_use": int(used), "descriptor_bytes": int(desc.numel()), "record_bytes": npairs * 16,
    "subset_64_seconds": round(dt_sub, 4), "subset_64_descriptor_pairs_per_s": 2016 * float(rows) * rows / dt_sub,
    "records_0_2015_equal_subset": same}))
