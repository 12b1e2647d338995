#!/usr/bin/env python
"""bench.py -- BASELINE config 2 on MI355X: VGA stream, 1000 features/frame, per-frame ORB
extraction + MatchByBruteForce against the previous frame.

A step = one batch of `--batch` synthetic VGA frames already resident in HBM:
extract_batch_dev (pyramid, FAST cells, quadtree, blur, IC angle + rBRIEF) followed by one
batched brute-force match launch (frame f vs frame f-1; frame 0 wraps to the batch's last frame
so every frame is matched once).  value = frames/s over all ranks.  With --gpus N every rank
runs the same workload on its own frames (frames are independent units: weak scaling, no
data-path collective); the per-rank frame counts are gathered once at the end.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the longest launch of a step), measured with hipEvent pairs on the library's stream; `cpu_baseline` is the
CPU oracle (oracle/, a port of the reference algorithm) timed single-threaded on a bounded sample.

`value` is the DEVICE-RESIDENT rate (frames already in HBM, the contract's definition).  The same line carries, measured
in the same process at N = 1 (`configs`): BASELINE config 1 (one frame per call, host waits), config 2 streamed (256
distinct frames from pinned host memory, H2D copies on a copy stream inside the timed region), config 3 (1080p, 4000
features, kNN2 ratio), a `sustained` figure over >= 1 s, and `batched_matching` (configs 4 and 5 through
ssvio_amd.dist, the code an 8-GPU run executes).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# VALU issue cost of one (query, train) descriptor pair in k_hamming_knn2 at the two issue rates of the chip: 2 cycles per wave
# instruction per SIMD for v_xor_b32, 4 for v_bcnt_u32_b32 / v_lshl_or_b32 / v_med3_u32 / v_min_u32.  (The dependent-chain
# micro-benchmarks of profiles/r1/ubench_valu_rate.txt read 2.46 and 4.13; rounds 1-3 priced the roof with those, until the kernel's
# nearest-neighbour form ran the full configs[4] walk at 2.66e12 pairs/s, 3 % above that price: the nominal rates are the roof.)
MATCH_CYCLES_PER_PAIR = 8 * 2.0 + 11 * 4.0                          # 8 xor, 8 bcnt, key, median, min: the 2-NN form (ratio test, raw lists)
VALU_PEAK_PAIRS = 256 * 4 * 2.4e9 * 64 / MATCH_CYCLES_PER_PAIR      # 1024 SIMDs x 2.4 GHz x 64 lanes = 2.62e12 pairs/s
# MatchByBruteForce and the cross-checks read the nearest neighbour only: the kernel's SECOND = false form drops the median
MATCH_CYCLES_PER_PAIR_NN = 8 * 2.0 + 10 * 4.0
VALU_PEAK_PAIRS_NN = 256 * 4 * 2.4e9 * 64 / MATCH_CYCLES_PER_PAIR_NN    # 2.81e12 pairs/s
# k_hamming_knn2_mfma: 8 x v_mfma_i32_32x32x32_i8 (32 cycles each per SIMD, MI355X_MICROARCH.md matrix table) per 32 x 32 descriptor pairs
MFMA_PEAK_PAIRS = 256 * 4 * 2.4e9 * 1024 / (8 * 32)                 # 9.83e12 pairs/s at the nominal clock


DEFAULT_BATCH = 768            # frames per step of the default line
PROFILE_DIR = os.path.join(ROOT, "profiles", "r3")
PMC_FILE = os.path.join(PROFILE_DIR, "r3_final_pmc_hbm.json")


def kernel_source_hash():
    """sha256 over the kernel sources: the committed rocprof / PMC figures are only carried into the line while this matches
    the hash stored beside them (profiles/r3/r3_final_source_hash.txt, written by tools/gpu_profile.sh) -- a kernel change
    without a fresh profile turns them into nulls instead of replaying stale numbers"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "ssvio_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "ssvio_amd", "csrc", "*.h"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def profiles_current():
    try:
        return open(os.path.join(PROFILE_DIR, "r3_final_source_hash.txt")).read().strip() == kernel_source_hash()
    except OSError:
        return False


def pmc_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 --pmc passes over this same command
    (tools/gpu_profile.sh: FETCH_SIZE and WRITE_SIZE in separate runs, FETCH_SIZE doubled on gfx950 as
    MI355X_MICROARCH.md prescribes).  bench.py cannot run the profiler on itself; None when the file is absent."""
    try:
        d = json.load(open(PMC_FILE))
    except (OSError, ValueError):
        return None
    for name, c in d.items():
        if name.split("::")[-1].split("<")[0] in (kernel, kernel + "_quad") and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            return int(c["FETCH_SIZE"]["avg_bytes_corrected"] + c["WRITE_SIZE"]["avg_bytes_corrected"])
    return None


STATS_FILE = os.path.join(PROFILE_DIR, "r3_final_kernel_stats.csv")


STATS_1080P_FILE = os.path.join(PROFILE_DIR, "r3_1080p_kernel_stats.csv")


def rocprof_avg_us(kernel, stats_file=None):
    """average duration of `kernel` in the committed `rocprofv3 --kernel-trace --stats` summary of this same command
    (tools/gpu_profile.sh), for comparison with the live hipEvent figures; None when the file is absent"""
    import csv
    try:
        for r in csv.DictReader(open(stats_file or STATS_FILE)):
            if r["Name"].replace("void ", "").startswith("ssvio::" + kernel + "<") or r["Name"].replace("void ", "").startswith("ssvio::" + kernel + "("):
                return round(float(r["AverageNs"]) / 1e3, 2)
    except (OSError, ValueError, KeyError):
        pass
    return None


SQ_FILE = os.path.join(PROFILE_DIR, "r3_final_pmc_sq.txt")
MIX_FILE = os.path.join(PROFILE_DIR, "valu_mix.json")


def valu_busy(kernel, kernel_us):
    """share of the chip's VALU issue capacity the kernel uses: SQ_INSTS_VALU (wave-instructions, summed over the chip; committed
    rocprofv3 --pmc pass, tools/gpu_pmc.sh) x the kernel's average issue cycles per instruction / (1024 SIMDs x kernel cycles at
    2.4 GHz).  The issue cycles come from the two classes measured on this chip (profiles/r2/ubench_valu_table.txt: ~2.4 cycles
    for and / or / xor / add / sub / right shifts / mov, ~4.2 for the rest, ~5 for v_cmp) weighted by the kernel's static
    instruction mix (tools/valu_mix.py -> profiles/r3/valu_mix.json).  -> (busy, average issue cycles) or (None, None)"""
    import re
    try:
        mix = json.load(open(MIX_FILE))
        cyc = next(v["avg_issue_cycles"] for k, v in sorted(mix.items(), reverse=True) if k == kernel or k.startswith(kernel + "<"))
        for line in open(SQ_FILE):
            if line.startswith(kernel + "<") or line.startswith(kernel + " "):
                m = re.search(r"SQ_INSTS_VALU=([0-9.e+]+)", line)
                if m and kernel_us > 0:
                    return round(float(m.group(1)) * cyc / (1024 * kernel_us * 2400.0), 3), cyc
    except (OSError, ValueError, StopIteration, KeyError):
        pass
    return None, None


def level_pixels(ex, w, h):
    return [ex.level_geometry(w, h, l)["w"] * ex.level_geometry(w, h, l)["h"] for l in range(ex.nlevels)]


def algorithmic_bytes(kernel, B, px, nkp_total, pairs_rows):
    """SURVEY.md §8(d) per-unit figures split per kernel; bytes for ONE launch over the batch."""
    P = sum(px)
    if kernel in ("k_fast_cells", "k_fast_score+nms", "k_fast"):
        return B * P                                   # every pyramid pixel read once
    if kernel == "k_blur":
        return B * 2 * P                               # read level, write blurred level
    if kernel == "k_resize":
        # 7 launches per step; average launch: read level l-1, write level l
        return B * (sum(px[:-1]) + sum(px[1:])) / max(len(px) - 1, 1)
    if kernel == "k_describe":
        return nkp_total * (600 + 512 + 60)            # IC patch + rBRIEF samples read, keypoint+descriptor written
    if kernel in ("k_copy_gray", "k_rgb2gray"):
        return B * 2 * px[0]
    if kernel in ("k_compact", "k_quadtree"):
        return nkp_total * 8 * 6                       # ~6 candidates per kept keypoint, read + write 4 B
    if kernel == "k_hamming_knn2":
        return sum(32 * (nq + nt) + 16 * nq for nq, nt in pairs_rows)
    if kernel == "k_match_finalize":
        return sum(8 * nq + 16 * nq for nq, nt in pairs_rows)
    if kernel == "k_bow_transform":
        return nkp_total * (32 + 6 * 10 * 32 + 20)     # descriptor + 6 levels x 10 child descriptors read, word/node/weight written
    if kernel == "k_match_grouped":
        return sum(36 * (nq + nt) + 24 * nq for nq, nt in pairs_rows)
    return 0


def _pcts(ms):
    ms = sorted(ms)
    return {"median_ms_per_frame": round(ms[len(ms) // 2], 3), "p95_ms_per_frame": round(ms[min(len(ms) - 1, int(0.95 * len(ms)))], 3)} if ms else {}


_MATCHERS = {"bf": "match_bruteforce", "ratio": "match_knn2_ratio", "crosscheck": "match_crosscheck", "dbow": "match_knn2_ratio"}   # dbow: the CPU leg times the all-rows 2-NN loop (no 1.1 M-node tree on the host side)


def cpu_baseline(frames, nfeatures, match_mode="bf", budget_s=10.0):
    """oracle extract + match vs previous frame, 1 thread (the reference's application-level behaviour), bounded sample of the same
    frames; median and p95 per frame"""
    from oracle import oracle as O
    ex = O.Extractor(nfeatures, 1.2, 8, 20, 7)
    fn = getattr(O, _MATCHERS[match_mode])
    prev = None
    n = 0
    per = []
    t0 = time.perf_counter()
    while True:
        ta = time.perf_counter()
        k, d = ex.extract(frames[n % len(frames)])
        if prev is not None:
            fn(d, prev) if match_mode in ("bf", "crosscheck") else fn(d, prev, 50, 0.6)
            per.append((time.perf_counter() - ta) * 1e3)
        prev = d
        n += 1
        if time.perf_counter() - t0 > budget_s or n >= 8 * len(frames):
            break
    dt = time.perf_counter() - t0
    out = {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": "%d frames of %dx%d (extract %d features + %s match vs previous), oracle C restatement, 1 thread, %.1f s"
                     % (n, frames.shape[2], frames.shape[1], nfeatures, match_mode, dt)}
    out.update(_pcts(per))
    return out


def cpu_baseline_all_cores(frames, nfeatures, match_mode="bf", budget_s=6.0):
    """BASELINE.md's second CPU point: the same oracle work spread over the host cores this job may use (one oracle
    extractor per thread; ctypes releases the GIL inside the C calls).  Reported beside cpu_baseline, never the target."""
    import concurrent.futures as cf
    from oracle import oracle as O
    cores = max(1, min(int(os.environ.get("OMP_NUM_THREADS", "16")), os.cpu_count() or 1, 16))
    fn = getattr(O, _MATCHERS[match_mode])
    deadline = time.perf_counter() + budget_s

    def worker(tid):
        ex = O.Extractor(nfeatures, 1.2, 8, 20, 7)
        prev = None; per = []
        while time.perf_counter() < deadline:
            ta = time.perf_counter()
            k, d = ex.extract(frames[(tid + len(per) * cores) % len(frames)])
            if prev is not None:
                fn(d, prev) if match_mode in ("bf", "crosscheck") else fn(d, prev, 50, 0.6)
            prev = d; per.append((time.perf_counter() - ta) * 1e3)
        return per

    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(cores) as pool:
        pers = list(pool.map(worker, range(cores)))
    dt = time.perf_counter() - t0
    done = sum(len(p) for p in pers)
    out = {"value": done / dt, "unit": "frames/s", "cores": cores, "kind": "port",
           "sample": "%d frames in %.1f s, one oracle extractor per thread" % (done, dt)}
    out.update(_pcts([m for p in pers for m in p]))
    return out


def cpu_points(frames, nfeatures, match_mode, b1, bn):
    """the two CPU points BASELINE.md plans per config"""
    return {"one_thread": cpu_baseline(frames, nfeatures, match_mode, b1), "all_cores": cpu_baseline_all_cores(frames, nfeatures, match_mode, bn)}


def mfma_roofline(pairs_per_s, world):
    """k_hamming_knn2_mfma against the dense int8 matrix peak: a descriptor pair is 256 multiply-adds = 512 ops"""
    peak = MFMA_PEAK_PAIRS * 512 / 1e12 * world
    ach = pairs_per_s * 512 / 1e12
    return {"kernel": "k_hamming_knn2_mfma", "bound": "mfma", "achieved": round(ach, 1), "peak": round(peak, 1), "unit": "TOP/s", "frac": round(ach / peak, 4),
            "peak_source": "1024 SIMDs x one v_mfma_i32_32x32x32_i8 (65 536 int8 ops) per 32 cycles x 2.4 GHz = 5.03 POP/s dense (MI355X_MICROARCH.md, matrix table: int8 = 2 x bf16); "
                           "whole-call time (both directions of a cross-check, merge, filter and summary launches included), so the kernel alone is higher"}


def batched_matching(rank, world, dev, dist, iters=10):
    """BASELINE configs 4 and 5 through ssvio_amd.dist -- the package code an 8-GPU run executes and that
    tests/test_dist_gloo.py (gloo, world 2) and tests/test_gpu_dist.py (one MI355X, vs the oracle) cover: keyframe-pair blocks dealt
    to the ranks, every rank matches its share on its GPU, and ONE all_gather (RCCL over xGMI when world > 1) returns the
    best-match lists (config 4) or the 16-byte per-pair summaries reduced on the device (config 5).  Outside the frames/s timed
    region; reported beside it."""
    import ssvio_amd
    from ssvio_amd import synth, dist as D, _lib as L
    fm = ssvio_amd.FeatureMatching(None, {"Threshold": 50, "NNRatio": 0.6}, device=dev.index)
    out = []

    def timed(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        if dist: dist.barrier()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        if dist: dist.barrier()
        dt = time.perf_counter() - t0
        if dist:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt / iters

    # ---- config 4: 1 query frame x 64 keyframes x 4000 descriptors, cross-check; the 64 pair-blocks are split over the ranks
    rows, nkf = 4000, 64
    q, base = synth.descriptor_sets(400, rows * nkf, rows)
    d = torch.from_numpy(np.concatenate([base, q])).to(dev)             # set 0 = query frame, sets 1..64 = keyframes
    cnt = torch.full((nkf + 1,), rows, dtype=torch.int32, device=dev)
    pairs = np.array([[0, k + 1] for k in range(nkf)], np.int32)
    blocks = D.DeviceBlocks(fm, d, rows, rows, cnt, dev)
    work = {}
    res = {}

    def cfg4():
        res["lists"], res["counts"] = D.window_matches(blocks, pairs, rank, world, L.MODE_CROSSCHECK, work=work)

    trace('cfg4 start')
    dt = timed(cfg4)
    trace('cfg4 done')
    total = int(res["counts"].sum().item())
    fm.set_kernel("mfma")
    dtm = timed(cfg4)
    totalm = int(res["counts"].sum().item())
    fm.set_kernel("valu")
    per = (nkf + world - 1) // world
    out.append({"config": "configs[3]: 1 query x %d keyframes x %d descriptors, cross-check, pair-blocks over %d rank(s) "
                          "(ssvio_amd.dist.window_matches), all_gather of DMatch lists (%d KB per rank)" % (nkf, rows, world, (per * rows * 16 + per * 4) // 1024),
                "scaling": "strong", "kernel": "k_hamming_knn2 (VALU xor + popcount, the library's default)",
                "ms": round(dt * 1e3, 4), "descriptor_pairs_per_s": 2.0 * nkf * rows * rows / dt,
                "query_matches_per_s": total / dt, "matches": total,
                "mfma_kernel": {"kernel": "k_hamming_knn2_mfma (SSVIO_KERNEL_MFMA: exact int8 contraction, same results)", "ms": round(dtm * 1e3, 4),
                                "descriptor_pairs_per_s": 2.0 * nkf * rows * rows / dtm, "matches": totalm, "speedup": round(dt / dtm, 3),
                                "roofline": mfma_roofline(2.0 * nkf * rows * rows / dtm, world)}})

    # ---- config 5 (timed subset, stated): loop-candidate all-pairs over 64 of the 4096 keyframes, 8000 descriptors each:
    # 2016 pair-blocks of 8000 x 8000 walked in super-blocks of 8 x 8 keyframes (36 tiles dealt to the ranks), per-pair summary
    # records reduced on the device, one all_gather.  The full set is 4160x this many pair-blocks of the same size.
    rows5, nset5, sb5 = 8000, 64, 8
    q5, _ = synth.descriptor_sets(500, rows5 * nset5, rows5)
    d5 = torch.from_numpy(q5).to(dev)
    cnt5 = torch.full((nset5,), rows5, dtype=torch.int32, device=dev)
    blocks5 = D.DeviceBlocks(fm, d5, rows5, rows5, cnt5, dev)
    work5 = {}

    def cfg5():
        res["rec"] = D.loop_candidates(blocks5, nset5, rank, world, sb=sb5, work=work5)

    iters = 3
    dt5 = timed(cfg5)
    trace('cfg5 done')
    recv = D.records_to_numpy(res["rec"])
    fm.set_kernel("mfma")
    dt5m = timed(cfg5)
    fm.set_kernel("valu")
    rec = D.records_to_numpy(res["rec"])
    assert np.array_equal(rec.view(np.uint8), recv.view(np.uint8)), "the two matcher kernels disagree"
    npairs5 = nset5 * (nset5 - 1) // 2
    assert rec["pair"].tolist() == list(range(npairs5))
    total5 = int(rec["n_matches"].sum())
    plan, load = D.deal_super_blocks(nset5, sb5, world)
    out.append({"config": "configs[4] subset: all %d unordered pairs of %d keyframes x %d descriptors (of 4096 keyframes), MatchByBruteForce, super-blocks of "
                          "%dx%d keyframes dealt to %d rank(s) (ssvio_amd.dist.loop_candidates), device-side {pair, n, sum_dist} records, one all_gather "
                          "(%d KB per rank)" % (npairs5, nset5, rows5, sb5, sb5, world, max(load) * 16 // 1024),
                "scaling": "strong", "ms": round(dt5 * 1e3, 4), "pair_blocks_per_rank": load,
                "descriptor_pairs_per_s": float(npairs5) * rows5 * rows5 / dt5, "query_matches_per_s": total5 / dt5, "matches": total5,
                "sum_dist": int(rec["sum_dist"].sum()),
                "kernel": "k_hamming_knn2 (VALU xor + popcount, the library's default)",
                "valu_frac": round(float(npairs5) * rows5 * rows5 / dt5 / (world * VALU_PEAK_PAIRS_NN), 4),
                "mfma_kernel": {"kernel": "k_hamming_knn2_mfma (SSVIO_KERNEL_MFMA), identical records", "ms": round(dt5m * 1e3, 4),
                                "descriptor_pairs_per_s": float(npairs5) * rows5 * rows5 / dt5m, "speedup": round(dt5 / dt5m, 3),
                                "roofline": mfma_roofline(float(npairs5) * rows5 * rows5 / dt5m, world),
                                "clock_note": "the chip holds ~2.07 GHz under this kernel (GRBM_GUI_ACTIVE, profiles/r3/r3_mfma_pmc.txt), the peak is priced at the nominal 2.4"},
                "full_config_measured_1gpu": "profiles/r3/config5_full_1gpu.json: all 8 386 560 pair-blocks on ONE MI355X, 249.1 s (VALU kernel) / 74.3 s (MFMA kernel)",
                "full_config_extrapolated_s": round(dt5 * (4096 * 4095 / 2) / npairs5 * world / 8, 1),
                "full_config_note": "8 386 560 pair-blocks on 8 GPUs at this per-GPU rate (VALU kernel)"})
    fm.close()
    return out


class Pipeline:
    """S handle pairs (extractor + matcher), each pair on its own HIP stream; step i runs on pair i % S"""

    def __init__(self, dev, features, S, threshold=50, ratio=0.6, torch_streams=False):
        import ssvio_amd
        self.exs = [ssvio_amd.ORBextractor(features, 1.2, 8, 20, 7, device=dev.index) for _ in range(S)]
        self.fms = [ssvio_amd.FeatureMatching(None, {"Threshold": threshold, "NNRatio": ratio}, device=dev.index) for _ in range(S)]
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(S)] if torch_streams else None
        for i, (e, m) in enumerate(zip(self.exs, self.fms)):
            if torch_streams:
                e.set_stream(self.streams[i].cuda_stream)
            m.set_stream(e.get_stream())
        self.S, self.n = S, 0

    def step(self, d_frames, B, W, H, pairs, mode, wait=False):
        i = self.n % self.S
        e, m = self.exs[i], self.fms[i]
        self.n += 1
        e.extract_batch_dev(d_frames, B, W, H, wait=wait and mode is None)      # with a matcher behind it (same stream) the one wait is below
        if mode is None:                                  # extraction alone
            return i
        _, d_desc, d_n, cap = e.device_results()
        m.match_batch_dev(d_desc, cap, cap, d_n, pairs, mode)
        if wait:
            m.sync()
        return i

    def sync(self):
        for e, m in zip(self.exs, self.fms):
            m.sync(); e.sync()

    def close(self):
        for h in self.fms + self.exs:                     # the matchers run on their extractors' streams: they go first
            h.close()


def extra_configs(dev, frames_vga, with_cpu=True):
    """The single-GPU BASELINE configs that are not the headline, measured in this same process (N = 1 only): every one is its own
    timed region with a device synchronise on both sides."""
    from ssvio_amd import synth, _lib as L
    out = []
    W, H = 640, 480

    # ---- configs[0] on the GPU: one VGA frame per step, 500 features, BF match vs the previous frame, host waits for every frame's matches (one wait)
    p = Pipeline(dev, 500, 1)
    d1 = torch.from_numpy(frames_vga[:16]).to(dev)
    pairs1 = np.array([[0, 0]], np.int32)
    for k in range(20):
        p.step(d1[k % 16], 1, W, H, pairs1, L.MODE_BRUTEFORCE, wait=True)
    torch.cuda.synchronize()
    n1 = 300
    t0 = time.perf_counter()
    for k in range(n1):
        p.step(d1[k % 16], 1, W, H, pairs1, L.MODE_BRUTEFORCE, wait=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kp1 = float(p.exs[0].last_counts(1)[0])
    p.close()
    # the same through the host-buffer entry points the unmodified Odometry loop would call (pageable numpy in and out: PCIe inclusive)
    import ssvio_amd
    ex = ssvio_amd.ORBextractor(500, 1.2, 8, 20, 7, device=dev.index); fm = ssvio_amd.FeatureMatching(None, {}, device=dev.index)
    prev = ex.Extract(frames_vga[0])[1]
    for k in range(10):
        prev = ex.Extract(frames_vga[k % 16])[1]
    t0 = time.perf_counter()
    for k in range(100):
        _, de = ex.Extract(frames_vga[k % 16])
        fm.MatchByBruteForce(de, prev)
        prev = de
    dth = time.perf_counter() - t0
    ex.close(); fm.close()
    out.append({"config": "configs[0] on the GPU: one 640x480 frame per step, 500 features, extract + MatchByBruteForce vs the previous frame enqueued back to back, host waits for every frame's matches",
                "value": round(n1 / dt, 1), "unit": "frames/s", "ms_per_frame": round(dt / n1 * 1e3, 4), "keypoints_per_frame": kp1,
                "residency": "device-resident frames and results",
                "cpu": cpu_points(frames_vga[:16], 500, "bf", 4.0, 4.0) if with_cpu else None,
                "host_buffer_api": {"value": round(100 / dth, 1), "unit": "frames/s", "ms_per_frame": round(dth / 100 * 1e3, 4),
                                    "what": "ORBextractor.Extract + MatchByBruteForce on pageable host arrays (PCIe inclusive, python wrappers)"}})

    # ---- configs[1] streamed: 256 distinct frames in pinned host memory, H2D copies on a copy stream, double-buffered against compute
    # (round 3, batch / in flight / spare buffers: 128/3/1 155.7 k, 128/3/3 154.9 k, 256/3/3 158.9 k, 128/4/4 132.1 k, 256/4/2 139.3 k, 128/6/6 150.3 k frames/s -- the copy stream wants a hardware queue of its own)
    B, S, nfr = int(os.environ.get("SSVIO_BENCH_STREAMED_BATCH", "256")), int(os.environ.get("SSVIO_BENCH_STREAMED_IN_FLIGHT", "3")), 512
    frames = synth.frame_stream(2000, nfr, H, W)
    pinned = torch.from_numpy(frames).pin_memory()
    p = Pipeline(dev, 1000, S, torch_streams=True)
    if os.environ.get("SSVIO_BENCH_STREAMED_GRAPH", "0") == "1":
        for e in p.exs: e.set_graph(True)       # one submission per step beside the copies: no gain (162 k against 163-164 k frames/s over 0.5-s regions; 0.1-s regions scatter by +-10 %)
    ncopy = int(os.environ.get("SSVIO_BENCH_COPY_STREAMS", "1"))      # a batch goes up in ncopy slices, one copy stream each (measured: 1 -> 122 k, 2 -> 82 k, 4 -> 65 k frames/s: more streams crowd the 4 hardware queues)
    copy_sts = [torch.cuda.Stream(device=dev) for _ in range(ncopy)]
    nbuf = S + int(os.environ.get("SSVIO_BENCH_STREAMED_SPARE_BUFFERS", "3"))
    d_in = [torch.empty((B, H, W), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    ready = [[torch.cuda.Event() for _ in range(ncopy)] for _ in range(nbuf)]; free = [torch.cuda.Event() for _ in range(nbuf)]
    used = [False] * nbuf
    pairs = np.array([[f, (f - 1) % B] for f in range(B)], np.int32)
    state = {"i": 0}

    def sstep():
        i = state["i"]; state["i"] += 1
        b = i % nbuf; chunk = i % (nfr // B)
        cs = p.streams[p.n % S]
        for c, copy_st in enumerate(copy_sts):
            f0, f1 = c * B // ncopy, (c + 1) * B // ncopy
            with torch.cuda.stream(copy_st):
                if used[b]:
                    copy_st.wait_event(free[b])                   # the step that read this buffer has finished
                d_in[b][f0:f1].copy_(pinned[chunk * B + f0:chunk * B + f1], non_blocking=True)
                ready[b][c].record(copy_st)
            cs.wait_event(ready[b][c])
        k = p.step(d_in[b], B, W, H, pairs, L.MODE_BRUTEFORCE)
        free[b].record(p.streams[k]); used[b] = True

    for _ in range(8):
        sstep()
    p.sync(); torch.cuda.synchronize()
    ns = 300 if B >= 256 else 600          # ~0.5 s: shorter regions scatter by +-10 % from run to run
    t0 = time.perf_counter()
    for _ in range(ns):
        sstep()
    p.sync(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kp = float(p.exs[0].last_counts(B).mean())
    p.close()
    out.append({"config": "configs[1] streamed: %d distinct 640x480 frames in pinned host memory, %d-frame steps, H2D on %d copy stream(s) into %d device buffers "
                          "ahead of %d steps in flight, 1000 features, extract + MatchByBruteForce vs previous frame" % (nfr, B, ncopy, nbuf, S),
                "value": round(ns * B / dt, 1), "unit": "frames/s", "ms_per_step": round(dt / ns * 1e3, 4), "steps": ns,
                "residency": "PCIe feed inside the timed region", "h2d_GBps": round(ns * B * W * H / dt / 1e9, 2), "keypoints_per_frame": kp})

    # ---- configs[2]: 1920x1080, 4000 features, extract + kNN2 ratio test (Threshold 50, NNRatio 0.6), 16-frame steps, 3 in flight
    W3, H3, B3 = 1920, 1080, int(os.environ.get("SSVIO_BENCH_1080P_BATCH", "32"))
    f3 = synth.frame_stream(3000, B3, H3, W3)
    d3 = torch.from_numpy(f3).to(dev)
    S3 = int(os.environ.get("SSVIO_BENCH_1080P_IN_FLIGHT", "4"))
    p = Pipeline(dev, 4000, S3)
    pairs3 = np.array([[f, (f - 1) % B3] for f in range(B3)], np.int32)
    for _ in range(6):
        p.step(d3, B3, W3, H3, pairs3, L.MODE_KNN2_RATIO)
    p.sync(); torch.cuda.synchronize()
    n3 = 60
    t0 = time.perf_counter()
    for _ in range(n3):
        p.step(d3, B3, W3, H3, pairs3, L.MODE_KNN2_RATIO)
    p.sync(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kp3 = float(p.exs[0].last_counts(B3).mean())
    _, nm3 = p.fms[(p.n - 1) % S3].fetch_batch(counts_only=True)
    # the same steps with the matrix-core matcher (identical match lists)
    for m in p.fms: m.set_kernel("mfma")
    for _ in range(S3):
        p.step(d3, B3, W3, H3, pairs3, L.MODE_KNN2_RATIO)
    p.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n3):
        p.step(d3, B3, W3, H3, pairs3, L.MODE_KNN2_RATIO)
    p.sync(); torch.cuda.synchronize()
    dt3m = time.perf_counter() - t0
    _, nm3m = p.fms[(p.n - 1) % S3].fetch_batch(counts_only=True)
    assert np.array_equal(nm3, nm3m), "the two matcher kernels disagree"
    for m in p.fms: m.set_kernel("valu")
    # per-kernel exclusive times at this size: the same steps on one handle pair, one in flight, event pairs around every kernel
    e3, m3 = p.exs[0], p.fms[0]
    e3.set_profiling(True); m3.set_profiling(True)
    for _ in range(6):
        e3.extract_batch_dev(d3, B3, W3, H3, wait=False)
        _, dd, dn, cap3 = e3.device_results()
        m3.match_batch_dev(dd, cap3, cap3, dn, pairs3, L.MODE_KNN2_RATIO)
        m3.sync()
    t3 = dict(e3.kernel_times()); t3.update(m3.kernel_times())
    e3.set_profiling(False); m3.set_profiling(False)
    px3 = level_pixels(e3, W3, H3)
    cnt3 = e3.last_counts(B3)
    rows3 = [(int(cnt3[a]), int(cnt3[b])) for a, b in pairs3]
    k3 = {k: {"avg_us": round(ms * 1e3 / max(n, 1), 2), "launches": n,
              "alg_bytes_per_launch": int(algorithmic_bytes(k, B3, px3, int(cnt3.sum()), rows3))} for k, (ms, n) in t3.items()}
    # the HBM roofline goes to the longest launch among the kernels that stream the pyramid; the matcher's launch (as long as k_fast's at
    # this size) moves 10 MB and is priced against the VALU roof that binds it instead
    mk3 = [k for k in k3 if k.startswith("k_hamming_knn2")]
    dom3 = max((k for k in k3 if not k.startswith("k_hamming_knn2") and not k.startswith("k_match")), key=lambda k: k3[k]["avg_us"])
    pairs3n = float(sum(a * b for a, b in rows3))
    matcher3 = {k: {"avg_us": k3[k]["avg_us"], "descriptor_pairs_per_s": round(pairs3n / (k3[k]["avg_us"] * 1e-6), 0), "bound": "valu",
                    "frac_of_valu_roof": round(pairs3n / (k3[k]["avg_us"] * 1e-6) / VALU_PEAK_PAIRS, 4)} for k in mk3}
    alg_step3 = B3 * (W3 * H3 + 3 * sum(px3)) + int(cnt3.sum()) * (1112 + 60)
    roof3 = {"kernel": dom3, "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
             "achieved": round(k3[dom3]["alg_bytes_per_launch"] / (k3[dom3]["avg_us"] * 1e-6) / 1e9, 2),
             "frac": round(k3[dom3]["alg_bytes_per_launch"] / (k3[dom3]["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 5),
             "avg_launch_us": k3[dom3]["avg_us"], "alg_bytes_per_launch": k3[dom3]["alg_bytes_per_launch"], "frames_per_launch": B3,
             "timing": "hipEvent pair on the library's stream, one step in flight (exclusive)",
             "rocprof_avg_launch_us": rocprof_avg_us(dom3, STATS_1080P_FILE),
             "rocprof_source": "profiles/r3/r3_1080p_kernel_stats.csv (rocprofv3 --kernel-trace --stats of bench.py --width 1920 --height 1080 --features 4000 --match-mode ratio --batch 32)",
             "alg_bytes_per_frame": alg_step3 // B3,
             "pipeline_frac": round(alg_step3 / (dt / n3) / 1e9 / HBM_PEAK_GBS, 5),
             "matcher_launch": matcher3,
             "kernels": k3}
    p.close()
    out.append({"config": "configs[2]: 1920x1080, 4000 features, 8 levels, extract + kNN2 ratio test vs previous frame, %d-frame steps, %d in flight" % (B3, S3),
                "value": round(n3 * B3 / dt, 1), "unit": "frames/s", "ms_per_step": round(dt / n3 * 1e3, 4), "steps": n3,
                "residency": "device-resident", "keypoints_per_frame": kp3, "matches_per_frame": round(float(nm3.mean()), 1),
                "mfma_matcher": {"value": round(n3 * B3 / dt3m, 1), "unit": "frames/s", "ms_per_step": round(dt3m / n3 * 1e3, 4),
                                 "what": "the same steps with ssvio_matcher_set_kernel(SSVIO_KERNEL_MFMA); identical match counts"},
                "roofline_1080p": roof3,
                "cpu": cpu_points(f3, 4000, "ratio", 6.0, 6.0) if with_cpu else None})

    # ---- ORB extraction alone (BASELINE's metric names it at both sizes): the same pipelines without the matcher launches
    def extract_only(d, Bx, Wx, Hx, features, S, n):
        p = Pipeline(dev, features, S)
        for _ in range(2 * S):
            p.step(d, Bx, Wx, Hx, None, None)
        p.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            p.step(d, Bx, Wx, Hx, None, None)
        p.sync(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        p.close()
        return round(n * Bx / dt, 1)
    Bv = int(os.environ.get("SSVIO_BENCH_EXTRACT_BATCH", str(min(len(frames_vga), DEFAULT_BATCH))))
    dv = torch.from_numpy(np.ascontiguousarray(frames_vga[:Bv])).to(dev) if Bv <= len(frames_vga) else torch.from_numpy(synth.frame_stream(1000, Bv, H, W)).to(dev)
    out.append({"config": "ORB extraction alone, device-resident: 640x480 / 1000 features in %d-frame steps, 1920x1080 / 4000 features in %d-frame steps, 4 in flight" % (Bv, B3),
                "value": extract_only(dv, Bv, W, H, 1000, 4, max(20, 15360 // Bv)), "unit": "frames/s", "value_1080p": extract_only(d3, B3, W3, H3, 4000, 4, 60)})
    return out


def next_rows(ex, fm, d_frames, B, W, H, pairs, dev, iters=20):
    """SURVEY.md §8 "next" rows on the bench's own batch, device-resident: MatchByDBoW (f1) on a synthetic vocabulary of
    ORBvoc's shape, the correspondence lists of Odometry::OptimizeTransform (f2) and the per-pixel cloud (f4, first half).
    Per-kernel GPU time from hipEvent pairs, one step in flight; outside the frames/s timed region."""
    from ssvio_amd import synth
    depth = np.stack([synth.depth_image(300 + (f % 8), H, W) for f in range(B)])
    d_depth = torch.from_numpy(np.ascontiguousarray(depth)).to(dev)
    cam = np.array([W / 2 + 0.59, H / 2 + 0.48, 0.79 * W, 0.79 * W, np.float32(10.0 / 65535.0)], np.float32)
    ex.extract_batch_dev(d_frames, B, W, H, wait=True)
    d_kp, d_desc, d_n, cap = ex.device_results()
    fm.set_profiling(True)
    for _ in range(iters):
        dm, dn = fm.match_dbow_batch_dev(d_desc, cap, cap, d_n, B, pairs)
        fm.assemble_batch_dev(dm, dn, cap, pairs, d_kp, cap, d_n, d_depth, H * W, W, H, W, cam)
    fm.sync(); ex.sync()
    t = fm.kernel_times()
    fm.set_profiling(False)
    _, nm = fm.fetch_batch(counts_only=True)
    bgr = synth.to_rgb(synth.texture(5, H, W), 1)
    fm.set_profiling(True)
    for _ in range(iters):
        fm.RGBCloud(depth[0], bgr, cam, 10.0)
    t.update(fm.kernel_times())
    fm.set_profiling(False)
    # Map::MapPointCloudFusion (f4, second half): 8 key-frame clouds of W*H points fused one after the other into a map resident in HBM
    clouds = [torch.from_numpy(fm.RGBCloud(depth[k], bgr, cam, 10.0).view(np.uint8)).to(dev) for k in range(8)]
    Ts = [np.eye(4, dtype=np.float32) for _ in range(8)]
    for k, T in enumerate(Ts):
        T[:3, 3] = (0.05 * k, 0.01 * k, 0.02 * k)
    fm.set_profiling(True)
    wall = []
    for _ in range(3):
        fm.MapSet(None)
        for c, T in zip(clouds, Ts):
            t0 = time.perf_counter(); fm.MapFuse(c, T, 0.02); wall.append(time.perf_counter() - t0)
    tv = fm.kernel_times()
    fm.set_profiling(False)
    map_points = len(fm.MapFetch()); fm.MapSet(None)
    vox = {k: round(tv[k][0] * 1e3 / max(tv[k][1], 1), 2) for k in ("k_vox_bounds", "k_vox_keys", "vox_radix_sort", "k_vox_segments", "k_vox_centroid") if k in tv}
    us = lambda k: round(t[k][0] * 1e3 / max(t[k][1], 1), 2) if k in t else None
    return {"workload": "%d frames of %dx%d: MatchByDBoW frame f vs f-1 (vocabulary k=10, L=6, 1111111 nodes, levelsup 4), correspondence "
                        "assembly on its matches, one %dx%d cloud" % (B, W, H, W, H),
            "avg_us_per_launch": {k: us(k) for k in ("k_bow_transform", "k_bow_sort", "k_match_sorted", "k_assemble", "k_rgb_cloud")},
            "launch_covers": {"k_bow_transform": "%d descriptor sets" % B, "k_bow_sort": "%d descriptor sets" % B,
                              "k_match_sorted": "%d frame pairs (2 kernels: match + ordered compaction)" % len(pairs),
                              "k_assemble": "%d frame pairs" % len(pairs), "k_rgb_cloud": "one frame, %d points" % (W * H)},
            "dbow_matches_per_frame": round(float(nm.mean()), 1),
            "map_fusion": {"workload": "Map::MapPointCloudFusion, 8 key-frame clouds of %d points each fused in turn into the device-resident map, "
                                       "leaf 0.02 (device clouds, ssvio_map_fuse_dev)" % (W * H),
                           "avg_us_per_launch": vox, "ms_per_key_frame_wall": round(float(np.median(wall)) * 1e3, 3), "map_points_after": map_points,
                           "note": "k_vox_bounds is the first launch of a call on an idle stream: its hipEvent pair also spans the queue's wake-up "
                                   "(rocprofv3 --kernel-trace: 11 us, profiles/r2/map_probe_kernel_stats.csv)"}}


def trace(msg):
    if os.environ.get("SSVIO_BENCH_TRACE"):
        print("[bench rank %s] %s" % (os.environ.get("RANK", "0"), msg), file=sys.stderr, flush=True)


def main():
    if os.environ.get("SSVIO_BENCH_TRACE"):
        import faulthandler
        faulthandler.dump_traceback_later(100, exit=False)      # a hang shows where
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=DEFAULT_BATCH, help="frames per step (round 3, 256 / 384 / 512 / 768 / 1024 with four steps in flight: 299 / 302 / 298 / 306-314 / 309 k frames/s "
                    "over a 20-step region, 308 / 304 / 300 / 311 / 311 k sustained, tools/probe/k20_grid.sh: the region's fill and drain weigh less the longer a step is, and so do "
                    "the ramps and tails of every launch; round 2, 96 / 128 / 160 / 192 / 256: 247 / 251 / 257 / 261 / 264 k)")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--features", type=int, default=1000)
    ap.add_argument("--match-mode", choices=["bf", "ratio", "crosscheck", "dbow"], default="bf",
                    help="bf = MatchByBruteForce (config 2); ratio = kNN2 + Threshold/NNRatio (config 3); "
                         "dbow = MatchByDBoW on a synthetic k=10, L=6 vocabulary (what Odometry calls, src/Odometry.cpp:169)")
    ap.add_argument("--no-next-rows", action="store_true", help="skip the SURVEY 8-f section (MatchByDBoW, correspondence assembly, cloud)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batched-matching", action="store_true", help="skip the configs[3]/[4] matching section")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip configs[0], configs[1] streamed, configs[2] and the sustained run (N = 1 only anyway)")
    ap.add_argument("--sync-steps", action="store_true", help="host waits for every step (latency mode)")
    ap.add_argument("--streams", type=int, default=4, help="steps in flight: step i runs on handle pair i %% streams, each pair on its own HIP stream "
                    "(round 3, final tree, 4 / 6 / 8 in flight: 301-303 / 298-301 / 298 k frames/s over 20 steps, 310 / 308 / 307 k sustained; with the matrix-core matcher "
                    "335 / 340 / 336 k sustained, tools/probe/streams_sweep.sh; before k_describe's workgroups shrank to 256 threads six in flight was ahead, 296-299 against 293.5 k)")
    ap.add_argument("--no-kernel-events", action="store_true", help="no hipEvent pairs inside the timed region (the roofline entry then uses "
                    "the exclusive pass); lets the library replay its captured hipGraph, which it does not do while kernels are bracketed")
    ap.add_argument("--all-kernel-events", action="store_true", help="hipEvent pairs around every kernel in the timed region (slower)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 or os.environ.get("SSVIO_BENCH_FORCE_DIST"):     # FORCE_DIST: exercise the RCCL path with one rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # rehearsal knobs for a one-GPU box: SSVIO_BENCH_BACKEND=gloo SSVIO_BENCH_ONE_DEVICE=1 put every rank on cuda:0
        if os.environ.get("SSVIO_BENCH_ONE_DEVICE"):
            local_rank = 0
        torch.cuda.set_device(local_rank)
        backend = os.environ.get("SSVIO_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    else:
        dist = None
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)

    trace('process group up')
    import ssvio_amd
    from ssvio_amd import synth, _lib as L

    B, W, H = args.batch, args.width, args.height
    frames = synth.frame_stream(1000 + rank, B, H, W)            # seeded, per-rank stream
    d_frames = torch.from_numpy(frames).to(dev)
    # S handle pairs, each pair (extractor + matcher) on its own HIP stream; step i runs on pair i % S, so that S steps
    # are in flight and one step's low-occupancy kernels (quadtree, small pyramid levels) overlap another's
    S = max(1, args.streams)
    exs = [ssvio_amd.ORBextractor(args.features, 1.2, 8, 20, 7, device=dev.index) for _ in range(S)]
    fms = [ssvio_amd.FeatureMatching(None, {"Threshold": 50, "NNRatio": 0.6}, device=dev.index) for _ in range(S)]
    ex, fm = exs[0], fms[0]
    pairs = np.array([[f, (f - 1) % B] for f in range(B)], np.int32)
    # pipelined form (SURVEY.md §8d config 2 "sustained fps (pipelined)"): extractor and matcher of a pair enqueue on ONE
    # stream and the host does not wait inside a step; --sync-steps restores one host wait per step
    for e, m in zip(exs, fms):
        m.set_stream(e.get_stream())
    mode = {"bf": L.MODE_BRUTEFORCE, "ratio": L.MODE_KNN2_RATIO, "crosscheck": L.MODE_CROSSCHECK, "dbow": -1}[args.match_mode]
    voc = None
    if args.match_mode == "dbow" or (rank == 0 and not args.no_next_rows):
        par, vdesc, vw = synth.vocabulary_tables_full(7, 10, 6)           # ORBvoc's shape: k = 10, L = 6, 1 111 111 nodes
        voc = ssvio_amd.Vocabulary(10, 6, par, vdesc, vw)
        for m in fms:
            m.mVoc_ = voc
    nstep = [0]
    inflight = [S]

    def step():
        e, m = exs[nstep[0] % inflight[0]], fms[nstep[0] % inflight[0]]
        nstep[0] += 1
        e.extract_batch_dev(d_frames, B, W, H, wait=args.sync_steps)
        d_kp, d_desc, d_n, cap = e.device_results()
        if mode < 0:
            m.match_dbow_batch_dev(d_desc, cap, cap, d_n, B, pairs)
        else:
            m.match_batch_dev(d_desc, cap, cap, d_n, pairs, mode)
        if args.sync_steps:
            m.sync()

    def sync_all():
        for e, m in zip(exs, fms):
            m.sync(); e.sync()

    def profiling(on, only_ex=None, only_fm=None):
        for e, m in zip(exs, fms):
            e.profile_only(only_ex); m.profile_only(only_fm)
            e.set_profiling(on and (only_fm is None)); m.set_profiling(on and (only_ex is None))

    def kernel_times():
        tot = {}
        for h in exs + fms:
            for k, (ms, n) in h.kernel_times().items():
                a, c = tot.get(k, (0.0, 0))
                tot[k] = (a + ms, c + n)
        return tot

    trace('handles created')
    for _ in range(max(args.warmup, S)):        # every pair allocates on its first step
        step()
    sync_all()
    # five more untimed steps, one at a time, with event pairs around every kernel: finds the kernel with the largest
    # share of GPU time
    profiling(True)
    inflight[0] = 1
    for _ in range(5):
        step()
    sync_all()
    inflight[0] = S
    wt = kernel_times()
    dom = max(wt, key=lambda k: wt[k][0] / max(wt[k][1], 1))    # the dominant kernel: the longest launch of a step
    dom_is_ex = dom in exs[0].kernel_times()
    profiling(False)
    # timed region: only the dominant kernel is bracketed (two events per step); pairs around all ~18 launches
    # cost ~0.1 ms per step on this stream, so the full table comes from a second pass of the same steps
    if args.no_kernel_events:
        profiling(False)
    elif args.all_kernel_events:
        profiling(True)
    else:
        profiling(True, only_ex=dom if dom_is_ex else None, only_fm=None if dom_is_ex else dom)
    torch.cuda.synchronize()
    trace('before timed region')
    if dist: dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()                           # the library's streams are not torch's: wait for them, then for the device
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0        # this rank's K steps, from the common start to its own last kernel; the job's time is the MAX below
    if dist: dist.barrier()              # (the closing barrier's own launch + wait is not part of any rank's steps)
    if dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        nfr = torch.tensor([B * args.steps], dtype=torch.int64, device=dev)
        dist.all_reduce(nfr)                                        # units all ranks processed
        total_frames = int(nfr.item())
    else:
        total_frames = B * args.steps

    trace('timed region done, dt %.4f' % dt)
    # per-kernel GPU time over the timed region
    times = kernel_times()
    dom_ms, dom_launches = times.get(dom, (0.0, 0))
    # sustained: the same steps for >= 1 s of wall clock, right behind the counted ones (the counted region of a default run is a
    # few ms: this is the figure that says the rate holds)
    sustained = None
    if world == 1 and not args.no_extra_configs:
        profiling(False)
        nsus = max(args.steps, int(1.2 / max(dt / args.steps, 1e-6)) + 1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(nsus):
            step()
        sync_all(); torch.cuda.synchronize()
        dts = time.perf_counter() - t1
        sustained = {"value": round(B * nsus / dts, 2), "unit": "frames/s", "steps": nsus, "seconds": round(dts, 3)}
    # the same timed region with the matrix-core matcher selected (SSVIO_KERNEL_MFMA): identical outputs, reported BESIDE `value`,
    # which stays the library's default kernel (k_hamming_knn2, the north-star kernel)
    mfma_line = None
    if mode >= 0:
        profiling(False)
        for m in fms: m.set_kernel("mfma")
        for _ in range(S): step()
        sync_all(); torch.cuda.synchronize()
        if dist: dist.barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps): step()
        sync_all(); torch.cuda.synchronize()
        dtm = time.perf_counter() - t1
        if dist:
            t = torch.tensor([dtm], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dtm = float(t.item())
        _, nmatch_m = fm.fetch_batch(counts_only=True)
        mfma_line = {"value": round(total_frames / dtm, 2), "unit": "frames/s", "ms_per_step": round(dtm / args.steps * 1e3, 4), "steps": args.steps,
                     "matches_per_frame": round(float(nmatch_m.mean()), 1),
                     "what": "the same steps with ssvio_matcher_set_kernel(SSVIO_KERNEL_MFMA): k_hamming_knn2_mfma instead of k_hamming_knn2, bit-identical matches"}
        if world == 1 and not args.no_extra_configs:
            nsus = max(args.steps, int(1.2 / max(dtm / args.steps, 1e-6)) + 1)
            t1 = time.perf_counter()
            for _ in range(nsus): step()
            sync_all(); torch.cuda.synchronize()
            mfma_line["sustained"] = round(B * nsus / (time.perf_counter() - t1), 2)
        profiling(True, only_fm="k_hamming_knn2")
        inflight[0] = 1
        for _ in range(5): step()
        sync_all()
        inflight[0] = S
        tm = kernel_times().get("k_hamming_knn2")
        if tm and tm[1]:
            mfma_line["k_hamming_knn2_mfma_avg_us"] = round(tm[0] * 1e3 / tm[1], 2)
        profiling(False)
        for m in fms: m.set_kernel("valu")
    # the same steps again, one in flight, with event pairs around every kernel: the per-kernel table (exclusive times)
    profiling(False); profiling(True)
    inflight[0] = 1
    for _ in range(args.steps):
        step()
    sync_all()
    times = kernel_times()
    profiling(False)
    counts = ex.last_counts(B)
    _, nmatch = fm.fetch_batch(counts_only=True)
    px = level_pixels(ex, W, H)
    pairs_rows = [(int(counts[a]), int(counts[b])) for a, b in pairs]
    kern = {}
    for name, (ms, launches) in times.items():
        ab = algorithmic_bytes(name, B, px, int(counts.sum()), pairs_rows)
        avg_us = ms * 1e3 / max(launches, 1)
        kern[name] = {"total_ms": round(ms, 3), "launches": launches, "avg_us": round(avg_us, 2),
                      "alg_bytes_per_launch": int(ab), "GBps": round(ab / (avg_us * 1e-6) / 1e9, 1) if avg_us > 0 else None}
    if dom_launches == 0:                      # --no-kernel-events: the exclusive pass is the only measurement
        dom_ms, dom_launches = times[dom]
    prof_ok = profiles_current() and (B, W, H, args.features) == (DEFAULT_BATCH, 640, 480, 1000)     # the committed PMC figures are per launch of the default command
    dom_avg_s = dom_ms * 1e-3 / dom_launches
    dom_bytes = algorithmic_bytes(dom, B, px, int(counts.sum()), pairs_rows)
    excl_s = times[dom][0] * 1e-3 / times[dom][1]
    achieved = dom_bytes / excl_s / 1e9
    busy, cyc = valu_busy(dom, excl_s * 1e6) if prof_ok else (None, None)
    alg_step = B * (W * H + 3 * sum(px)) + int(counts.sum()) * (1112 + 60)          # W*H + 3 P + N*(1112 + 60) per frame
    roof = {"kernel": dom, "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5),
            "frac_note": "algorithmic bytes of one launch / the kernel's EXCLUSIVE launch time (hipEvent pair on the library's stream, one step in flight; "
                         "rocprof_avg_launch_us is rocprofv3's figure for the same command) / 8 TB/s",
            "traffic": pmc_traffic(dom) if prof_ok else None,
            "traffic_source": "profiles/r3/r3_final_pmc_hbm.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, default batch); null when the "
                              "kernel sources have changed since that profile (profiles/r3/r3_final_source_hash.txt)",
            "avg_launch_us": round(excl_s * 1e6, 2), "alg_bytes_per_launch": int(dom_bytes),
            "rocprof_avg_launch_us": rocprof_avg_us(dom) if prof_ok else None,
            "in_flight_avg_launch_us": round(dom_avg_s * 1e6, 2),
            "in_flight_frac": round(dom_bytes / dom_avg_s / 1e9 / HBM_PEAK_GBS, 5),
            "in_flight_note": "hipEvent pair around the kernel inside the timed region: with several steps in flight it includes the wait for CUs "
                              "held by the other steps' kernels",
            "valu_busy": busy, "valu_issue_cycles_per_instruction": cyc,
            "valu_busy_note": "SQ_INSTS_VALU x the kernel's average issue cycles (two measured issue classes weighted by its static instruction mix, "
                              "profiles/r3/valu_mix.json) over the SIMD cycles of the exclusive launch at 2.4 GHz",
            "pipeline_frac": round(alg_step / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 5),
            "pipeline_note": "algorithmic bytes of a whole step (%d frames x %.2f MB) / ms_per_step / 8 TB/s" % (B, alg_step / B / 1e6),
            "binds": "VALU issue, not HBM: the path is integer byte work with ~45 lane-instructions per pyramid pixel (DESIGN.md section 3); the HBM fraction is reported because the contract asks for it"}
    # the matcher is VALU-bound (xor+popcount), not HBM-bound: report its VALU fraction beside it
    if "k_hamming_knn2" in times:
        ms, launches = times["k_hamming_knn2"]
        pairs_eval = sum(nq * nt for nq, nt in pairs_rows)
        rate = pairs_eval / (ms * 1e-3 / launches)
        nn = args.match_mode != "ratio"              # the nearest-neighbour form of the kernel (no second key)
        valu = {"kernel": "k_hamming_knn2", "descriptor_pairs_per_s": rate, "valu_instructions_per_pair": 18 if nn else 19,
                "issue_cycles_per_pair": round(MATCH_CYCLES_PER_PAIR_NN if nn else MATCH_CYCLES_PER_PAIR, 2), "peak_pairs_per_s": VALU_PEAK_PAIRS_NN if nn else VALU_PEAK_PAIRS,
                "frac": round(rate / (VALU_PEAK_PAIRS_NN if nn else VALU_PEAK_PAIRS), 4), "timing": "exclusive pass"}
    else:
        valu = None

    trace('kernel table done')
    nxt = None if (args.no_next_rows or rank != 0) else next_rows(exs[0], fms[0], d_frames, B, W, H, pairs, dev)   # no collective inside: rank 0 only
    bm = None if args.no_batched_matching else batched_matching(rank, world, dev, dist)
    xc = extra_configs(dev, frames, not args.no_cpu_baseline) if (world == 1 and not args.no_extra_configs and (W, H) == (640, 480)) else None
    if rank == 0:
        out = {
            "metric": "ORB extract + match frames/s (%s)" % ("BASELINE configs[1], VGA, device-resident" if (W, H, args.features, args.match_mode) == (640, 480, 1000, "bf")
                                                             else "BASELINE configs[2], 1080p, device-resident" if (W, H, args.features, args.match_mode) == (1920, 1080, 4000, "ratio")
                                                             else "non-BASELINE workload, see config.workload"),
            "value": round(total_frames / dt, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%s: %dx%d stream, %d features/frame, 8 levels, extract + %s vs previous frame"
                                   % ("configs[1]" if (W, H, args.features, args.match_mode) == (640, 480, 1000, "bf") else "configs[2]" if (W, H, args.features, args.match_mode) == (1920, 1080, 4000, "ratio") else "custom",
                                      W, H, args.features, {"bf": "MatchByBruteForce", "ratio": "kNN2 ratio test", "crosscheck": "cross-check", "dbow": "MatchByDBoW (synthetic k=10 L=6 vocabulary)"}[args.match_mode]),
                       "frames_per_step": B, "parallelism": "frames sharded over %d rank(s), replicas, no collective on the data path" % world,
                       "steps_in_flight": S, "residency": "device-resident: the %d frames of a step are in HBM before the timed region (configs[].residency for the streamed run)" % B},
            "sustained": sustained,
            "configs": xc,
            "keypoints_per_frame": round(float(counts.mean()), 1),
            "matches_per_frame": round(float(nmatch.mean()), 1),
            "query_matches_per_s": round(float(nmatch.sum()) * args.steps * world / dt, 1),
            "roofline": roof, "valu": valu, "kernels": kern,
            "kernel_events": {"roofline": "hipEvent pairs around %s in the timed region" % dom,
                              "kernels": "second pass of the same steps, one step in flight, pairs around every kernel (exclusive times)"},
            "host_waits": "per step" if args.sync_steps else "none inside the timed region (pipelined)",
            "mfma_matcher": mfma_line,
            "batched_matching": bm,
            "next_rows": nxt,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(frames, args.features, args.match_mode)
            out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(frames, args.features, args.match_mode)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    trace("closing handles")
    for h in fms + exs:                                   # the matchers run on their extractors' streams: they go first
        h.close()
    trace("handles closed")
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    trace("main done")


if __name__ == "__main__":
    main()
    # every handle is closed and the line is out: the process ends through the interpreter's and the HIP runtime's normal teardown,
    # so that a fault or hang in it shows in the exit status.  (The hang once seen at the end of one run in four was the bench's own
    # close order, fixed in round 2.)  SSVIO_BENCH_HARD_EXIT=1 skips the teardown.
    sys.stdout.flush(); sys.stderr.flush()
    if os.environ.get("SSVIO_BENCH_HARD_EXIT"):
        os._exit(0)
