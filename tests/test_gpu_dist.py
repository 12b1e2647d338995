"""GPU: the batched-matching drivers of ssvio_amd.dist on one MI355X (world 1, no process group) -- the code an 8-GPU run
executes per rank -- against the oracle, pair by pair: BASELINE config 5 as a reduced all-pairs set (loop_candidates: the
super-block walk, device-side summary records) and config 4 at full size (window_matches: 1 query frame x 64 keyframes x
4000 descriptors, cross-check).  Integer work: bit-exact counts, distance sums and DMatch rows."""
import concurrent.futures as cf
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    import torch
    import ssvio_amd
    from ssvio_amd import synth, dist as D, _lib as L
    from oracle import oracle as O
    return torch, ssvio_amd, synth, D, L, O


def _oracle_all(fn, jobs):
    with cf.ThreadPoolExecutor(16) as pool:                      # ctypes releases the GIL inside the oracle's C calls
        return list(pool.map(lambda ab: fn(*ab), jobs))


@pytest.mark.parametrize("sb", [4, 5, 32])
def test_loop_candidates_reduced_all_pairs(mods, sb):
    torch, S, synth, D, L, O = mods
    rng = np.random.default_rng(7)
    nsets, stride = 12, 1344
    counts = [int(c) for c in rng.integers(700, 1301, nsets)]
    counts[4] = 0; counts[9] = 1300
    base = synth.descriptor_sets(55, 8, 1300)[1]
    sets = [np.ascontiguousarray(base[rng.permutation(1300)[:c]] ^ np.packbits(rng.random((c, 256)) < 0.05, axis=1, bitorder="little")) for c in counts]
    buf = np.zeros((nsets, stride, 32), np.uint8)
    for s, d in enumerate(sets):
        buf[s, :len(d)] = d
    fm = S.FeatureMatching(None, {"Threshold": 50, "NNRatio": 0.6})
    blocks = D.DeviceBlocks(fm, torch.from_numpy(buf).cuda(), stride, 1300, torch.tensor(counts, dtype=torch.int32).cuda(), "cuda:0")
    work = {}
    for _ in range(2):                                           # second pass reuses the record buffer
        rec = D.records_to_numpy(D.loop_candidates(blocks, nsets, 0, 1, sb=sb, work=work))
    pairs = D.all_pairs(nsets)
    want = _oracle_all(O.match_bruteforce, [(sets[a], sets[b]) for a, b in pairs])
    assert rec["pair"].tolist() == list(range(len(pairs)))
    assert rec["n_matches"].tolist() == [len(w) for w in want]
    assert rec["sum_dist"].tolist() == [int(w["distance"].sum()) for w in want]
    # the same walk with full lists: every pair-block's DMatch rows (sequential ids through window_matches)
    lists, cnt = D.lists_to_numpy(*D.window_matches(blocks, pairs, 0, 1, L.MODE_BRUTEFORCE))
    for k, w in enumerate(want):
        assert cnt[k] == len(w) and np.array_equal(lists[k, :cnt[k]].view(np.uint8), w.view(np.uint8)), k
    fm.close()


def test_window_matches_config4_full_size(mods):
    """BASELINE configs[3]: 1 query frame x 64 keyframes x 4000 descriptors, cross-check -- every list vs the oracle"""
    torch, S, synth, D, L, O = mods
    rows, nkf = 4000, 64
    q, base = synth.descriptor_sets(400, rows * nkf, rows)
    sets = [base] + [q[k * rows:(k + 1) * rows] for k in range(nkf)]          # set 0 = query frame, sets 1..64 = keyframes
    d = torch.from_numpy(np.concatenate(sets)).cuda()
    cnt = torch.full((nkf + 1,), rows, dtype=torch.int32).cuda()
    fm = S.FeatureMatching(None, {"Threshold": 50, "NNRatio": 0.6})
    blocks = D.DeviceBlocks(fm, d, rows, rows, cnt, "cuda:0")
    pairs = np.array([[0, k + 1] for k in range(nkf)], np.int32)
    lists, n = D.lists_to_numpy(*D.window_matches(blocks, pairs, 0, 1, L.MODE_CROSSCHECK))
    want = _oracle_all(O.match_crosscheck, [(sets[a], sets[b]) for a, b in pairs])
    assert n.tolist() == [len(w) for w in want] and sum(n) > 10000
    for k, w in enumerate(want):
        assert np.array_equal(lists[k, :n[k]].view(np.uint8), w.view(np.uint8)), k
    # per-pair summaries of the same batch, sequential ids
    with blocks.stream():
        rec = torch.empty((nkf, 4), dtype=torch.int32, device="cuda")
        blocks.summaries_into(rec, L.PAIR_ID_SEQUENTIAL, 100)
    blocks.publish()
    rec = D.records_to_numpy(rec)
    assert rec["pair"].tolist() == list(range(100, 100 + nkf)) and rec["n_matches"].tolist() == n.tolist()
    assert rec["sum_dist"].tolist() == [int(w["distance"].sum()) for w in want]
    fm.close()


def test_config5_full_size_blocks_through_the_driver(mods):
    """4 keyframes x 8000 descriptors (BASELINE configs[4]'s block size), all 6 pairs through loop_candidates; the oracle checks
    two pairs in full, the rest by the records' own consistency with the full lists"""
    torch, S, synth, D, L, O = mods
    rows, nsets = 8000, 4
    q, _ = synth.descriptor_sets(500, rows * nsets, rows)
    sets = [q[k * rows:(k + 1) * rows] for k in range(nsets)]
    fm = S.FeatureMatching(None, {"Threshold": 50, "NNRatio": 0.6})
    blocks = D.DeviceBlocks(fm, torch.from_numpy(q).cuda(), rows, rows, torch.full((nsets,), rows, dtype=torch.int32).cuda(), "cuda:0")
    rec = D.records_to_numpy(D.loop_candidates(blocks, nsets, 0, 1, sb=2))
    pairs = D.all_pairs(nsets)
    lists, n = D.lists_to_numpy(*D.window_matches(blocks, pairs, 0, 1, L.MODE_BRUTEFORCE))
    assert rec["n_matches"].tolist() == n.tolist()
    assert rec["sum_dist"].tolist() == [int(lists[k, :n[k]]["distance"].sum()) for k in range(len(pairs))]
    want = _oracle_all(O.match_bruteforce, [(sets[a], sets[b]) for a, b in pairs[[0, 5]]])
    for k, w in zip((0, 5), want):
        assert n[k] == len(w) and np.array_equal(lists[k, :n[k]].view(np.uint8), w.view(np.uint8))
    fm.close()


@pytest.mark.parametrize("kernel", ["valu", "mfma"])
def test_native_walk_equals_the_python_walk(mods, kernel):
    """ssvio_loop_candidates_dev (the rank's super-blocks in one C-ABI call, pair lists built on the device) against the per-tile
    Python walk over ssvio_match_batch_dev + ssvio_match_batch_summary_dev: same records, for 1 and 3 ranks' shares and two modes"""
    torch, S, synth, D, L, O = mods
    rng = np.random.default_rng(17)
    nsets, stride = 11, 704
    counts = [int(c) for c in rng.integers(300, 701, nsets)]
    counts[2] = 0; counts[7] = 700
    base = synth.descriptor_sets(56, 8, 700)[1]
    buf = np.zeros((nsets, stride, 32), np.uint8)
    for s, c in enumerate(counts):
        buf[s, :c] = base[rng.permutation(700)[:c]] ^ np.packbits(rng.random((c, 256)) < 0.05, axis=1, bitorder="little")
    fm = S.FeatureMatching(None, {"Threshold": 50, "NNRatio": 0.6})
    fm.set_kernel(kernel)
    blocks = D.DeviceBlocks(fm, torch.from_numpy(buf).cuda(), stride, 700, torch.tensor(counts, dtype=torch.int32).cuda(), "cuda:0")

    class PythonWalk:                                             # the same blocks without the native walk
        def __getattr__(self, name):
            if name == "loop_walk":
                raise AttributeError(name)
            return getattr(blocks, name)
    npairs = nsets * (nsets - 1) // 2
    for mode in (L.MODE_BRUTEFORCE, L.MODE_KNN2_RATIO):
        for world, sb in ((1, 4), (3, 3)):
            nat = np.zeros(npairs, L.REC_DTYPE); py = np.zeros(npairs, L.REC_DTYPE)
            for rank in range(world):                              # a rank's own records are everything but the other ranks' zeros
                a = D.records_to_numpy(_one_rank(D, blocks, nsets, rank, world, sb, mode))
                b = D.records_to_numpy(_one_rank(D, PythonWalk(), nsets, rank, world, sb, mode))
                assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), (mode, world, rank)
                nat["pair"] += a["pair"]; nat["n_matches"] += a["n_matches"]; nat["sum_dist"] += a["sum_dist"]
            assert nat["pair"].tolist() == list(range(npairs))     # every pair exactly once over the ranks
    fm.close()


def _one_rank(D, blocks, nsets, rank, world, sb, mode):
    """loop_candidates for one rank's share without a process group: what the rank contributes before the all_gather"""
    import unittest.mock as mock
    import torch
    def alone(local, w, group=None):                               # the other ranks contribute padding records only
        out = torch.full((w,) + tuple(local.shape), -1, dtype=local.dtype, device=local.device)
        out[rank] = local
        return out
    with mock.patch.object(D, "gather_fixed", alone):
        return D.loop_candidates(blocks, nsets, rank, world, sb=sb, mode=mode)
