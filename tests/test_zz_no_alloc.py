"""GPU: ssvio_match_batch_dev_noalloc (the entry point that allocates nothing) against ssvio_match_batch_dev, through
the C ABI, bit for bit, on two calls in a row -- the second with a smaller input, so stale scratch contents of the
first call would show -- in every mode; plus the capacity rejection.  Collected last (file name) so that a failure here
cannot hide other tests under `pytest -x`."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sets(synth, seed, counts, stride):
    rng = np.random.default_rng(seed)
    base = synth.descriptor_sets(seed, 8, max(counts) + 8)[1]
    buf = np.zeros((len(counts), stride, 32), np.uint8)
    for s, c in enumerate(counts):
        buf[s, :c] = base[rng.permutation(len(base))[:c]] ^ np.packbits(rng.random((c, 256)) < 0.05, axis=1, bitorder="little")
    return buf


def _both(fm, torch, L, buf, counts, stride, max_rows, pairs, mode):
    """-> (reference lists, reference counts, no-alloc lists, no-alloc counts) as numpy"""
    d_desc = torch.from_numpy(buf).cuda(); d_cnt = torch.tensor(counts, dtype=torch.int32).cuda()
    fm.match_batch_dev(d_desc, stride, max_rows, d_cnt, pairs, mode)
    fm.sync()
    want, want_n = fm.fetch_batch()
    d_pairs = torch.from_numpy(pairs).cuda()
    d_out = torch.full((len(pairs), max_rows, 4), -7, dtype=torch.int32, device="cuda")
    d_n = torch.full((len(pairs),), -7, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    # the property the entry point exists for: it allocates nothing.  Every growing allocation of the library is made to fail while
    # it runs, and the device's free memory must not move across it
    free0 = torch.cuda.mem_get_info()[0]
    L.lib().ssvio_test_fail_next_allocs(1 << 20)
    try:
        fm.match_batch_dev_noalloc(d_desc, stride, max_rows, d_cnt, d_pairs, len(pairs), mode, d_out, d_n)
    finally:
        L.lib().ssvio_test_fail_next_allocs(0)
    fm.sync()
    assert torch.cuda.mem_get_info()[0] == free0, "the no-allocation entry point changed the device's free memory"
    got = d_out.cpu().numpy().view(L.DM_DTYPE).reshape(len(pairs), max_rows)
    return want, want_n, got, d_n.cpu().numpy()


def test_noalloc_equals_allocating_entry_point_on_two_calls_in_a_row():
    import torch
    import ssvio_amd as S
    from ssvio_amd import synth, _lib as L
    from oracle import oracle as O
    fm = S.FeatureMatching(None, {"Threshold": 50, "NNRatio": 0.6})
    # first call: 40 pairs of up to 1300 rows; second call: 9 pairs of up to 300 rows (smaller in pairs, rows and chunks)
    calls = [([1300, 1024, 0, 700, 1299, 65, 1, 911, 1300, 512], 1344, 1300, 40, 11),
             ([300, 17, 256, 0, 299, 64], 320, 300, 9, 12)]
    for mode, ref in ((L.MODE_BRUTEFORCE, O.match_bruteforce), (L.MODE_KNN2_RATIO, lambda a, c: O.match_knn2_ratio(a, c, 50, 0.6)),
                      (L.MODE_CROSSCHECK, O.match_crosscheck), (L.MODE_CROSSCHECK_CV, O.match_crosscheck_cv)):
        for counts, stride, max_rows, npairs, seed in calls:
            rng = np.random.default_rng(seed + mode)
            buf = _sets(synth, seed, counts, stride)
            pairs = np.stack([rng.integers(0, len(counts), npairs), rng.integers(0, len(counts), npairs)], 1).astype(np.int32)
            want, want_n, got, got_n = _both(fm, torch, L, buf, counts, stride, max_rows, pairs, mode)
            assert np.array_equal(got_n, want_n), (mode, seed)
            for p in range(npairs):
                assert np.array_equal(got[p, :got_n[p]].view(np.uint8), want[p, :want_n[p]].view(np.uint8)), (mode, seed, p)
            for p in (0, npairs - 1):                                   # and both equal the oracle
                a, c = pairs[p]
                w = ref(buf[a, :counts[a]], buf[c, :counts[c]])
                assert got_n[p] == len(w) and np.array_equal(got[p, :got_n[p]].view(np.uint8), w.view(np.uint8))
    fm.close()


def test_noalloc_large_block_and_capacity_limit():
    """one 8000 x 8000 block per pair (BASELINE config 5's block) fits; a batch beyond the fixed scratch is refused whole"""
    import torch
    import ssvio_amd as S
    from ssvio_amd import synth, _lib as L
    fm = S.FeatureMatching(None, {"Threshold": 50, "NNRatio": 0.6})
    rows = 8000
    buf = _sets(synth, 21, [rows, rows, 7777], rows)
    pairs = np.array([[0, 1], [1, 2], [2, 0], [1, 1]], np.int32)
    want, want_n, got, got_n = _both(fm, torch, L, buf, [rows, rows, 7777], rows, rows, pairs, L.MODE_BRUTEFORCE)
    assert np.array_equal(got_n, want_n) and all(np.array_equal(got[p, :got_n[p]].view(np.uint8), want[p, :want_n[p]].view(np.uint8)) for p in range(4))
    # 600 pairs x 8000 rows x 8 B = 38.4 MB > 32 MiB
    npairs = 600
    big = np.zeros((npairs, 2), np.int32)
    d_desc = torch.from_numpy(buf).cuda(); d_cnt = torch.tensor([rows, rows, 7777], dtype=torch.int32).cuda()
    d_pairs = torch.from_numpy(big).cuda()
    d_out = torch.zeros((8, 4), dtype=torch.int32, device="cuda"); d_n = torch.full((npairs,), -7, dtype=torch.int32, device="cuda")
    with pytest.raises(S.SsvioError) as e:
        fm.match_batch_dev_noalloc(d_desc, rows, rows, d_cnt, d_pairs, npairs, L.MODE_BRUTEFORCE, d_out, d_n)
    assert e.value.code == L.ERR_CAPACITY and "scratch" in str(e.value)
    torch.cuda.synchronize()
    assert (d_n.cpu().numpy() == -7).all()                               # nothing was launched
    fm.close()
