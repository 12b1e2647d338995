"""GPU parity of the second 2-NN kernel, k_hamming_knn2_mfma (the Hamming distance as an exact int8 contraction on the
matrix cores), through the C ABI: against the CPU oracle in every mode, and against k_hamming_knn2 bit for bit on
batches -- ties, distance 0 and 256, ragged and empty sets, chunk and tile boundaries (32-row tiles, 128-row stages,
chunks of up to 8160 rows)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    import torch
    import ssvio_amd
    from ssvio_amd import synth, _lib as L
    from oracle import oracle as O
    return torch, ssvio_amd, synth, L, O


@pytest.fixture(scope="module")
def fm(mods):
    S = mods[1]
    m = S.FeatureMatching(None, {"Threshold": 50, "NNRatio": 0.6})
    m.set_kernel("mfma_all")
    assert m.get_kernel() == "mfma_all"
    yield m
    m.close()


def test_kernel_selection_is_validated(mods):
    torch, S, synth, L, O = mods
    m = S.FeatureMatching(None, {})
    assert m.get_kernel() == "valu"                       # the default stays the xor / popcount kernel
    with pytest.raises(S.SsvioError):
        m.set_kernel(7)
    m.set_kernel("mfma"); assert m.get_kernel() == "mfma"
    m.close()


def _all_modes(fm, O, q, t):
    idx, d1, d2 = fm.knn2_raw(q, t)
    ri, r1, r2 = O.knn2_raw(q, t)
    assert np.array_equal(d1, r1) and np.array_equal(d2, r2) and np.array_equal(idx, ri)
    ok, m = fm.MatchByBruteForce(q, t)
    assert np.array_equal(m.view(np.uint8), O.match_bruteforce(q, t).view(np.uint8)) and ok == (len(m) > 0)
    ok, m = fm.MatchByKnnRatio(q, t)
    assert np.array_equal(m.view(np.uint8), O.match_knn2_ratio(q, t, 50, 0.6).view(np.uint8))
    ok, m = fm.MatchByCrossCheck(q, t)
    assert np.array_equal(m.view(np.uint8), O.match_crosscheck(q, t).view(np.uint8))
    ok, m = fm.MatchByCrossCheckCV(q, t)
    assert np.array_equal(m.view(np.uint8), O.match_crosscheck_cv(q, t).view(np.uint8))


@pytest.mark.parametrize("nq,nt", [(1, 1), (1, 300), (300, 1), (31, 33), (32, 32), (255, 257), (129, 128), (500, 500), (1000, 1010),
                                   (385, 4097), (4000, 4000), (2049, 777), (777, 8161), (9000, 8200)])
def test_all_modes_match_oracle(mods, fm, nq, nt):
    torch, S, synth, L, O = mods
    q, t = synth.descriptor_sets(nq * 31 + nt, nq, nt)
    _all_modes(fm, O, q, t)


def test_ties_extremes_and_row_field_boundaries(mods, fm):
    """lowest-index ties across tiles, stages and lane halves; distances 0 and 256; all-zero / all-one rows (|a| = 0 / 256)"""
    torch, S, synth, L, O = mods
    q, t = synth.descriptor_sets(5, 64, 40)
    t = np.concatenate([t] * 9 + [t[:7]])                # every train row 9-10 times: 367 rows, ties in every tile
    q[:10] = t[5:15]
    q[10] = 0; q[11] = 255                              # |a| = 0 and 256
    t[100] = 0; t[101] = 255; t[366] = 255
    _all_modes(fm, O, q, t)
    idx, d1, d2 = fm.knn2_raw(q, t)
    assert (d1[:10] == 0).all() and (idx[:10] == np.arange(5, 15)).all() and d1[10] == 0 and idx[10] == 100 and idx[11] == 101
    q = np.zeros((4, 32), np.uint8); t = np.full((3, 32), 255, np.uint8)     # every pair at distance 256
    _all_modes(fm, O, q, t)
    q = np.full((70, 32), 255, np.uint8); t = np.zeros((8161, 32), np.uint8); t[8160] = 255      # the match is the last row of a second chunk
    idx, d1, d2 = fm.knn2_raw(q, t)
    assert (idx == 8160).all() and (d1 == 0).all() and (d2 == 256).all()
    _all_modes(fm, O, q, t)


def _sets(synth, seed, counts, stride):
    rng = np.random.default_rng(seed)
    base = synth.descriptor_sets(seed, 8, max(counts) + 8)[1]
    buf = np.zeros((len(counts), stride, 32), np.uint8)
    for s, c in enumerate(counts):
        buf[s, :c] = base[rng.permutation(len(base))[:c]] ^ np.packbits(rng.random((c, 256)) < 0.05, axis=1, bitorder="little")
    return buf


@pytest.mark.parametrize("counts,stride,max_rows,npairs,seed", [
    ([1300, 1024, 0, 700, 1299, 65, 1, 911, 1300, 512], 1344, 1300, 40, 11),
    ([300, 17, 256, 0, 299, 64], 320, 300, 9, 12),
    ([4000, 3999, 4000, 2500, 33], 4000, 4000, 12, 13),
    ([8000, 8000, 7777], 8000, 8000, 4, 21)])
def test_batches_equal_the_valu_kernel_and_the_oracle(mods, counts, stride, max_rows, npairs, seed):
    torch, S, synth, L, O = mods
    fm = S.FeatureMatching(None, {"Threshold": 50, "NNRatio": 0.6})
    buf = _sets(synth, seed, counts, stride)
    d_desc = torch.from_numpy(buf).cuda(); d_cnt = torch.tensor(counts, dtype=torch.int32).cuda()
    for mode, ref in ((L.MODE_BRUTEFORCE, O.match_bruteforce), (L.MODE_KNN2_RATIO, lambda a, c: O.match_knn2_ratio(a, c, 50, 0.6)),
                      (L.MODE_CROSSCHECK, O.match_crosscheck), (L.MODE_CROSSCHECK_CV, O.match_crosscheck_cv)):
        rng = np.random.default_rng(seed + mode)
        pairs = np.stack([rng.integers(0, len(counts), npairs), rng.integers(0, len(counts), npairs)], 1).astype(np.int32)
        res = {}
        for kernel in ("valu", "mfma"):
            fm.set_kernel(kernel)
            fm.match_batch_dev(d_desc, stride, max_rows, d_cnt, pairs, mode)
            fm.sync()
            res[kernel] = fm.fetch_batch()
        (want, want_n), (got, got_n) = res["valu"], res["mfma"]
        assert np.array_equal(got_n, want_n), (mode, seed)
        for p in range(npairs):
            assert np.array_equal(got[p, :got_n[p]].view(np.uint8), want[p, :want_n[p]].view(np.uint8)), (mode, seed, p)
        for p in (0, npairs - 1):
            a, c = pairs[p]
            w = ref(buf[a, :counts[a]], buf[c, :counts[c]])
            assert got_n[p] == len(w) and np.array_equal(got[p, :got_n[p]].view(np.uint8), w.view(np.uint8))
    # the entry point that allocates nothing runs the selected kernel as well
    fm.set_kernel("mfma")
    pairs = np.stack([np.arange(len(counts)), np.arange(len(counts))[::-1]], 1).astype(np.int32)
    fm.match_batch_dev(d_desc, stride, max_rows, d_cnt, pairs, L.MODE_BRUTEFORCE); fm.sync()
    want, want_n = fm.fetch_batch()
    d_pairs = torch.from_numpy(pairs).cuda()
    d_out = torch.full((len(pairs), max_rows, 4), -7, dtype=torch.int32, device="cuda"); d_n = torch.full((len(pairs),), -7, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    fm.match_batch_dev_noalloc(d_desc, stride, max_rows, d_cnt, d_pairs, len(pairs), L.MODE_BRUTEFORCE, d_out, d_n); fm.sync()
    got = d_out.cpu().numpy().view(L.DM_DTYPE).reshape(len(pairs), max_rows); got_n = d_n.cpu().numpy()
    assert np.array_equal(got_n, want_n) and all(np.array_equal(got[p, :got_n[p]].view(np.uint8), want[p, :want_n[p]].view(np.uint8)) for p in range(len(pairs)))
    fm.close()
