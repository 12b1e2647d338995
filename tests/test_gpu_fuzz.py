"""GPU parity on randomly drawn configurations (fixed seeds): image sizes that put tile, strip and cell borders in
odd places, scale factors on both sides of the streaming resize kernel's range, pyramid depths, quotas and FAST
thresholds.  Bit-exact end to end against the oracle, like test_gpu_extract.py; configurations whose upper levels
are too small for a 30-px cell must be refused on both sides."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        w = int(rng.integers(97, 900)); h = int(rng.integers(97, 700))
        sf = float(np.float32(rng.choice([1.1, 1.2, 1.2, 1.25, 1.33, 1.5, 1.7, 2.0, 2.3])))
        nl = int(rng.integers(1, 10))
        nf = int(rng.choice([50, 300, 1000, 1200, 2500]))
        ini = int(rng.choice([20, 30, 12])); mn = int(rng.choice([7, 5, 10]))
        out.append((w, h, sf, nl, nf, ini, min(mn, ini), int(rng.integers(0, 1 << 30))))
    return out


@pytest.mark.parametrize("w,h,sf,nl,nf,ini,mn,seed", _cases(24, 20261110) + [(880, 130, 1.2, 3, 600, 20, 7, 5), (900, 100, 1.2, 2, 300, 20, 7, 6),
                                                          (130, 420, 1.2, 2, 300, 20, 7, 7)])
def test_random_configuration_equals_oracle(w, h, sf, nl, nf, ini, mn, seed):
    import ssvio_amd as S
    from ssvio_amd import synth
    from oracle import oracle as O
    kind = seed % 3
    img = synth.texture(seed, h, w) if kind else (np.random.default_rng(seed).integers(0, 256, (h, w)).astype(np.uint8))
    if kind == 2:
        img = (img // 3 + 80).astype(np.uint8)                      # low contrast: the minTh retry fires in many cells
    g = S.ORBextractor(nf, sf, nl, ini, mn); c = O.Extractor(nf, sf, nl, ini, mn)
    try:
        want = c.extract(img)
    except ValueError:
        with pytest.raises(S.SsvioError):
            g.Extract(img)
        g.close()
        return
    if round((w - 32) / (h - 32)) > 8:                               # documented limit: at most 8 quadtree roots per level
        with pytest.raises(S.SsvioError, match="quadtree roots"):
            g.Extract(img)
        g.close()
        return
    kps, desc = g.Extract(img)
    rk, rd = want
    assert len(kps) == len(rk)
    assert np.array_equal(kps.view(np.uint8), rk.view(np.uint8)), "keypoints"
    assert np.array_equal(desc, rd), "descriptors"
    # twice on the same handle, then as a batch of 3 with a different frame between: same answer (no state carried over)
    other = np.ascontiguousarray(img[::-1])
    ks, ds, ns = g.extract_batch(np.stack([img, other, img]))
    for f in (0, 2):
        assert ns[f] == len(rk) and np.array_equal(ks[f, :ns[f]].view(np.uint8), rk.view(np.uint8)) and np.array_equal(ds[f, :ns[f]], rd)
    g.close()
