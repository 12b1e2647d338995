"""Deterministic synthetic inputs: frames with plenty of FAST
corners, frame pairs / streams, and descriptor sets for match-only runs.

No library RNG: a counter-based splitmix64 so that every seed gives the same
bytes on every machine and numpy version.  Host-side input generation only; not
part of the extract/match compute path.
"""
import numpy as np

_G = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


class Rng:
    def __init__(self, seed):
        self.seed = np.uint64(seed & 0xFFFFFFFFFFFFFFFF)
        self.ctr = 0

    def u64(self, n):
        with np.errstate(over="ignore"):
            i = np.arange(self.ctr + 1, self.ctr + n + 1, dtype=np.uint64)
            z = self.seed + i * _G
            z = (z ^ (z >> np.uint64(30))) * _M1
            z = (z ^ (z >> np.uint64(27))) * _M2
            z = z ^ (z >> np.uint64(31))
        self.ctr += n
        return z

    def uniform(self, n):
        return (self.u64(n) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))

    def randint(self, lo, hi, n):
        """integers in [lo, hi)"""
        return (lo + (self.u64(n) % np.uint64(max(hi - lo, 1))).astype(np.int64))

    def normal(self, n):
        m = (n + 1) // 2
        u1 = np.maximum(self.uniform(m), 1e-300); u2 = self.uniform(m)
        r = np.sqrt(-2.0 * np.log(u1))
        z = np.concatenate([r * np.cos(2 * np.pi * u2), r * np.sin(2 * np.pi * u2)])
        return z[:n]

    def bytes(self, n):
        m = (n + 7) // 8
        return self.u64(m).view(np.uint8)[:n].copy()


def _value_noise(rng, h, w, cell, amp):
    gh, gw = h // cell + 2, w // cell + 2
    g = rng.uniform(gh * gw).reshape(gh, gw) * 2 - 1
    ys = np.arange(h) / cell; xs = np.arange(w) / cell
    y0 = ys.astype(int); x0 = xs.astype(int)
    fy = (ys - y0)[:, None]; fx = (xs - x0)[None, :]
    a = g[y0][:, x0]; b = g[y0][:, x0 + 1]; c = g[y0 + 1][:, x0]; d = g[y0 + 1][:, x0 + 1]
    return amp * ((a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy)


def texture(seed, h, w, n_shapes=None, noise_sigma=2.0):
    """8-bit gray texture: random rectangles + discs + 4-octave value noise + pixel noise."""
    rng = Rng(seed)
    if n_shapes is None:
        n_shapes = max(200, min(4000, (h * w) // 500))
    img = np.full((h, w), 128.0)
    nr = n_shapes * 2 // 3; nd = n_shapes - nr
    rx = rng.randint(-20, w, nr); ry = rng.randint(-20, h, nr)
    rw = rng.randint(6, 90, nr); rh = rng.randint(6, 90, nr); rg = rng.randint(10, 246, nr)
    dx = rng.randint(0, w, nd); dy = rng.randint(0, h, nd); dr = rng.randint(4, 40, nd); dg = rng.randint(10, 246, nd)
    order = rng.randint(0, 2, n_shapes)
    ir = idd = 0
    yy, xx = np.mgrid[0:h, 0:w]
    for k in range(n_shapes):
        if (order[k] == 0 and ir < nr) or idd >= nd:
            x0, y0 = max(rx[ir], 0), max(ry[ir], 0)
            img[y0:max(ry[ir] + rh[ir], 0), x0:max(rx[ir] + rw[ir], 0)] = rg[ir]; ir += 1
        else:
            r = dr[idd]; cx, cy = dx[idd], dy[idd]
            ys0, ys1 = max(cy - r, 0), min(cy + r + 1, h); xs0, xs1 = max(cx - r, 0), min(cx + r + 1, w)
            sub = (yy[ys0:ys1, xs0:xs1] - cy) ** 2 + (xx[ys0:ys1, xs0:xs1] - cx) ** 2 <= r * r
            img[ys0:ys1, xs0:xs1][sub] = dg[idd]; idd += 1
    for cell, amp in ((64, 16.0), (32, 8.0), (16, 4.0), (8, 2.0)):
        img += _value_noise(rng, h, w, cell, amp)
    if noise_sigma > 0:
        img += noise_sigma * rng.normal(h * w).reshape(h, w)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def _warp(img, shift, deg):
    """bilinear warp: rotate by deg about the centre then translate (host-side numpy)."""
    h, w = img.shape
    th = np.deg2rad(deg); c, s = np.cos(th), np.sin(th)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    cx, cy = (w - 1) / 2, (h - 1) / 2
    xs = c * (xx - cx) + s * (yy - cy) + cx - shift[0]
    ys = -s * (xx - cx) + c * (yy - cy) + cy - shift[1]
    xs = np.clip(xs, 0, w - 1.001); ys = np.clip(ys, 0, h - 1.001)
    x0 = xs.astype(int); y0 = ys.astype(int); fx = xs - x0; fy = ys - y0
    f = img.astype(np.float64)
    return (f[y0, x0] * (1 - fx) * (1 - fy) + f[y0, x0 + 1] * fx * (1 - fy) +
            f[y0 + 1, x0] * (1 - fx) * fy + f[y0 + 1, x0 + 1] * fx * fy)


def frame_pair(seed, h=480, w=640):
    """BASELINE config 1: frame A and frame B = A shifted 5 px, rotated 2 deg, noise sigma 2."""
    a = texture(seed, h, w, noise_sigma=0.0)
    rng = Rng(seed ^ 0x5EED)
    b = _warp(a, (5.0, 0.0), 2.0) + 2.0 * rng.normal(h * w).reshape(h, w)
    a2 = a.astype(np.float64) + 2.0 * rng.normal(h * w).reshape(h, w)
    to8 = lambda f: np.clip(np.rint(f), 0, 255).astype(np.uint8)
    return to8(a2), to8(b)


def to_rgb(gray, seed=0):
    """3-channel 8UC3 frame whose RGB2GRAY is close to `gray` (channels differ slightly)."""
    g = gray.astype(np.int16)
    rng = Rng(seed ^ 0xC0104)
    off = rng.randint(-6, 7, 3)
    rgb = np.stack([np.clip(g + off[0], 0, 255), np.clip(g + off[1], 0, 255), np.clip(g + off[2], 0, 255)], axis=-1)
    return rgb.astype(np.uint8)


def frame_stream(seed, n, h=480, w=640, margin=48, noise_sigma=1.5):
    """BASELINE config 2: n frames cut from one base texture along a seeded random walk."""
    base = texture(seed, h + 2 * margin, w + 2 * margin, noise_sigma=0.0)
    rng = Rng(seed ^ 0x57EA)
    steps = rng.randint(-3, 4, 2 * n).reshape(n, 2)
    pos = np.clip(np.cumsum(steps, axis=0) + margin, 0, 2 * margin)
    out = np.empty((n, h, w), np.uint8)
    for i in range(n):
        f = base[pos[i, 1]:pos[i, 1] + h, pos[i, 0]:pos[i, 0] + w].astype(np.float32)
        f += noise_sigma * rng.normal(h * w).reshape(h, w).astype(np.float32)
        out[i] = np.clip(np.rint(f), 0, 255).astype(np.uint8)
    return out


def descriptor_sets(seed, nq, nt, flip_p=0.08, inlier_frac=0.7):
    """train = iid uniform 256-bit rows; query = shuffled train rows with bits flipped w.p. flip_p
    for inlier_frac of the rows and fresh uniform rows for the rest."""
    rng = Rng(seed)
    train = rng.bytes(nt * 32).reshape(nt, 32)
    src = rng.randint(0, nt, nq)
    query = train[src].copy()
    inl = rng.uniform(nq) < inlier_frac
    flips = (rng.uniform(nq * 256).reshape(nq, 256) < flip_p)
    query ^= np.packbits(flips, axis=1, bitorder="little")
    fresh = rng.bytes(nq * 32).reshape(nq, 32)
    query[~inl] = fresh[~inl]
    return np.ascontiguousarray(query), np.ascontiguousarray(train)


def vocabulary_tables(seed, k=10, L=3, flip0=0.30, early_leaf_p=0.0, stop_p=0.05):
    """node table of a synthetic DBoW3-style vocabulary tree (depth-first ids like DBoW3's HKmeansStep): a child's
    descriptor is its parent's with a share of bits flipped that halves per level (clusters inside clusters);
    early_leaf_p turns inner nodes into leaves early, stop_p gives leaves weight 0 (stopped words).
    Returns parent[int32], desc[n][32], weight[float64]."""
    rng = Rng(seed)
    parent = [-1]; desc = [rng.bytes(32)]; level = [0]
    stack = [0]
    while stack:
        n = stack.pop()
        if level[n] >= L:
            continue
        if n != 0 and early_leaf_p > 0 and rng.uniform(1)[0] < early_leaf_p:
            continue
        p = flip0 / (2 ** level[n])
        kids = []
        for _ in range(k):
            flips = rng.uniform(256) < p
            d = desc[n] ^ np.packbits(flips, bitorder="little")
            parent.append(n); desc.append(d); level.append(level[n] + 1); kids.append(len(parent) - 1)
        stack.extend(reversed(kids))        # depth first, first child first
    # depth-first creation gives parent[n] < n but the ids above are breadth-inside-depth; both satisfy the table's contract
    n = len(parent)
    w = 0.5 + 4.0 * rng.uniform(n)
    w[rng.uniform(n) < stop_p] = 0.0
    par = np.array(parent, np.int32); par[0] = -1
    return par, np.array(desc, np.uint8), w


def vocabulary_tables_full(seed, k=10, L=6):
    """complete k-ary tree of depth L built level by level with array operations (ORBvoc's shape, k = 10, L = 6, is
    1 111 111 nodes): a child flips each parent bit with probability 2^-(level + 2) (AND of random bit planes).
    Breadth-first ids (parent[n] < n, children of a node contiguous and ascending).  Returns parent, desc, weight."""
    rng = Rng(seed)
    parents = [np.array([-1], np.int32)]; descs = [rng.bytes(32).reshape(1, 32)]
    first = 0
    for level in range(L):
        cur = descs[-1]; n = len(cur)
        flips = rng.bytes(n * k * 32).reshape(n * k, 32)
        for _ in range(level + 1):
            flips &= rng.bytes(n * k * 32).reshape(n * k, 32)
        descs.append(np.repeat(cur, k, axis=0) ^ flips)
        parents.append(np.repeat(np.arange(first, first + n, dtype=np.int32), k))
        first += n
    parent = np.concatenate(parents); desc = np.ascontiguousarray(np.concatenate(descs))
    w = 0.5 + 4.0 * rng.uniform(len(parent))
    w[rng.uniform(len(parent)) < 0.01] = 0.0
    return parent, desc, w


def descriptors_near_leaves(seed, n, parent, desc, flip_p=0.03):
    """n descriptors: random leaves of the tree with bits flipped w.p. flip_p"""
    rng = Rng(seed)
    leaf = np.ones(len(parent), bool); leaf[parent[1:]] = False
    ids = np.flatnonzero(leaf)
    pick = ids[rng.randint(0, len(ids), n)]
    flips = rng.uniform(n * 256).reshape(n, 256) < flip_p
    return np.ascontiguousarray(desc[pick] ^ np.packbits(flips, axis=1, bitorder="little"))


def depth_image(seed, h, w, hole_p=0.1):
    """16U depth plane: smooth surface 0.5..6 m at DepthScale 10 (65535 = 10 m) with zero-valued holes"""
    rng = Rng(seed)
    base = 0.5 + 5.5 * (0.5 + 0.5 * _value_noise(rng, h, w, 64, 1.0))
    d = np.clip(base / 10.0 * 65535.0, 0, 65535).astype(np.uint16)
    holes = _value_noise(rng, h, w, 16, 1.0) > (1.0 - 2.0 * hole_p) * 0.6
    d[holes] = 0
    return np.ascontiguousarray(d)      # the noise planes come out of fancy indexing in Fortran order
