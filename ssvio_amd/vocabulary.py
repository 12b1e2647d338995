"""Vocabulary -- the DBoW3::Vocabulary that FeatureMatching's constructor loads (src/FeatureMatching.cpp:8-9), as the node
table the C ABI takes (include/ssvio_gpu.h, ssvio_vocab_create).

DBoW3 itself is an un-vendored dependency of the reference and its binary vocabulary (vocabulary/orbvoc.dbow3) is missing
from the reference tree, so two sources are supported: arrays, and the plain-text format that DBoW3's load_fromtxt /
ORB-SLAM2's ORBvoc.txt use:
    line 1:       k L scoring weighting
    other lines:  parent_id is_leaf d0 ... d31 weight        (node ids 1, 2, ... in file order; words in leaf order)
and DBoW3's uncompressed binary stream (Vocabulary::toStream / fromStream, what `load("orbvoc.dbow3")` takes first):
    u64 88877711233 | u8 compressed | u32 nnodes | i32 k, L, scoring, weighting
    nnodes-1 records: u32 id, u32 parent, f64 weight, 3 x i32 {cols, rows, type} = {32, 1, CV_8U}, 32 descriptor bytes
    u32 nwords | nwords x (u32 word id, u32 node id)
restated from DBoW3's published source; no .dbow3 file exists here to check it against (parity unpinned), so the reader
validates everything it can and refuses what it does not understand.  The QuickLZ-compressed variant and the YAML
container are not read.
"""
import struct
import ctypes as C
import numpy as np
from . import _lib as L

L1_NORM, L2_NORM, CHI_SQUARE, KL, BHATTACHARYYA, DOT_PRODUCT = range(6)      # DBoW3 ScoringType
TF_IDF, TF, IDF, BINARY = range(4)                                           # DBoW3 WeightingType


class Vocabulary:
    def __init__(self, k, L_, parent, desc, weight, word=None, scoring=L1_NORM, weighting=TF_IDF):
        self.k, self.L, self.scoring, self.weighting = int(k), int(L_), int(scoring), int(weighting)
        self.parent = np.ascontiguousarray(parent, np.int32)
        self.desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        self.weight = np.ascontiguousarray(weight, np.float64)
        n = len(self.parent)
        if word is None:                      # words are numbered in leaf order (DBoW3 createWords / load_fromtxt)
            leaf = np.ones(n, bool); leaf[self.parent[1:]] = False
            if n > 1: leaf[0] = False
            word = np.full(n, -1, np.int32); word[leaf] = np.arange(int(leaf.sum()), dtype=np.int32)
        self.word = np.ascontiguousarray(word, np.int32)
        if not (len(self.desc) == n and len(self.weight) == n and len(self.word) == n):
            raise ValueError("vocabulary arrays disagree in length")
        self._dev = {}

    # ---- text format
    @classmethod
    def load_txt(cls, path):
        with open(path) as f:
            k, L_, scoring, weighting = (int(v) for v in f.readline().split()[:4])
            rows = [ln.split() for ln in f if ln.strip()]
        n = len(rows) + 1
        parent = np.zeros(n, np.int32); desc = np.zeros((n, 32), np.uint8); weight = np.zeros(n, np.float64)
        word = np.full(n, -1, np.int32); nw = 0
        parent[0] = -1
        for i, r in enumerate(rows, start=1):
            parent[i] = int(r[0])
            desc[i] = [int(v) for v in r[2:34]]
            weight[i] = float(r[34])
            if int(r[1]) > 0:
                word[i] = nw; nw += 1
        return cls(k, L_, parent, desc, weight, word, scoring, weighting)

    def save_txt(self, path):
        kids = np.zeros(len(self.parent), bool); kids[self.parent[1:]] = True
        with open(path, "w") as f:
            f.write("%d %d %d %d\n" % (self.k, self.L, self.scoring, self.weighting))
            for i in range(1, len(self.parent)):
                f.write("%d %d %s %r\n" % (self.parent[i], 0 if kids[i] else 1, " ".join(str(int(b)) for b in self.desc[i]), float(self.weight[i])))

    # ---- DBoW3 binary stream (uncompressed)
    DBOW3_SIG = 88877711233

    @classmethod
    def is_dbow3(cls, path):
        with open(path, "rb") as f:
            head = f.read(8)
        return len(head) == 8 and struct.unpack("<Q", head)[0] == cls.DBOW3_SIG

    @classmethod
    def load_dbow3(cls, path):
        buf = open(path, "rb").read()
        if len(buf) < 29 or struct.unpack_from("<Q", buf, 0)[0] != cls.DBOW3_SIG:
            raise ValueError("not a DBoW3 binary vocabulary")
        if buf[8] != 0:
            raise ValueError("QuickLZ-compressed DBoW3 vocabulary: save it uncompressed (Vocabulary::save(file, false)) first")
        n, k, L_, scoring, weighting = struct.unpack_from("<Iiiii", buf, 9)
        rec = 4 + 4 + 8 + 12 + 32
        if n < 1 or len(buf) < 29 + (n - 1) * rec + 4:
            raise ValueError("truncated DBoW3 vocabulary")
        r = np.frombuffer(buf, np.uint8, (n - 1) * rec, 29).reshape(n - 1, rec)
        ids = r[:, 0:4].copy().view("<u4").ravel().astype(np.int64); par = r[:, 4:8].copy().view("<u4").ravel().astype(np.int64)
        hdr = r[:, 16:28].copy().view("<i4").reshape(-1, 3)
        if n > 1 and not (np.sort(hdr, axis=1) == [0, 1, 32]).all():
            raise ValueError("descriptor header is not {32 cols, 1 row, CV_8U}")
        if n > 1 and (ids.min() < 1 or ids.max() >= n or len(np.unique(ids)) != n - 1 or (par >= ids).any()):
            raise ValueError("node ids / parents are not a tree numbered parent-first")
        # children are tried in file order (fromStream push_back): equal to ascending id, or the table cannot express it
        order = np.lexsort((np.arange(n - 1), par))
        if (np.diff(ids[order])[np.diff(par[order]) == 0] <= 0).any():
            raise ValueError("children of a node are not stored in ascending id order")
        parent = np.full(n, -1, np.int32); desc = np.zeros((n, 32), np.uint8); weight = np.zeros(n, np.float64)
        parent[ids] = par; desc[ids] = r[:, 28:60]; weight[ids] = r[:, 8:16].copy().view("<f8").ravel()
        off = 29 + (n - 1) * rec
        nw = struct.unpack_from("<I", buf, off)[0]
        if len(buf) < off + 4 + 8 * nw:
            raise ValueError("truncated word list")
        wl = np.frombuffer(buf, "<u4", 2 * nw, off + 4).reshape(nw, 2).astype(np.int64)
        leaf = np.ones(n, bool); leaf[parent[1:]] = False
        if n > 1: leaf[0] = False
        if nw != int(leaf.sum()) or (nw and (wl[:, 0].max() >= nw or wl[:, 1].max() >= n or not leaf[wl[:, 1]].all()
                                            or len(np.unique(wl[:, 0])) != nw or len(np.unique(wl[:, 1])) != nw)):
            raise ValueError("word list does not number the leaves")
        word = np.full(n, -1, np.int32); word[wl[:, 1]] = wl[:, 0]
        return cls(k, L_, parent, desc, weight, word, scoring, weighting)

    def save_dbow3(self, path):
        """uncompressed toStream: header, nodes in the order of its stack walk from the root, then the word list"""
        n = len(self.parent)
        kids = [[] for _ in range(n)]
        for i in range(1, n): kids[self.parent[i]].append(i)
        out = [struct.pack("<QBIiiii", self.DBOW3_SIG, 0, n, self.k, self.L, self.scoring, self.weighting)]
        stack = [0]
        while stack:
            pid = stack.pop()
            for c in kids[pid]:
                out.append(struct.pack("<IIdiii", c, pid, float(self.weight[c]), 32, 1, 0) + self.desc[c].tobytes())
                if kids[c]: stack.append(c)
        words = np.flatnonzero(self.word >= 0)
        words = words[np.argsort(self.word[words])]
        out.append(struct.pack("<I", len(words)))
        out += [struct.pack("<II", int(self.word[i]), int(i)) for i in words]
        with open(path, "wb") as f:
            f.write(b"".join(out))

    @classmethod
    def load(cls, path):
        """DBoW3 Vocabulary::load: the binary stream when the signature is there, else the text layout"""
        return cls.load_dbow3(path) if cls.is_dbow3(path) else cls.load_txt(path)

    # ---- device handle (one per GPU)
    def handle(self, device=0):
        h = self._dev.get(device)
        if h is None:
            h = C.c_void_p()
            L.check(L.lib().ssvio_vocab_create(device, self.k, self.L, len(self.parent), L.ptr(self.parent), L.ptr(self.desc),
                                               L.ptr(self.weight), L.ptr(self.word), self.scoring, self.weighting, C.byref(h)))
            self._dev[device] = h
        return h

    def close(self):
        for h in self._dev.values():
            if h.value and L is not None and L.lib is not None:
                L.lib().ssvio_vocab_destroy(h)
        self._dev = {}

    __del__ = close

    # ---- host halves of Vocabulary::transform(features, v, fv, levelsup): the two maps, from the per-feature results
    def bow_vector(self, word, weight):
        """DBoW3::BowVector as (word ids ascending, values): addWeight / addIfNotExist in feature order, then the
        normalisation the scoring type asks for"""
        acc = {}
        for w, v in zip(np.asarray(word).tolist(), np.asarray(weight, np.float64).tolist()):
            if not v > 0:
                continue
            if w in acc:
                if self.weighting in (TF_IDF, TF):
                    acc[w] += v
            else:
                acc[w] = v
        ids = sorted(acc)
        vals = np.array([acc[i] for i in ids], np.float64)
        must = self.scoring != DOT_PRODUCT
        if self.weighting in (TF_IDF, TF) and len(ids) and not must:
            vals = vals / float(len(ids))
        if must and len(ids):
            norm = 0.0
            if self.scoring == L2_NORM:
                for v in vals.tolist(): norm += v * v
                norm = float(np.sqrt(norm))
            else:
                for v in vals.tolist(): norm += abs(v)
            if norm > 0.0:
                vals = vals / norm
        return np.array(ids, np.int32), vals

    @staticmethod
    def feature_vector(node, weight):
        """DBoW3::FeatureVector as {node id: ascending feature indices}"""
        fv = {}
        for i, (n, v) in enumerate(zip(np.asarray(node).tolist(), np.asarray(weight).tolist())):
            if v > 0:
                fv.setdefault(n, []).append(i)
        return dict(sorted(fv.items()))
