"""Vocabulary -- the DBoW3::Vocabulary that FeatureMatching's constructor loads (src/FeatureMatching.cpp:8-9), as the node
table the C ABI takes (include/ssvio_gpu.h, ssvio_vocab_create).

DBoW3 itself is an un-vendored dependency of the reference and its binary vocabulary (vocabulary/orbvoc.dbow3) is missing
from the reference tree, so two sources are supported: arrays, and the plain-text format that DBoW3's load_fromtxt /
ORB-SLAM2's ORBvoc.txt use:
    line 1:       k L scoring weighting
    other lines:  parent_id is_leaf d0 ... d31 weight        (node ids 1, 2, ... in file order; words in leaf order)
The compressed binary .dbow3 container is not read (its layout cannot be checked here).
"""
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
