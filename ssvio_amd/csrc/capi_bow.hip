// capi_bow.hip -- the C ABI of include/ssvio_gpu.h, part 3: the vocabulary handle and MatchByDBoW.
#include "capi_internal.h"

// =====================================================================================
// MatchByDBoW (src/FeatureMatching.cpp:58-125, :192-208)
struct ssvio_vocab {
    int device = 0, k = 0, L = 0, nnodes = 0, scoring = 0, weighting = 0;
    uint4* d_cdesc = nullptr; int* d_childBegin = nullptr; int* d_child = nullptr; int* d_word = nullptr; double* d_weight = nullptr;
};

static VocabDev vocab_dev(const ssvio_vocab* v)
{
    VocabDev d{};
    d.cdesc = v->d_cdesc; d.childBegin = v->d_childBegin; d.child = v->d_child; d.word = v->d_word; d.weight = v->d_weight;
    d.L = v->L; d.nnodes = v->nnodes;
    return d;
}

static int bow_check(ssvio_matcher* m, const ssvio_vocab* v, int levelsup)
{
    if (mt_check(m)) return SSVIO_ERR_INVALID;
    if (!v) { set_error("null vocabulary (MatchByDBoW needs one: ssvio_vocab_create)"); return SSVIO_ERR_INVALID; }
    if (v->device != m->device) { set_error("vocabulary lives on device %d, matcher on %d", v->device, m->device); return SSVIO_ERR_INVALID; }
    if (levelsup < 0) { set_error("levelsup < 0"); return SSVIO_ERR_INVALID; }
    return SSVIO_OK;
}

static int bow_grow(ssvio_matcher* m, size_t rows)
{
    hipStream_t s = m->stream;
    int rc = grow(m->d_bowWord, m->bowWordCap, rows, s); if (rc) return rc;
    rc = grow(m->d_bowNode, m->bowNodeCap, rows, s); if (rc) return rc;
    rc = grow(m->d_bowWeight, m->bowWeightCap, rows, s); if (rc) return rc;
    rc = grow(m->d_sortIdx, m->sortIdxCap, rows, s); if (rc) return rc;
    rc = grow(m->d_sortNode, m->sortNodeCap, rows, s); if (rc) return rc;
    return grow(m->d_gnode, m->gnodeCap, rows, s);
}

// order the rows of `nsets` transformed sets by (node id, index); same row indexing as bow_run
static void bow_sort(ssvio_matcher* m, int setStrideRows, const int* d_counts, int nsets, int maxRows, size_t rowBase)
{
    BowSortArgs a{};
    a.gnode = m->d_gnode + rowBase; a.setStrideRows = setStrideRows; a.counts = d_counts; a.maxRows = maxRows;
    a.sortedIdx = m->d_sortIdx + rowBase; a.sortedNode = m->d_sortNode + rowBase;
    ProfScope p(m->prof, "k_bow_sort", m->stream);
    launch_bow_sort(a, nsets, m->stream);
}

// transform the rows of `nsets` descriptor sets; per-feature results land at rowBase + set * setStrideRows + i
static void bow_run(ssvio_matcher* m, const ssvio_vocab* v, const uint8_t* d_desc, int setStrideRows, const int* d_counts, int nsets,
                    int maxRows, int levelsup, size_t rowBase, bool toHost = false)
{
    BowArgs a{};
    a.v = vocab_dev(v); a.nidLevel = v->L - levelsup;
    a.desc = (const uint4*)d_desc; a.setStrideRows = setStrideRows; a.counts = d_counts; a.nsets = nsets; a.maxRows = maxRows;
    a.word = m->d_bowWord + rowBase; a.node = m->d_bowNode + rowBase; a.weight = m->d_bowWeight + rowBase; a.gnode = m->d_gnode + rowBase;
    if (toHost) {      // only gnode feeds later kernels: the three per-feature results go straight to mapped host memory (bow_host_area)
        a.word = (int32_t*)m->d_bowm + rowBase; a.node = (int32_t*)m->d_bowm + m->bowmRows + rowBase;
        a.weight = (double*)(m->d_bowm + 8 * m->bowmRows) + rowBase;
    }
    ProfScope p(m->prof, "k_bow_transform", m->stream);
    launch_bow_transform(a, maxRows, m->stream);
}

// mapped host area for the per-feature transform results of `rows` features: [rows] word, [rows] node, [rows] weight
static int bow_host_area(ssvio_matcher* m, size_t rows)
{
    if (16 * rows > m->bowmCap) {
        SSVIO_HIP(hipStreamSynchronize(m->stream));
        hfree(m->h_bowm); m->bowmCap = 0; m->d_bowm = nullptr;
        const size_t want = 16 * (rows + rows / 4);
        SSVIO_HIP(hipHostMalloc((void**)&m->h_bowm, want, hipHostMallocMapped));
        SSVIO_HIP(hipHostGetDevicePointer((void**)&m->d_bowm, m->h_bowm, 0));
        m->bowmCap = want;
    }
    m->bowmRows = rows;
    return SSVIO_OK;
}

static void bow_host_copy(const ssvio_matcher* m, size_t base, int rows, int32_t* word, int32_t* node, double* weight)
{
    if (word) memcpy(word, (const int32_t*)m->h_bowm + base, (size_t)rows * 4);
    if (node) memcpy(node, (const int32_t*)m->h_bowm + m->bowmRows + base, (size_t)rows * 4);
    if (weight) memcpy(weight, (const double*)(m->h_bowm + 8 * m->bowmRows) + base, (size_t)rows * 8);
}

extern "C" {

int ssvio_vocab_create(int device, int k, int L, int nnodes, const int32_t* parent, const uint8_t* desc, const double* weight,
                       const int32_t* word, int scoring, int weighting, ssvio_vocab** out)
{
    if (!out) { set_error("null argument"); return SSVIO_ERR_INVALID; }
    *out = nullptr;
    if (!parent || !desc || !weight || !word || nnodes < 1 || k < 1 || L < 0) { set_error("bad vocabulary arguments"); return SSVIO_ERR_INVALID; }
    if (nnodes > (1 << 30)) { set_error("too many nodes"); return SSVIO_ERR_INVALID; }
    std::vector<int> begin((size_t)nnodes + 1, 0), child((size_t)std::max(nnodes - 1, 1)), fill((size_t)nnodes, 0);
    for (int n = 1; n < nnodes; ++n) {
        if (parent[n] < 0 || parent[n] >= n) { set_error("node %d: parent %d is not an earlier node", n, parent[n]); return SSVIO_ERR_INVALID; }
        begin[(size_t)parent[n] + 1]++;
    }
    for (int n = 0; n < nnodes; ++n) {
        if (begin[(size_t)n + 1] >= (1 << 20)) { set_error("node %d has 2^20 or more children", n); return SSVIO_ERR_INVALID; }
        begin[(size_t)n + 1] += begin[n];
    }
    std::vector<uint8_t> cdesc((size_t)std::max(nnodes - 1, 1) * 32);
    for (int n = 1; n < nnodes; ++n) {                      // ascending id inside every child list
        const int c = begin[parent[n]] + fill[parent[n]]++;
        child[c] = n;
        memcpy(&cdesc[(size_t)c * 32], desc + (size_t)n * 32, 32);
    }
    const int rc = pick_device(device);
    if (rc) return rc;
    SSVIO_HIP(hipSetDevice(device));
    ssvio_vocab* v = new ssvio_vocab();
    v->device = device; v->k = k; v->L = L; v->nnodes = nnodes; v->scoring = scoring; v->weighting = weighting;
    bool ok = hipMalloc(&v->d_cdesc, cdesc.size()) == hipSuccess && hipMalloc(&v->d_childBegin, begin.size() * sizeof(int)) == hipSuccess &&
              hipMalloc(&v->d_child, child.size() * sizeof(int)) == hipSuccess && hipMalloc(&v->d_word, (size_t)nnodes * sizeof(int)) == hipSuccess &&
              hipMalloc(&v->d_weight, (size_t)nnodes * sizeof(double)) == hipSuccess;
    ok = ok && hipMemcpy(v->d_cdesc, cdesc.data(), cdesc.size(), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(v->d_childBegin, begin.data(), begin.size() * sizeof(int), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(v->d_child, child.data(), child.size() * sizeof(int), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(v->d_word, word, (size_t)nnodes * sizeof(int), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(v->d_weight, weight, (size_t)nnodes * sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) { set_error("cannot place the vocabulary on the device"); ssvio_vocab_destroy(v); return SSVIO_ERR_HIP; }
    *out = v;
    return SSVIO_OK;
}

void ssvio_vocab_destroy(ssvio_vocab* v)
{
    if (!v) return;
    (void)hipSetDevice(v->device);
    (void)hipDeviceSynchronize();
    dfree(v->d_cdesc); dfree(v->d_childBegin); dfree(v->d_child); dfree(v->d_word); dfree(v->d_weight);
    delete v;
}

int ssvio_vocab_info(const ssvio_vocab* v, int* k, int* L, int* nnodes, int* scoring, int* weighting)
{
    if (!v) { set_error("null vocabulary"); return SSVIO_ERR_INVALID; }
    if (k) *k = v->k; if (L) *L = v->L; if (nnodes) *nnodes = v->nnodes; if (scoring) *scoring = v->scoring; if (weighting) *weighting = v->weighting;
    return SSVIO_OK;
}

int ssvio_bow_transform(ssvio_matcher* m, const ssvio_vocab* v, const uint8_t* desc, int n, int stride, int levelsup,
                        int32_t* word, int32_t* node, double* weight)
{
    int rc = bow_check(m, v, levelsup); if (rc) return rc;
    if (n < 0) { set_error("negative row count"); return SSVIO_ERR_INVALID; }
    if (n == 0) return SSVIO_OK;
    if (!desc || stride < 32) { set_error("bad descriptor matrix"); return SSVIO_ERR_INVALID; }
    SSVIO_HIP(hipSetDevice(m->device));
    hipStream_t s = m->stream;
    rc = bow_grow(m, (size_t)n); if (rc) return rc;
    rc = bow_host_area(m, (size_t)n); if (rc) return rc;
    const uint8_t* d_rows; const uint8_t* d_none;
    rc = stage_pair(m, desc, n, stride, desc, 0, 32, &d_rows, &d_none); if (rc) return rc;
    m->h_counts2[0] = n; m->h_counts2[1] = 0;
    bow_run(m, v, d_rows, 0, m->d_counts2m, 1, n, levelsup, 0, true);
    SSVIO_HIP(hipGetLastError());
    SSVIO_HIP(hipStreamSynchronize(s));
    bow_host_copy(m, 0, n, word, node, weight);
    m->prof.resolve();
    m->lastBowQ = n; m->lastBowT = 0;
    return SSVIO_OK;
}

int ssvio_match_dbow(ssvio_matcher* m, const ssvio_vocab* v, const uint8_t* q, int nq, int qstride,
                     const uint8_t* t, int nt, int tstride, int levelsup, ssvio_dmatch* out, int cap, int* n)
{
    int rc = bow_check(m, v, levelsup); if (rc) return rc;
    if (n) *n = 0;
    if (nq < 0 || nt < 0) { set_error("negative row count"); return SSVIO_ERR_INVALID; }
    if (!(nq > 0 && nt > 0)) return SSVIO_OK;                          // src/FeatureMatching.cpp:61-63
    if (!q || !t || !out || !n) { set_error("null argument"); return SSVIO_ERR_INVALID; }
    if (qstride < 32 || tstride < 32) { set_error("descriptor stride < 32"); return SSVIO_ERR_INVALID; }
    if (nt > (1 << 23) - 2 || nq > (1 << 23)) { set_error("more than 2^23 rows per set"); return SSVIO_ERR_INVALID; }
    if (cap < nq) { set_error("output capacity %d < %d query rows", cap, nq); return SSVIO_ERR_CAPACITY; }
    SSVIO_HIP(hipSetDevice(m->device));
    hipStream_t s = m->stream;
    rc = grow(m->d_cand, m->candCap, (size_t)nq, s); if (rc) return rc;
    rc = grow(m->d_keepNode, m->keepNodeCap, (size_t)nq, s); if (rc) return rc;
    // no copy commands and one wait: descriptors (small sets), counts, per-feature transform results and the match list all live
    // in mapped host memory that the kernels read / write directly
    rc = ensure_result(m, (size_t)nq * sizeof(ssvio_dmatch)); if (rc) return rc;
    const uint8_t* d_q; const uint8_t* d_t;
    rc = stage_pair(m, q, nq, qstride, t, nt, tstride, &d_q, &d_t); if (rc) return rc;
    // When both sets sit in one block (the mapped staging area), they are two sets of one launch: set stride = their distance in rows.
    const bool oneBlock = d_t > d_q && (size_t)(d_t - d_q) % 32 == 0 && (size_t)(d_t - d_q) / 32 >= (size_t)nq && (size_t)(d_t - d_q) / 32 < (size_t)nq + 64;
    const size_t tBase = oneBlock ? (size_t)(d_t - d_q) / 32 : (size_t)nq;      // where the train side's per-feature rows start
    rc = bow_grow(m, tBase + nt); if (rc) return rc;
    rc = bow_host_area(m, tBase + nt); if (rc) return rc;
    m->h_counts2[0] = nq; m->h_counts2[1] = nt;
    *(volatile int*)m->h_result = 0;
    const int* cq = m->d_counts2m; const int* ct = m->d_counts2m + 1;
    ssvio_dmatch* outTo = (ssvio_dmatch*)(m->d_resultm + 16); int* noutTo = (int*)m->d_resultm;
    const bool sorted = std::max(nq, nt) <= kBowSortMax && !getenv("SSVIO_DBOW_UNSORTED");
    if (oneBlock) {
        bow_run(m, v, d_q, (int)tBase, cq, 2, std::max(nq, nt), levelsup, 0, true);      // ComputeBoWVector(desp1), (desp2), :68-69
        if (sorted) bow_sort(m, (int)tBase, cq, 2, std::max(nq, nt), 0);
    } else {
        bow_run(m, v, d_q, 0, cq, 1, nq, levelsup, 0, true);
        bow_run(m, v, d_t, 0, ct, 1, nt, levelsup, tBase, true);
        if (sorted) { bow_sort(m, 0, cq, 1, nq, 0); bow_sort(m, 0, ct, 1, nt, tBase); }
    }
    if (sorted) {
        SortedMatchArgs g{};
        g.qdesc = (const uint4*)d_q; g.tdesc = (const uint4*)d_t;
        g.qSortedIdx = m->d_sortIdx; g.qSortedNode = m->d_sortNode; g.tSortedIdx = m->d_sortIdx + tBase; g.tSortedNode = m->d_sortNode + tBase;
        g.qSetStrideRows = 0; g.tSetStrideRows = 0; g.qcounts = cq; g.tcounts = ct; g.pairs = nullptr;
        g.threshold = m->threshold; g.nnRatio = m->nnRatio; g.maxRows = nq; g.maxT = nt;
        g.cand = m->d_cand; g.keep = m->d_keepNode; g.out = outTo; g.nout = noutTo;
        ProfScope p(m->prof, "k_match_sorted", s);
        launch_match_sorted(g, nq, nt, 1, s);
        SSVIO_HIP(hipGetLastError());
    } else {                                                           // sets beyond the LDS sort: wave per query over all train rows
        GroupedArgs g{};
        g.qdesc = (const uint4*)d_q; g.tdesc = (const uint4*)d_t; g.qgnode = m->d_gnode; g.tgnode = m->d_gnode + tBase;
        g.qSetStrideRows = 0; g.tSetStrideRows = 0; g.qcounts = cq; g.tcounts = ct; g.pairs = nullptr;
        g.threshold = m->threshold; g.nnRatio = m->nnRatio; g.maxRows = nq; g.maxT = nt;
        g.cand = m->d_cand; g.keepNode = m->d_keepNode; g.out = outTo; g.nout = noutTo;
        ProfScope p(m->prof, "k_match_grouped", s);
        launch_match_grouped(g, nq, 1, s);
        SSVIO_HIP(hipGetLastError());
    }
    SSVIO_HIP(hipStreamSynchronize(s));
    const int got = *(const volatile int*)m->h_result;
    const int cnt = std::min(std::max(got, 0), nq);
    if (cnt > 0) memcpy(out, m->h_result + 16, (size_t)cnt * sizeof(ssvio_dmatch));
    *n = cnt;
    m->prof.resolve();
    m->lastBowQ = nq; m->lastBowT = nt; m->lastBowTBase = tBase;
    return SSVIO_OK;
}

int ssvio_bow_last_transform(ssvio_matcher* m, int side, int32_t* word, int32_t* node, double* weight, int cap, int* n)
{
    if (mt_check(m)) return SSVIO_ERR_INVALID;
    if (side != 0 && side != 1) { set_error("side must be 0 (desp1) or 1 (desp2)"); return SSVIO_ERR_INVALID; }
    const int rows = side ? m->lastBowT : m->lastBowQ;
    const size_t base = side ? m->lastBowTBase : 0;
    if (n) *n = rows;
    if (rows == 0) return SSVIO_OK;
    if (cap < rows) { set_error("capacity %d < %d rows", cap, rows); return SSVIO_ERR_CAPACITY; }
    bow_host_copy(m, base, rows, word, node, weight);                  // the last host call left them in mapped host memory
    return SSVIO_OK;
}

int ssvio_match_dbow_batch_dev(ssvio_matcher* m, const ssvio_vocab* v, const void* d_desc, int set_stride_rows, int max_rows,
                               const int32_t* d_counts, int nsets, const int32_t* pairs, int npairs, int levelsup,
                               const void** d_matches, const int32_t** d_nmatches)
{
    int rc = bow_check(m, v, levelsup); if (rc) return rc;
    if (!d_desc || !d_counts || !pairs || npairs <= 0 || nsets <= 0 || max_rows <= 0 || set_stride_rows < max_rows) { set_error("bad batch arguments"); return SSVIO_ERR_INVALID; }
    if (max_rows > (1 << 23) - 2) { set_error("more than 2^23 rows per set"); return SSVIO_ERR_INVALID; }
    for (int i = 0; i < 2 * npairs; ++i) if (pairs[i] < 0 || pairs[i] >= nsets) { set_error("pair %d names set %d of %d", i / 2, pairs[i], nsets); return SSVIO_ERR_INVALID; }
    SSVIO_HIP(hipSetDevice(m->device));
    hipStream_t s = m->stream;
    rc = upload_pairs(m, pairs, npairs); if (rc) return rc;
    rc = bow_grow(m, (size_t)nsets * set_stride_rows); if (rc) return rc;
    rc = grow(m->d_cand, m->candCap, (size_t)npairs * max_rows, s); if (rc) return rc;
    rc = grow(m->d_keepNode, m->keepNodeCap, (size_t)npairs * max_rows, s); if (rc) return rc;
    rc = grow(m->d_out, m->outCap, (size_t)npairs * max_rows, s); if (rc) return rc;
    rc = grow(m->d_nout, m->noutCap, (size_t)npairs, s); if (rc) return rc;
    bow_run(m, v, (const uint8_t*)d_desc, set_stride_rows, d_counts, nsets, max_rows, levelsup, 0);   // once per set, not per pair
    if (max_rows <= kBowSortMax && !getenv("SSVIO_DBOW_UNSORTED")) {
        bow_sort(m, set_stride_rows, d_counts, nsets, max_rows, 0);
        SortedMatchArgs g{};
        g.qdesc = (const uint4*)d_desc; g.tdesc = (const uint4*)d_desc;
        g.qSortedIdx = m->d_sortIdx; g.qSortedNode = m->d_sortNode; g.tSortedIdx = m->d_sortIdx; g.tSortedNode = m->d_sortNode;
        g.qSetStrideRows = set_stride_rows; g.tSetStrideRows = set_stride_rows; g.qcounts = d_counts; g.tcounts = d_counts; g.pairs = m->d_pairs;
        g.threshold = m->threshold; g.nnRatio = m->nnRatio; g.maxRows = max_rows; g.maxT = max_rows;
        g.cand = m->d_cand; g.keep = m->d_keepNode; g.out = m->d_out; g.nout = m->d_nout;
        ProfScope p(m->prof, "k_match_sorted", s);
        launch_match_sorted(g, max_rows, max_rows, npairs, s);
        SSVIO_HIP(hipGetLastError());
    } else {
        GroupedArgs g{};
        g.qdesc = (const uint4*)d_desc; g.tdesc = (const uint4*)d_desc; g.qgnode = m->d_gnode; g.tgnode = m->d_gnode;
        g.qSetStrideRows = set_stride_rows; g.tSetStrideRows = set_stride_rows; g.qcounts = d_counts; g.tcounts = d_counts; g.pairs = m->d_pairs;
        g.threshold = m->threshold; g.nnRatio = m->nnRatio; g.maxRows = max_rows; g.maxT = max_rows;
        g.cand = m->d_cand; g.keepNode = m->d_keepNode; g.out = m->d_out; g.nout = m->d_nout;
        ProfScope p(m->prof, "k_match_grouped", s);
        launch_match_grouped(g, max_rows, npairs, s);
        SSVIO_HIP(hipGetLastError());
    }
    m->lastPairs = npairs; m->lastMaxRows = max_rows; m->lastHadPairs = true;
    m->lastBowQ = 0; m->lastBowT = 0;
    if (d_matches) *d_matches = m->d_out;
    if (d_nmatches) *d_nmatches = m->d_nout;
    return SSVIO_OK;
}

}  // extern "C"
