// capi_match.hip -- the C ABI of include/ssvio_gpu.h, part 2: the matcher handle (class FeatureMatching): brute-force / 2-NN /
// cross-check matching, MatchByProject, the batched device-resident entry points.
#include "capi_internal.h"

// descriptor rows host -> device.  Continuous rows (the usual cv::Mat) are one copy; strided rows are packed on the
// host first: a 2-D copy of 32-byte rows costs a transfer per row (18 ms for 1000 rows on this stack).
int upload_rows(uint8_t* dst, const uint8_t* src, int n, int stride, std::vector<uint8_t>& staging, hipStream_t s)
{
    if (n <= 0) return SSVIO_OK;
    if (stride == 32) { SSVIO_HIP(hipMemcpyAsync(dst, src, (size_t)n * 32, hipMemcpyHostToDevice, s)); return SSVIO_OK; }
    staging.resize((size_t)n * 32);
    for (int i = 0; i < n; ++i) memcpy(staging.data() + (size_t)i * 32, src + (size_t)i * stride, 32);
    SSVIO_HIP(hipMemcpyAsync(dst, staging.data(), (size_t)n * 32, hipMemcpyHostToDevice, s));
    return SSVIO_OK;
}

// Choose how many pieces the train set is cut into.  Target: ~2048 workgroups (8 waves per SIMD) of at least 64 train rows each, so
// that every SIMD ends up with the same number of waves.  The kernel splits the rows a set really holds evenly over the chunks, so
// the chunk count -- not the chunk size -- is what this fixes; chunkRows is the upper bound for the partial lists' layout.
// Query rows: a set handed over by the extractor holds up to 32 rows less than it can (the quadtree's overshoot slack); counting
// query blocks without that slack keeps a block that will be empty out of the balance.
static void pick_chunks(int maxQ, int maxT, int npairs, bool mfma, int slack, int& qBlocksOut, int& nChunks, int& chunkRows)
{
    if (mfma) {
        // k_hamming_knn2_mfma: a workgroup expands its query rows into registers once (~100 instructions per 32 rows) and a chunk holds
        // at most kMfmaChunkRows rows; ~1024 workgroups (two per CU, two rounds) of at least 256 train rows each
        static const int targetM = [] { const char* e = getenv("SSVIO_MFMA_BLOCKS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 1024; }();
        qBlocksOut = knn2_mfma_query_blocks(maxQ, slack);
        const long qBlocks = (long)qBlocksOut * npairs;
        long want = (targetM + qBlocks / 2) / qBlocks;
        const long minChunks = (maxT + kMfmaChunkRows - 1) / kMfmaChunkRows, maxChunks = std::max<long>(minChunks, std::min<long>(64, maxT / 256));
        want = std::max(minChunks, std::min(want, maxChunks));
        nChunks = (int)std::max<long>(want, 1);
        chunkRows = ((maxT + nChunks - 1) / nChunks + 31) / 32 * 32;
        return;
    }
    static const int forcedRows = [] { const char* e = getenv("SSVIO_MATCH_CHUNKROWS"); return e ? atoi(e) : 0; }();   // tuning knob
    static const int target = [] { const char* e = getenv("SSVIO_MATCH_BLOCKS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 4096; }();    // workgroups per launch the chunking aims at (2048 until round 3: 2-5 % slower on every shape of match_r_sweep.py)
    qBlocksOut = knn2_query_blocks(maxQ, slack);
    const long qBlocks = (long)qBlocksOut * npairs;
    long want = (target + qBlocks / 2) / qBlocks;
    const long maxChunks = std::max(1, (maxT + 63) / 64);
    want = std::min<long>(std::min<long>(want, maxChunks), 64);
    nChunks = (int)std::max<long>(want, 1);
    chunkRows = ((maxT + nChunks - 1) / nChunks + 7) / 8 * 8;
    if (forcedRows >= 8) { chunkRows = forcedRows / 8 * 8; nChunks = std::max(1, (maxT + chunkRows - 1) / chunkRows); }
}

// the pair list of the batched entry points, on the device (an unchanged list is not uploaded again)
int upload_pairs(ssvio_matcher* m, const int32_t* pairs, int npairs)
{
    if (m->d_pairs && m->h_pairs.size() == (size_t)npairs * 2 && memcmp(m->h_pairs.data(), pairs, (size_t)npairs * 2 * sizeof(int32_t)) == 0) return SSVIO_OK;
    // the cache describes the device copy only after everything below has succeeded: a failed allocation or copy must not leave an
    // earlier list looking current (the next call with that list would launch on a stale or null d_pairs)
    m->h_pairs.clear();
    int rc = grow(m->d_pairs, m->pairsCap, (size_t)npairs * 2, m->stream); if (rc) return rc;
    SSVIO_HIP(hipStreamSynchronize(m->stream));    // earlier launches may still read the old list
    SSVIO_HIP(hipMemcpyAsync(m->d_pairs, pairs, (size_t)npairs * 2 * sizeof(int), hipMemcpyHostToDevice, m->stream));
    SSVIO_HIP(hipStreamSynchronize(m->stream));
    m->h_pairs.assign(pairs, pairs + (size_t)npairs * 2);
    return SSVIO_OK;
}

// how a batch of pairs is cut into launches: chunks of the train set (forward pass; reverse pass of the cross-check) and the
// partial-list entries (uint2) they need.  limitEntries > 0 caps the scratch (the no-allocation entry point): chunks are
// halved until both lists fit, SSVIO_ERR_CAPACITY if one chunk per pair does not.
int match_plan(int maxQ, int maxT, int npairs, int mode, int kernel, int slack, size_t limitEntries, MatchPlan& p)
{
    p.maxRows = std::max(maxQ, maxT);
    p.mfma = kernel == SSVIO_KERNEL_MFMA || kernel == SSVIO_KERNEL_MFMA_ALL;
    const bool smallOk = kernel != SSVIO_KERNEL_MFMA_ALL;
    p.qBlocksR = 0;
    pick_chunks(maxQ, maxT, npairs, p.mfma, slack, p.qBlocks, p.nChunks, p.chunkRows);
    p.small = smallOk && knn2_small_applies(maxQ, maxT, npairs); p.smallRev = false;
    if (p.small) { p.nChunks = 1; p.chunkRows = maxT; }
    p.nChunksR = 0; p.chunkRowsR = 0;
    const bool rev = mode == SSVIO_MODE_CROSSCHECK || mode == SSVIO_MODE_CROSSCHECK_CV;
    if (mode == SSVIO_MODE_CROSSCHECK_CV) { p.nChunks = 1; p.chunkRows = maxT; p.small = false; }     // no forward search: the forward list holds the votes
    if (rev) {
        pick_chunks(maxT, maxQ, npairs, p.mfma, slack, p.qBlocksR, p.nChunksR, p.chunkRowsR);
        p.smallRev = smallOk && knn2_small_applies(maxT, maxQ, npairs);
        if (p.smallRev) { p.nChunksR = 1; p.chunkRowsR = maxQ; }
    }
    const size_t per = (size_t)npairs * p.maxRows;
    if (limitEntries) {
        const size_t lim = rev ? limitEntries / 2 : limitEntries;
        if (per > lim) { set_error("%d pairs x %d rows need %zu bytes of scratch, the fixed scratch holds %zu", npairs, p.maxRows, per * 8 * (rev ? 2 : 1), limitEntries * 8); return SSVIO_ERR_CAPACITY; }
        const int gran = p.mfma ? 32 : 8;
        auto fit = [&](int& nChunks, int& chunkRows, int rowsT) {
            if ((size_t)nChunks * per <= lim) return;
            const int want = (int)(lim / per);
            chunkRows = ((rowsT + want - 1) / want + gran - 1) / gran * gran;
            nChunks = std::max(1, (rowsT + chunkRows - 1) / chunkRows);
        };
        fit(p.nChunks, p.chunkRows, maxT);
        if (rev) fit(p.nChunksR, p.chunkRowsR, maxQ);
        if (p.mfma && ((!p.small && mode != SSVIO_MODE_CROSSCHECK_CV && p.chunkRows > kMfmaChunkRows) || (rev && !p.smallRev && p.chunkRowsR > kMfmaChunkRows))) {
            set_error("%d pairs x %d rows: the fixed scratch leaves chunks of more than %d train rows, beyond the MFMA kernel", npairs, p.maxRows, kMfmaChunkRows);
            return SSVIO_ERR_CAPACITY;
        }
    }
    p.partialEntries = (size_t)p.nChunks * per; p.partialRevEntries = (size_t)p.nChunksR * per;
    return SSVIO_OK;
}

// the launches of one batch: 2-NN per chunk (both directions for the cross-check), merge, filter.  Every buffer is the caller's.
int match_launch(ssvio_matcher* m, const MatchPlan& p, const uint8_t* d_qdesc, int qStrideRows, const int* d_qcounts,
                 const uint8_t* d_tdesc, int tStrideRows, const int* d_tcounts, const int* d_pairs, int npairs, int maxQ, int maxT, int mode,
                 uint2* partial, uint2* partialRev, ssvio_dmatch* out, int* nout, int32_t* raw, bool profile)
{
    hipStream_t s = m->stream;
    KernelProfiler off;                                  // a disabled profiler: no events are created or recorded
    KernelProfiler& prof = profile ? m->prof : off;
    MatchArgs a{};
    a.qdesc = (const uint4*)d_qdesc; a.tdesc = (const uint4*)d_tdesc;
    a.qSetStrideRows = qStrideRows; a.tSetStrideRows = tStrideRows;
    a.qcounts = d_qcounts; a.tcounts = d_tcounts; a.pairs = d_pairs; a.swap = 0;
    a.nChunks = p.nChunks; a.chunkRows = p.chunkRows; a.partial = partial; a.maxRows = p.maxRows;
    // the second-best distance is only read by the ratio test (mode 1) and the raw output (mode 3); MatchByBruteForce
    // (src/FeatureMatching.cpp:30-54) and the cross-checks use the nearest neighbour alone
    const bool needSecond = mode == SSVIO_MODE_KNN2_RATIO || mode == 3;
    if (mode != SSVIO_MODE_CROSSCHECK_CV) {
        ProfScope ps(prof, "k_hamming_knn2", s);
        if (p.small) launch_knn2_small(a, maxQ, npairs, needSecond, s); else if (p.mfma) launch_knn2_mfma(a, maxQ, p.qBlocks, npairs, needSecond, s); else launch_knn2(a, maxQ, p.qBlocks, npairs, needSecond, s);
        SSVIO_HIP(hipGetLastError());
    }
    FinalizeArgs f{};
    f.partial = partial; f.nChunks = p.nChunks; f.maxRows = p.maxRows;
    f.qcounts = d_qcounts; f.tcounts = d_tcounts; f.pairs = d_pairs;
    f.mode = mode; f.threshold = m->threshold; f.nnRatio = m->nnRatio;
    f.out = out; f.nout = nout; f.raw = raw;
    if (mode == SSVIO_MODE_CROSSCHECK || mode == SSVIO_MODE_CROSSCHECK_CV) {
        MatchArgs r = a;
        r.qdesc = (const uint4*)d_tdesc; r.tdesc = (const uint4*)d_qdesc;
        r.qSetStrideRows = tStrideRows; r.tSetStrideRows = qStrideRows;
        r.qcounts = d_tcounts; r.tcounts = d_qcounts; r.swap = 1;
        r.nChunks = p.nChunksR; r.chunkRows = p.chunkRowsR; r.partial = partialRev;
        {
            ProfScope ps(prof, "k_hamming_knn2", s);
            if (p.smallRev) launch_knn2_small(r, maxT, npairs, false, s); else if (p.mfma) launch_knn2_mfma(r, maxT, p.qBlocksR, npairs, false, s); else launch_knn2(r, maxT, p.qBlocksR, npairs, false, s);
            SSVIO_HIP(hipGetLastError());
        }
        f.partialRev = partialRev; f.nChunksRev = p.nChunksR;
    }
    {
        ProfScope ps(prof, "k_match_finalize", s);
        launch_finalize(f, npairs, maxQ, maxT, s);
        SSVIO_HIP(hipGetLastError());
    }
    return SSVIO_OK;
}

// shared driver of the allocating entry points: descriptors on the device; pairs == nullptr means the single pair (0,0)
static int match_run(ssvio_matcher* m, const uint8_t* d_qdesc, int qStrideRows, const int* d_qcounts,
                     const uint8_t* d_tdesc, int tStrideRows, const int* d_tcounts,
                     const int* d_pairs, int npairs, int maxQ, int maxT, int mode,
                     ssvio_dmatch* outTo = nullptr, int* noutTo = nullptr, int32_t* rawTo = nullptr, int slack = 32)
{
    hipStream_t s = m->stream;
    MatchPlan p;
    int rc = match_plan(maxQ, maxT, npairs, mode, m->kernel, slack, 0, p); if (rc) return rc;
    rc = grow(m->d_partial, m->partialCap, p.partialEntries, s); if (rc) return rc;
    rc = grow(m->d_nout, m->noutCap, (size_t)npairs, s); if (rc) return rc;
    if (mode == 3) { rc = grow(m->d_raw, m->rawCap, (size_t)npairs * 3 * p.maxRows, s); if (rc) return rc; }
    else { rc = grow(m->d_out, m->outCap, (size_t)npairs * p.maxRows, s); if (rc) return rc; }
    if (p.partialRevEntries) { rc = grow(m->d_partialRev, m->partialRevCap, p.partialRevEntries, s); if (rc) return rc; }
    rc = match_launch(m, p, d_qdesc, qStrideRows, d_qcounts, d_tdesc, tStrideRows, d_tcounts, d_pairs, npairs, maxQ, maxT, mode,
                      m->d_partial, m->d_partialRev, outTo ? outTo : m->d_out, noutTo ? noutTo : m->d_nout, rawTo ? rawTo : m->d_raw, true);
    if (rc) return rc;
    m->lastPairs = npairs; m->lastMaxRows = p.maxRows; m->lastHadPairs = false;
    return SSVIO_OK;
}

// the batched matcher over a pair list that is already in m->d_pairs (capi_loop.hip builds it on the device)
int match_batch_on_device_pairs(ssvio_matcher* m, const uint8_t* d_desc, int strideRows, const int* d_counts, int npairs, int maxRows, int mode)
{
    const int rc = match_run(m, d_desc, strideRows, d_counts, d_desc, strideRows, d_counts, m->d_pairs, npairs, maxRows, maxRows, mode,
                             nullptr, nullptr, nullptr, /*slack=*/0);          // key-frame sets of a loop-candidate walk: as full as they say
    if (rc) return rc;
    m->lastHadPairs = true;
    return SSVIO_OK;
}

int stage_pair(ssvio_matcher* m, const uint8_t* q, int nq, int qstride, const uint8_t* t, int nt, int tstride, const uint8_t** d_q, const uint8_t** d_t)
{
    hipStream_t s = m->stream;
    static const bool noMapped = getenv("SSVIO_NO_MAPPED_INPUT") != nullptr;
    if (!noMapped && nq + nt <= 4096) {
        // small sets (a frame pair): the host packs the rows into mapped memory and the kernels read them from there (every row crosses
        // the bus a handful of times, which is still cheaper than two copy commands)
        const size_t toff = align256((size_t)nq * 32), need = toff + (size_t)nt * 32;
        if (need > m->descmCap) {
            SSVIO_HIP(hipStreamSynchronize(s));
            hfree(m->h_descm); m->descmCap = 0; m->d_descm = nullptr;
            SSVIO_HIP(hipHostMalloc((void**)&m->h_descm, 2 * need, hipHostMallocMapped));
            SSVIO_HIP(hipHostGetDevicePointer((void**)&m->d_descm, m->h_descm, 0));
            m->descmCap = 2 * need;
        }
        if (qstride == 32) memcpy(m->h_descm, q, (size_t)nq * 32); else for (int i = 0; i < nq; ++i) memcpy(m->h_descm + (size_t)i * 32, q + (size_t)i * qstride, 32);
        if (tstride == 32) memcpy(m->h_descm + toff, t, (size_t)nt * 32); else for (int i = 0; i < nt; ++i) memcpy(m->h_descm + toff + (size_t)i * 32, t + (size_t)i * tstride, 32);
        *d_q = m->d_descm; *d_t = m->d_descm + toff;
        return SSVIO_OK;
    }
    int rc = grow_rows(m->d_q, m->qRows, (size_t)nq, s); if (rc) return rc;
    rc = grow_rows(m->d_t, m->tRows, (size_t)nt, s); if (rc) return rc;
    rc = upload_rows(m->d_q, q, nq, qstride, m->stageQ, s); if (rc) return rc;
    rc = upload_rows(m->d_t, t, nt, tstride, m->stageT, s); if (rc) return rc;
    *d_q = m->d_q; *d_t = m->d_t;
    return SSVIO_OK;
}

int ensure_result(ssvio_matcher* m, size_t rowBytes)
{
    if (16 + rowBytes <= m->resultCap) return SSVIO_OK;
    SSVIO_HIP(hipStreamSynchronize(m->stream));
    hfree(m->h_result); m->resultCap = 0; m->d_resultm = nullptr;
    const size_t want = 16 + rowBytes + rowBytes / 4;
    SSVIO_HIP(hipHostMalloc((void**)&m->h_result, want, hipHostMallocMapped));
    SSVIO_HIP(hipHostGetDevicePointer((void**)&m->d_resultm, m->h_result, 0));
    m->resultCap = want;
    return SSVIO_OK;
}

// host-pointer single pair
static int match_host(ssvio_matcher* m, const uint8_t* q, int nq, int qstride, const uint8_t* t, int nt, int tstride, int mode,
                      ssvio_dmatch* out, int cap, int* n, int32_t* idx, int32_t* d1, int32_t* d2)
{
    if (mt_check(m)) return SSVIO_ERR_INVALID;
    if (nq < 0 || nt < 0) { set_error("negative row count"); return SSVIO_ERR_INVALID; }
    if (n) *n = 0;
    if (!(nq > 0 && nt > 0)) return SSVIO_OK;                          // src/FeatureMatching.cpp:25-27
    if (!q || !t) { set_error("null descriptor pointer"); return SSVIO_ERR_INVALID; }
    if (qstride < 32 || tstride < 32) { set_error("descriptor stride < 32"); return SSVIO_ERR_INVALID; }
    if (nt > (1 << 23) || nq > (1 << 23)) { set_error("more than 2^23 rows per set"); return SSVIO_ERR_INVALID; }
    if (mode != 3 && cap < nq) { set_error("output capacity %d < %d query rows", cap, nq); return SSVIO_ERR_CAPACITY; }
    SSVIO_HIP(hipSetDevice(m->device));
    hipStream_t s = m->stream;
    const uint8_t* d_qrows; const uint8_t* d_trows;
    int rc = stage_pair(m, q, nq, qstride, t, nt, tstride, &d_qrows, &d_trows); if (rc) return rc;
    // one wait per call and no copy commands: the kernels read the counts from, and write the count and the rows to, mapped host memory
    const int maxRows = std::max(nq, nt);
    const size_t rowBytes = mode == 3 ? (size_t)3 * maxRows * sizeof(int32_t) : (size_t)maxRows * sizeof(ssvio_dmatch);
    rc = ensure_result(m, rowBytes); if (rc) return rc;
    m->h_counts2[0] = nq; m->h_counts2[1] = nt;
    *(volatile int*)m->h_result = 0;
    rc = match_run(m, d_qrows, 0, m->d_counts2m, d_trows, 0, m->d_counts2m + 1, nullptr, 1, nq, nt, mode,
                   (ssvio_dmatch*)(m->d_resultm + 16), (int*)m->d_resultm, (int32_t*)(m->d_resultm + 16), /*slack=*/0);
    if (rc) return rc;
    SSVIO_HIP(hipStreamSynchronize(s));
    if (mode == 3) {
        const int32_t* raw = (const int32_t*)(m->h_result + 16);
        for (int i = 0; i < nq; ++i) { if (idx) idx[i] = raw[i]; if (d1) d1[i] = raw[maxRows + i]; if (d2) d2[i] = raw[2 * (size_t)maxRows + i]; }
    } else {
        const int got = *(const volatile int*)m->h_result;
        const int cnt = std::min(std::max(got, 0), nq);
        if (cnt > 0) memcpy(out, m->h_result + 16, (size_t)cnt * sizeof(ssvio_dmatch));
        if (n) *n = cnt;
    }
    m->prof.resolve();
    return SSVIO_OK;
}

extern "C" {

int ssvio_matcher_create(int threshold, float nn_ratio, int device, ssvio_matcher** out)
{
    if (!out) { set_error("null argument"); return SSVIO_ERR_INVALID; }
    *out = nullptr;
    const int rc = pick_device(device);
    if (rc) return rc;
    SSVIO_HIP(hipSetDevice(device));
    ssvio_matcher* m = new ssvio_matcher();
    m->device = device; m->threshold = threshold; m->nnRatio = nn_ratio;
    { const char* e = getenv("SSVIO_MATCH_KERNEL"); m->kernel = (e && strcmp(e, "mfma") == 0) ? SSVIO_KERNEL_MFMA : (e && strcmp(e, "mfma_all") == 0) ? SSVIO_KERNEL_MFMA_ALL : SSVIO_KERNEL_VALU; }
    if (hipStreamCreateWithFlags(&m->own, hipStreamNonBlocking) != hipSuccess || hipMalloc(&m->d_counts2, 2 * sizeof(int)) != hipSuccess ||
        hipHostMalloc((void**)&m->h_counts2, 2 * sizeof(int), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void**)&m->d_counts2m, m->h_counts2, 0) != hipSuccess) {
        set_error("cannot create HIP stream / allocate");
        if (m->own) (void)hipStreamDestroy(m->own);
        dfree(m->d_counts2); hfree(m->h_counts2);
        delete m; return SSVIO_ERR_HIP;
    }
    m->stream = m->own;
    const int rs = noalloc_scratch_init(device);          // fixed scratch of ssvio_match_batch_dev_noalloc, once per device
    if (rs) { ssvio_matcher_destroy(m); return rs; }
    *out = m;
    return SSVIO_OK;
}

void ssvio_matcher_destroy(ssvio_matcher* m)
{
    if (!m) return;
    (void)hipSetDevice(m->device);
    // a borrowed stream (ssvio_matcher_set_stream) may already be gone -- its owner, typically the extractor, is often destroyed
    // first: wait for the device instead of touching it
    if (m->stream == m->own) (void)hipStreamSynchronize(m->stream); else (void)hipDeviceSynchronize();
    m->prof.resolve(); m->prof.destroy();
    dfree(m->d_q); dfree(m->d_t); dfree(m->d_counts2); dfree(m->d_partial); dfree(m->d_partialRev);
    dfree(m->d_out); dfree(m->d_raw); dfree(m->d_nout); dfree(m->d_pairs);
    dfree(m->d_p3d); dfree(m->d_p2d); dfree(m->d_cand); dfree(m->d_keep);
    dfree(m->d_bowWord); dfree(m->d_bowNode); dfree(m->d_bowWeight); dfree(m->d_gnode); dfree(m->d_keepNode); dfree(m->d_sortIdx); dfree(m->d_sortNode);
    dfree(m->d_frameIn); dfree(m->d_frameOut); dfree(m->d_table);
    dfree(m->d_mapA); dfree(m->d_mapB); dfree(m->d_voxKeys); dfree(m->d_voxTmp); dfree(m->d_voxState); hfree(m->h_voxTotal);
    hfree(m->h_counts2); hfree(m->h_result); hfree(m->h_descm); hfree(m->h_bowm); hfree(m->h_fin); hfree(m->h_fout);
    if (m->own) (void)hipStreamDestroy(m->own);
    delete m;
}

int ssvio_matcher_set_stream(ssvio_matcher* m, void* hip_stream)
{
    if (mt_check(m)) return SSVIO_ERR_INVALID;
    SSVIO_HIP(hipSetDevice(m->device));
    SSVIO_HIP(hipStreamSynchronize(m->stream));
    m->stream = hip_stream ? (hipStream_t)hip_stream : m->own;
    return SSVIO_OK;
}

int ssvio_matcher_set_kernel(ssvio_matcher* m, int kernel)
{
    if (mt_check(m)) return SSVIO_ERR_INVALID;
    if (kernel != SSVIO_KERNEL_VALU && kernel != SSVIO_KERNEL_MFMA && kernel != SSVIO_KERNEL_MFMA_ALL) { set_error("kernel must be SSVIO_KERNEL_VALU, _MFMA or _MFMA_ALL"); return SSVIO_ERR_INVALID; }
    m->kernel = kernel;
    return SSVIO_OK;
}

int ssvio_matcher_get_kernel(ssvio_matcher* m, int* kernel)
{
    if (mt_check(m)) return SSVIO_ERR_INVALID;
    if (!kernel) { set_error("null argument"); return SSVIO_ERR_INVALID; }
    *kernel = m->kernel;
    return SSVIO_OK;
}

int ssvio_matcher_profile_only(ssvio_matcher* m, const char* kernel)
{
    if (mt_check(m)) return SSVIO_ERR_INVALID;
    m->profOnly = kernel ? kernel : "";
    m->prof.only = m->profOnly.empty() ? nullptr : m->profOnly.c_str();
    return SSVIO_OK;
}

int ssvio_matcher_set_profiling(ssvio_matcher* m, int enable)
{
    if (mt_check(m)) return SSVIO_ERR_INVALID;
    SSVIO_HIP(hipSetDevice(m->device));
    SSVIO_HIP(hipStreamSynchronize(m->stream));
    m->prof.clear();
    m->prof.enabled = enable != 0;
    return SSVIO_OK;
}

int ssvio_matcher_kernel_times(ssvio_matcher* m, const char** names, double* total_ms, int32_t* launches, int cap, int* count)
{
    if (mt_check(m)) return SSVIO_ERR_INVALID;
    SSVIO_HIP(hipSetDevice(m->device));
    SSVIO_HIP(hipStreamSynchronize(m->stream));
    return prof_out(m->prof, names, total_ms, launches, cap, count);
}

int ssvio_match_bruteforce(ssvio_matcher* m, const uint8_t* q, int nq, int qstride, const uint8_t* t, int nt, int tstride,
                           ssvio_dmatch* out, int cap, int* n)
{ return match_host(m, q, nq, qstride, t, nt, tstride, SSVIO_MODE_BRUTEFORCE, out, cap, n, nullptr, nullptr, nullptr); }

int ssvio_match_knn2_ratio(ssvio_matcher* m, const uint8_t* q, int nq, int qstride, const uint8_t* t, int nt, int tstride,
                           ssvio_dmatch* out, int cap, int* n)
{ return match_host(m, q, nq, qstride, t, nt, tstride, SSVIO_MODE_KNN2_RATIO, out, cap, n, nullptr, nullptr, nullptr); }

int ssvio_match_crosscheck(ssvio_matcher* m, const uint8_t* q, int nq, int qstride, const uint8_t* t, int nt, int tstride,
                           ssvio_dmatch* out, int cap, int* n)
{ return match_host(m, q, nq, qstride, t, nt, tstride, SSVIO_MODE_CROSSCHECK, out, cap, n, nullptr, nullptr, nullptr); }

int ssvio_match_crosscheck_cv(ssvio_matcher* m, const uint8_t* q, int nq, int qstride, const uint8_t* t, int nt, int tstride,
                              ssvio_dmatch* out, int cap, int* n)
{ return match_host(m, q, nq, qstride, t, nt, tstride, SSVIO_MODE_CROSSCHECK_CV, out, cap, n, nullptr, nullptr, nullptr); }

int ssvio_match_knn2_raw(ssvio_matcher* m, const uint8_t* q, int nq, int qstride, const uint8_t* t, int nt, int tstride,
                         int32_t* idx, int32_t* d1, int32_t* d2)
{ return match_host(m, q, nq, qstride, t, nt, tstride, 3, nullptr, 0, nullptr, idx, d1, d2); }

int ssvio_match_project(ssvio_matcher* m, const double* p3d, const uint8_t* q, int nq, int qstride,
                        const double* p2d, const uint8_t* t, int nt, int tstride, double radian,
                        const double* K, const double* R, const double* tr, ssvio_dmatch* out, int cap, int* n)
{
    if (mt_check(m)) return SSVIO_ERR_INVALID;
    if (n) *n = 0;
    if (nq < 0 || nt < 0) { set_error("negative count"); return SSVIO_ERR_INVALID; }
    if (nq == 0) return SSVIO_OK;                                   // the loop of src/FeatureMatching.cpp:135 does not run
    if (!p3d || !q || !K || !R || !tr || !n || (nt > 0 && (!p2d || !t))) { set_error("null argument"); return SSVIO_ERR_INVALID; }
    if (qstride < 32 || (nt > 0 && tstride < 32)) { set_error("descriptor stride < 32"); return SSVIO_ERR_INVALID; }
    if (nt > (1 << 23)) { set_error("more than 2^23 rows per set"); return SSVIO_ERR_INVALID; }
    if (cap < nq) { set_error("output capacity %d < %d reference points", cap, nq); return SSVIO_ERR_CAPACITY; }
    SSVIO_HIP(hipSetDevice(m->device));
    hipStream_t s = m->stream;
    int rc = grow(m->d_cand, m->candCap, (size_t)nq, s); if (rc) return rc;
    rc = grow(m->d_keep, m->keepCap, (size_t)nq, s); if (rc) return rc;
    // every wave scans all nt projected points and many train rows: inputs are copied to HBM (reading them across the bus once per
    // wave would cost more than the copies); the match list is stored straight to mapped host memory, so the call waits once
    rc = grow_rows(m->d_q, m->qRows, (size_t)nq, s); if (rc) return rc;
    rc = grow_rows(m->d_t, m->tRows, (size_t)std::max(nt, 1), s); if (rc) return rc;
    rc = grow(m->d_p3d, m->p3dCap, (size_t)nq * 3, s); if (rc) return rc;
    rc = grow(m->d_p2d, m->p2dCap, (size_t)std::max(nt, 1) * 2, s); if (rc) return rc;
    rc = upload_rows(m->d_q, q, nq, qstride, m->stageQ, s); if (rc) return rc;
    SSVIO_HIP(hipMemcpyAsync(m->d_p3d, p3d, (size_t)nq * 3 * sizeof(double), hipMemcpyHostToDevice, s));
    if (nt > 0) {
        rc = upload_rows(m->d_t, t, nt, tstride, m->stageT, s); if (rc) return rc;
        SSVIO_HIP(hipMemcpyAsync(m->d_p2d, p2d, (size_t)nt * 2 * sizeof(double), hipMemcpyHostToDevice, s));
    }
    const uint8_t* d_q = m->d_q; const uint8_t* d_t = m->d_t;
    const double* d_p3 = m->d_p3d; const double* d_p2 = m->d_p2d;
    rc = ensure_result(m, (size_t)nq * sizeof(ssvio_dmatch)); if (rc) return rc;
    *(volatile int*)m->h_result = 0;
    ProjectArgs a{};
    a.p3d = d_p3; a.qdesc = (const uint4*)d_q; a.nq = nq; a.p2d = d_p2; a.tdesc = (const uint4*)d_t; a.nt = nt;
    a.radian = radian;
    for (int i = 0; i < 9; ++i) { a.K[i] = K[i]; a.R[i] = R[i]; }
    for (int i = 0; i < 3; ++i) a.t[i] = tr[i];
    a.threshold = m->threshold; a.nnRatio = m->nnRatio;
    a.cand = m->d_cand; a.keep = m->d_keep; a.out = (ssvio_dmatch*)(m->d_resultm + 16); a.nout = (int*)m->d_resultm;
    {
        ProfScope p(m->prof, "k_match_project", s);
        launch_project(a, s);
        SSVIO_HIP(hipGetLastError());
    }
    SSVIO_HIP(hipStreamSynchronize(s));
    const int got = *(const volatile int*)m->h_result;
    const int cnt = std::min(std::max(got, 0), nq);
    if (cnt > 0) memcpy(out, m->h_result + 16, (size_t)cnt * sizeof(ssvio_dmatch));
    *n = cnt;
    m->prof.resolve();
    return SSVIO_OK;
}

int ssvio_match_batch_dev(ssvio_matcher* m, const void* d_desc, int set_stride_rows, int max_rows, const int32_t* d_counts,
                          const int32_t* pairs, int npairs, int mode, const void** d_matches, const int32_t** d_nmatches)
{
    if (mt_check(m)) return SSVIO_ERR_INVALID;
    if (!d_desc || !d_counts || !pairs || npairs <= 0 || max_rows <= 0 || set_stride_rows < max_rows) { set_error("bad batch arguments"); return SSVIO_ERR_INVALID; }
    if (mode < 0 || mode == 3 || mode > SSVIO_MODE_CROSSCHECK_CV) { set_error("mode must be SSVIO_MODE_BRUTEFORCE, _KNN2_RATIO, _CROSSCHECK or _CROSSCHECK_CV"); return SSVIO_ERR_INVALID; }
    if (max_rows > (1 << 23)) { set_error("more than 2^23 rows per set"); return SSVIO_ERR_INVALID; }
    SSVIO_HIP(hipSetDevice(m->device));
    int rc = SSVIO_OK;
    rc = upload_pairs(m, pairs, npairs); if (rc) return rc;
    rc = match_run(m, (const uint8_t*)d_desc, set_stride_rows, d_counts, (const uint8_t*)d_desc, set_stride_rows, d_counts,
                   m->d_pairs, npairs, max_rows, max_rows, mode);
    if (rc) return rc;
    m->lastHadPairs = true;
    if (d_matches) *d_matches = m->d_out;
    if (d_nmatches) *d_nmatches = m->d_nout;
    return SSVIO_OK;
}

int ssvio_match_batch_summary_dev(ssvio_matcher* m, int id_mode, int id_base, void* d_records)
{
    if (mt_check(m)) return SSVIO_ERR_INVALID;
    if (!d_records || (id_mode != SSVIO_PAIR_ID_SEQUENTIAL && id_mode != SSVIO_PAIR_ID_TRIANGULAR)) { set_error("bad summary arguments"); return SSVIO_ERR_INVALID; }
    if (m->lastPairs <= 0 || !m->d_out || !m->d_nout || !m->lastHadPairs) {
        set_error("no batched result with a pair list to summarise"); return SSVIO_ERR_INVALID;
    }
    SSVIO_HIP(hipSetDevice(m->device));
    ProfScope p(m->prof, "k_pair_summary", m->stream);
    launch_pair_summary(m->d_out, m->d_nout, m->lastMaxRows, m->d_pairs, m->lastPairs, id_mode, id_base, (ssvio_pair_summary*)d_records, m->stream);
    SSVIO_HIP(hipGetLastError());
    return SSVIO_OK;
}

int ssvio_matcher_sync(ssvio_matcher* m)
{
    if (mt_check(m)) return SSVIO_ERR_INVALID;
    SSVIO_HIP(hipSetDevice(m->device));
    SSVIO_HIP(hipStreamSynchronize(m->stream));
    m->prof.resolve();
    return SSVIO_OK;
}

int ssvio_match_batch_fetch(ssvio_matcher* m, int npairs, int max_rows, ssvio_dmatch* out, int32_t* n)
{
    if (mt_check(m)) return SSVIO_ERR_INVALID;
    if (npairs != m->lastPairs || max_rows != m->lastMaxRows || !n) { set_error("fetch does not match the last batch call"); return SSVIO_ERR_INVALID; }
    SSVIO_HIP(hipSetDevice(m->device));
    SSVIO_HIP(hipMemcpyAsync(n, m->d_nout, (size_t)npairs * sizeof(int), hipMemcpyDeviceToHost, m->stream));
    if (out) SSVIO_HIP(hipMemcpyAsync(out, m->d_out, (size_t)npairs * max_rows * sizeof(ssvio_dmatch), hipMemcpyDeviceToHost, m->stream));
    SSVIO_HIP(hipStreamSynchronize(m->stream));
    return SSVIO_OK;
}

}  // extern "C"
