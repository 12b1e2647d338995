// capi_frame.hip -- the C ABI of include/ssvio_gpu.h, part 4: the steps behind the matcher.
#include "capi_internal.h"

// =====================================================================================
// behind the matcher
static int cam_dev(const ssvio_camera* cam, int w, int h, int stride_px, FrameCam& c)
{
    if (!cam) { set_error("null camera"); return SSVIO_ERR_INVALID; }
    if (w <= 0 || h <= 0 || stride_px < w) { set_error("bad depth image geometry %dx%d stride %d", w, h, stride_px); return SSVIO_ERR_INVALID; }
    c.cx = cam->cx; c.cy = cam->cy;
    c.invFx = 1 / cam->fx; c.invFy = 1 / cam->fy;                    // include/Init.h:19-20
    c.depthScale = cam->depth_scale; c.w = w; c.h = h; c.stridePx = stride_px;
    return SSVIO_OK;
}


// carve the assembled-list block into its seven arrays + counts: in HBM (batched, device-resident form) or, for the host entry
// point, in mapped host memory that k_assemble stores to directly (`host` then receives the host view of the same block)
static int assemble_outputs(ssvio_matcher* m, int npairs, int maxRows, AssembleArgs& a, uint8_t** host = nullptr)
{
    const size_t rows = (size_t)npairs * maxRows;
    const size_t s3 = align256(rows * 12), s2 = align256(rows * 8), sm = align256(rows * 16), sc = align256((size_t)npairs * 8);
    int rc;
    uint8_t* p;
    if (host) { rc = grow_mapped(m->h_fout, m->d_fout, m->foutCap, 4 * s3 + s2 + 2 * sm + sc, m->stream); if (rc) return rc; p = m->d_fout; *host = m->h_fout; }
    else { rc = grow(m->d_frameOut, m->frameOutCap, 4 * s3 + s2 + 2 * sm + sc, m->stream); if (rc) return rc; p = m->d_frameOut; }
    a.last3d = (float*)p; p += s3; a.cur3d = (float*)p; p += s3; a.edgeLast3d = (float*)p; p += s3; a.edgeCur3d = (float*)p; p += s3;
    a.cur2d = (float*)p; p += s2; a.preGood = (ssvio_dmatch*)p; p += sm; a.finalGood = (ssvio_dmatch*)p; p += sm; a.counts = (int*)p;
    return SSVIO_OK;
}

extern "C" {

int ssvio_get_3d_points(ssvio_matcher* m, const ssvio_keypoint* kps, int n, const uint16_t* depth, int w, int h, int stride_px,
                        const ssvio_camera* cam, float* out)
{
    if (mt_check(m)) return SSVIO_ERR_INVALID;
    FrameCam c; int rc = cam_dev(cam, w, h, stride_px, c); if (rc) return rc;
    if (n < 0) { set_error("negative count"); return SSVIO_ERR_INVALID; }
    if (n == 0) return SSVIO_OK;
    if (!kps || !depth || !out) { set_error("null argument"); return SSVIO_ERR_INVALID; }
    SSVIO_HIP(hipSetDevice(m->device));
    hipStream_t s = m->stream;
    const size_t kb = align256((size_t)n * sizeof(ssvio_keypoint)), db = align256((size_t)h * stride_px * 2);
    rc = grow(m->d_frameIn, m->frameInCap, kb + db, s); if (rc) return rc;
    rc = grow(m->d_frameOut, m->frameOutCap, (size_t)n * 12, s); if (rc) return rc;
    SSVIO_HIP(hipMemcpyAsync(m->d_frameIn, kps, (size_t)n * sizeof(ssvio_keypoint), hipMemcpyHostToDevice, s));
    SSVIO_HIP(hipMemcpyAsync(m->d_frameIn + kb, depth, (size_t)h * stride_px * 2, hipMemcpyHostToDevice, s));
    {
        ProfScope p(m->prof, "k_get_3d_points", s);
        launch_get_3d_points((const ssvio_keypoint*)m->d_frameIn, n, (const uint16_t*)(m->d_frameIn + kb), c, (float*)m->d_frameOut, s);
        SSVIO_HIP(hipGetLastError());
    }
    SSVIO_HIP(hipMemcpyAsync(out, m->d_frameOut, (size_t)n * 12, hipMemcpyDeviceToHost, s));
    SSVIO_HIP(hipStreamSynchronize(s));
    m->prof.resolve();
    return SSVIO_OK;
}

int ssvio_assemble_correspondences(ssvio_matcher* m, const ssvio_dmatch* matches, int nm,
                                   const ssvio_keypoint* last_kps, int nlast, const uint16_t* last_depth,
                                   const ssvio_keypoint* cur_kps, int ncur, const uint16_t* cur_depth,
                                   int w, int h, int stride_px, const ssvio_camera* cam,
                                   float* last3d, float* cur2d, float* cur3d, ssvio_dmatch* pre_good, int* n_pre,
                                   float* edge_last3d, float* edge_cur3d, ssvio_dmatch* final_good, int* n_final)
{
    if (mt_check(m)) return SSVIO_ERR_INVALID;
    FrameCam c; int rc = cam_dev(cam, w, h, stride_px, c); if (rc) return rc;
    if (n_pre) *n_pre = 0;
    if (n_final) *n_final = 0;
    if (nm < 0 || nlast < 0 || ncur < 0) { set_error("negative count"); return SSVIO_ERR_INVALID; }
    if (nm == 0) return SSVIO_OK;                                   // the loop of src/Odometry.cpp:44 does not run
    if (!matches || !last_kps || !cur_kps || !last_depth || !cur_depth || !last3d || !cur2d || !cur3d || !pre_good || !n_pre ||
        !edge_last3d || !edge_cur3d || !final_good || !n_final) { set_error("null argument"); return SSVIO_ERR_INVALID; }
    for (int i = 0; i < nm; ++i)
        if (matches[i].trainIdx < 0 || matches[i].trainIdx >= nlast || matches[i].queryIdx < 0 || matches[i].queryIdx >= ncur) {
            set_error("match %d (query %d, train %d) is outside the keypoint lists (%d, %d)", i, matches[i].queryIdx, matches[i].trainIdx, ncur, nlast);
            return SSVIO_ERR_INVALID;
        }
    SSVIO_HIP(hipSetDevice(m->device));
    hipStream_t s = m->stream;
    // matches, keypoints and counts go into mapped host memory (read across the bus by one workgroup), the two depth planes are copied
    // (only ~nm pixels of each are read, but they sit in pageable memory), all seven lists come back through mapped memory: one wait
    const int kpStride = std::max(std::max(nlast, ncur), 1);
    const size_t depthPx = (size_t)h * stride_px;
    const size_t mb = align256((size_t)nm * 16), kb = align256((size_t)2 * kpStride * sizeof(ssvio_keypoint)), db = align256(2 * depthPx * 2);
    rc = grow_mapped(m->h_fin, m->d_fin, m->finCap, mb + kb + 256, s); if (rc) return rc;
    rc = grow(m->d_frameIn, m->frameInCap, db, s); if (rc) return rc;
    memcpy(m->h_fin, matches, (size_t)nm * 16);
    if (ncur) memcpy(m->h_fin + mb, cur_kps, (size_t)ncur * sizeof(ssvio_keypoint));                                           // set 0 = current
    if (nlast) memcpy(m->h_fin + mb + (size_t)kpStride * sizeof(ssvio_keypoint), last_kps, (size_t)nlast * sizeof(ssvio_keypoint));
    int* cnts = (int*)(m->h_fin + mb + kb); cnts[0] = nm; cnts[1] = ncur; cnts[2] = nlast;
    uint8_t* d_d = m->d_frameIn;
    SSVIO_HIP(hipMemcpyAsync(d_d, cur_depth, depthPx * 2, hipMemcpyHostToDevice, s));
    SSVIO_HIP(hipMemcpyAsync(d_d + depthPx * 2, last_depth, depthPx * 2, hipMemcpyHostToDevice, s));
    const int* d_c = (const int*)(m->d_fin + mb + kb);
    AssembleArgs a{};
    a.matches = (const ssvio_dmatch*)m->d_fin; a.nmatches = d_c; a.maxRows = nm; a.pairs = nullptr;
    a.kps = (const ssvio_keypoint*)(m->d_fin + mb); a.kpStride = kpStride; a.kpCounts = d_c + 1;
    a.depth = (const uint16_t*)d_d; a.depthFrameStride = depthPx; a.cam = c;
    uint8_t* hostOut = nullptr;
    rc = assemble_outputs(m, 1, nm, a, &hostOut); if (rc) return rc;
    {
        ProfScope p(m->prof, "k_assemble", s);
        launch_assemble(a, 1, s);
        SSVIO_HIP(hipGetLastError());
    }
    SSVIO_HIP(hipStreamSynchronize(s));
    auto hv = [&](const void* dev) { return hostOut + ((const uint8_t*)dev - m->d_fout); };          // host view of a carved array
    const int* got = (const int*)hv(a.counts);
    const size_t np = (size_t)std::min(std::max(got[0], 0), nm), nf = (size_t)std::min(std::max(got[1], 0), nm);
    memcpy(last3d, hv(a.last3d), np * 12); memcpy(cur2d, hv(a.cur2d), np * 8); memcpy(cur3d, hv(a.cur3d), np * 12); memcpy(pre_good, hv(a.preGood), np * 16);
    memcpy(edge_last3d, hv(a.edgeLast3d), nf * 12); memcpy(edge_cur3d, hv(a.edgeCur3d), nf * 12); memcpy(final_good, hv(a.finalGood), nf * 16);
    *n_pre = (int)np; *n_final = (int)nf;
    m->prof.resolve();
    return SSVIO_OK;
}

int ssvio_assemble_batch_dev(ssvio_matcher* m, const void* d_matches, const int32_t* d_nmatches, int max_rows,
                             const int32_t* pairs, int npairs, const void* d_kps, int kp_stride, const int32_t* d_kpcounts,
                             const void* d_depth, size_t depth_frame_stride, int w, int h, int stride_px,
                             const ssvio_camera* cam, ssvio_correspondences_dev* out)
{
    if (mt_check(m)) return SSVIO_ERR_INVALID;
    FrameCam c; int rc = cam_dev(cam, w, h, stride_px, c); if (rc) return rc;
    if (!d_matches || !d_nmatches || !pairs || !d_kps || !d_kpcounts || !d_depth || !out || npairs <= 0 || max_rows <= 0 || kp_stride <= 0 ||
        depth_frame_stride < (size_t)h * stride_px) { set_error("bad batch arguments"); return SSVIO_ERR_INVALID; }
    SSVIO_HIP(hipSetDevice(m->device));
    rc = upload_pairs(m, pairs, npairs); if (rc) return rc;
    AssembleArgs a{};
    a.matches = (const ssvio_dmatch*)d_matches; a.nmatches = d_nmatches; a.maxRows = max_rows; a.pairs = m->d_pairs;
    a.kps = (const ssvio_keypoint*)d_kps; a.kpStride = kp_stride; a.kpCounts = d_kpcounts;
    a.depth = (const uint16_t*)d_depth; a.depthFrameStride = depth_frame_stride; a.cam = c;
    rc = assemble_outputs(m, npairs, max_rows, a); if (rc) return rc;
    {
        ProfScope p(m->prof, "k_assemble", m->stream);
        launch_assemble(a, npairs, m->stream);
        SSVIO_HIP(hipGetLastError());
    }
    out->last3d = a.last3d; out->cur2d = a.cur2d; out->cur3d = a.cur3d; out->pre_good = a.preGood;
    out->edge_last3d = a.edgeLast3d; out->edge_cur3d = a.edgeCur3d; out->final_good = a.finalGood; out->counts = a.counts;
    return SSVIO_OK;
}

// index map over a DMatch list on the device: table[key] = first position (0xFFFFFFFF = none)
static int first_index_table(ssvio_matcher* m, const ssvio_dmatch* d_v, int n, int useTrain, int tableSize)
{
    hipStream_t s = m->stream;
    int rc = grow(m->d_table, m->tableCap, (size_t)std::max(tableSize, 1), s); if (rc) return rc;
    SSVIO_HIP(hipMemsetAsync(m->d_table, 0xFF, (size_t)std::max(tableSize, 1) * 4, s));
    ProfScope p(m->prof, "k_first_index", s);
    launch_first_index((const int32_t*)d_v, n, 4, useTrain ? 1 : 0, m->d_table, tableSize, s);
    SSVIO_HIP(hipGetLastError());
    return SSVIO_OK;
}

int ssvio_track_map_points(ssvio_matcher* m, const ssvio_dmatch* last, int nlast, const ssvio_dmatch* cur, int ncur,
                           ssvio_dmatch* out, int cap, int* n)
{
    if (mt_check(m)) return SSVIO_ERR_INVALID;
    if (n) *n = 0;
    if (nlast < 0 || ncur < 0) { set_error("negative count"); return SSVIO_ERR_INVALID; }
    if (nlast == 0 || ncur == 0) return SSVIO_OK;                    // no pair can meet: last_match_vec becomes empty (src/Map.cpp:121-122)
    if (!last || !cur || !out || !n) { set_error("null argument"); return SSVIO_ERR_INVALID; }
    if (cap < nlast) { set_error("output capacity %d < %d", cap, nlast); return SSVIO_ERR_CAPACITY; }
    int maxKey = -1;                                                 // the map covers the keys the join can ask for
    for (int i = 0; i < nlast; ++i) maxKey = std::max(maxKey, last[i].queryIdx);
    if (maxKey < 0) return SSVIO_OK;
    if (maxKey >= (1 << 28)) { set_error("queryIdx %d too large for an index map", maxKey); return SSVIO_ERR_INVALID; }
    SSVIO_HIP(hipSetDevice(m->device));
    hipStream_t s = m->stream;
    const size_t lb = align256((size_t)nlast * 16), cb = align256((size_t)ncur * 16);
    int rc;
    if (maxKey + 1 <= kTrackLdsTable && nlast <= 8192 && ncur <= 8192) {
        // the per-frame case: both lists in, the new list and its count out through mapped host memory, the index map in LDS --
        // one launch, no copy commands, one wait
        rc = grow_mapped(m->h_fin, m->d_fin, m->finCap, lb + cb, s); if (rc) return rc;
        rc = grow_mapped(m->h_fout, m->d_fout, m->foutCap, 16 + (size_t)nlast * 16, s); if (rc) return rc;
        memcpy(m->h_fin, last, (size_t)nlast * 16); memcpy(m->h_fin + lb, cur, (size_t)ncur * 16);
        *(volatile int*)m->h_fout = 0;
        {
            ProfScope p(m->prof, "k_track", s);
            launch_track((const ssvio_dmatch*)m->d_fin, nlast, (const ssvio_dmatch*)(m->d_fin + lb), ncur, nullptr, maxKey + 1,
                         (ssvio_dmatch*)(m->d_fout + 16), (int*)m->d_fout, s);
            SSVIO_HIP(hipGetLastError());
        }
        SSVIO_HIP(hipStreamSynchronize(s));
        const int got = *(const volatile int*)m->h_fout;
        const int cnt = std::min(std::max(got, 0), nlast);
        if (cnt > 0) memcpy(out, m->h_fout + 16, (size_t)cnt * sizeof(ssvio_dmatch));
        *n = cnt;
        m->prof.resolve();
        return SSVIO_OK;
    }
    rc = grow(m->d_frameIn, m->frameInCap, lb + cb, s); if (rc) return rc;
    rc = grow(m->d_out, m->outCap, (size_t)nlast, s); if (rc) return rc;
    rc = grow(m->d_nout, m->noutCap, (size_t)1, s); if (rc) return rc;
    SSVIO_HIP(hipMemcpyAsync(m->d_frameIn, last, (size_t)nlast * 16, hipMemcpyHostToDevice, s));
    SSVIO_HIP(hipMemcpyAsync(m->d_frameIn + lb, cur, (size_t)ncur * 16, hipMemcpyHostToDevice, s));
    rc = first_index_table(m, (const ssvio_dmatch*)(m->d_frameIn + lb), ncur, 1, maxKey + 1); if (rc) return rc;
    {
        ProfScope p(m->prof, "k_track", s);
        launch_track((const ssvio_dmatch*)m->d_frameIn, nlast, (const ssvio_dmatch*)(m->d_frameIn + lb), ncur, m->d_table, maxKey + 1, m->d_out, m->d_nout, s);
        SSVIO_HIP(hipGetLastError());
    }
    int cnt = 0;
    SSVIO_HIP(hipMemcpyAsync(&cnt, m->d_nout, sizeof(int), hipMemcpyDeviceToHost, s));
    SSVIO_HIP(hipStreamSynchronize(s));
    if (cnt > 0) SSVIO_HIP(hipMemcpy(out, m->d_out, (size_t)cnt * sizeof(ssvio_dmatch), hipMemcpyDeviceToHost));
    *n = cnt;
    m->prof.resolve();
    return SSVIO_OK;
}

int ssvio_in_match_vec(ssvio_matcher* m, const ssvio_dmatch* vec, int n, int nkp, int32_t* out)
{
    if (mt_check(m)) return SSVIO_ERR_INVALID;
    if (n < 0 || nkp < 0) { set_error("negative count"); return SSVIO_ERR_INVALID; }
    if (nkp == 0) return SSVIO_OK;
    if (!out || (n > 0 && !vec)) { set_error("null argument"); return SSVIO_ERR_INVALID; }
    SSVIO_HIP(hipSetDevice(m->device));
    hipStream_t s = m->stream;
    int rc = grow(m->d_frameIn, m->frameInCap, (size_t)std::max(n, 1) * 16, s); if (rc) return rc;
    if (n) SSVIO_HIP(hipMemcpyAsync(m->d_frameIn, vec, (size_t)n * 16, hipMemcpyHostToDevice, s));
    rc = first_index_table(m, (const ssvio_dmatch*)m->d_frameIn, n, 0, nkp); if (rc) return rc;
    SSVIO_HIP(hipMemcpyAsync(out, m->d_table, (size_t)nkp * 4, hipMemcpyDeviceToHost, s));    // 0xFFFFFFFF reads as -1
    SSVIO_HIP(hipStreamSynchronize(s));
    m->prof.resolve();
    return SSVIO_OK;
}

int ssvio_manage_map_points(ssvio_matcher* m, const ssvio_keypoint* kps, int nkp, const uint16_t* depth, int w, int h, int stride_px,
                            const ssvio_camera* cam, const float* world_trans, const ssvio_dmatch* last_match_vec, int nm,
                            const int32_t* obs, int nobs, int32_t* match_num, int32_t* map_point, float* world)
{
    if (mt_check(m)) return SSVIO_ERR_INVALID;
    FrameCam c; int rc = cam_dev(cam, w, h, stride_px, c); if (rc) return rc;
    if (nkp < 0 || nm < 0 || nobs < 0) { set_error("negative count"); return SSVIO_ERR_INVALID; }
    if (nkp == 0) return SSVIO_OK;                                   // the loop of src/Map.cpp:128 does not run
    if (!kps || !depth || !world_trans || !match_num || !map_point || !world || (nm > 0 && !last_match_vec) || (nobs > 0 && !obs)) {
        set_error("null argument"); return SSVIO_ERR_INVALID;
    }
    SSVIO_HIP(hipSetDevice(m->device));
    hipStream_t s = m->stream;
    // keypoints, matches and observation pairs go into mapped host memory, the depth plane is copied, the three per-keypoint arrays
    // come back through mapped memory: one wait
    const size_t kb = align256((size_t)nkp * sizeof(ssvio_keypoint)), db = align256((size_t)h * stride_px * 2),
                 mb = align256((size_t)std::max(nm, 1) * 16), ob = align256((size_t)std::max(nobs, 1) * 8);
    const size_t ib = align256((size_t)nkp * 4);
    rc = grow_mapped(m->h_fin, m->d_fin, m->finCap, kb + mb + ob, s); if (rc) return rc;
    rc = grow_mapped(m->h_fout, m->d_fout, m->foutCap, 2 * ib + (size_t)nkp * 12, s); if (rc) return rc;
    rc = grow(m->d_frameIn, m->frameInCap, db, s); if (rc) return rc;
    rc = grow(m->d_table, m->tableCap, (size_t)2 * nkp, s); if (rc) return rc;       // [0, nkp): matches by queryIdx, [nkp, 2 nkp): observations by point id
    memcpy(m->h_fin, kps, (size_t)nkp * sizeof(ssvio_keypoint));
    if (nm) memcpy(m->h_fin + kb, last_match_vec, (size_t)nm * 16);
    if (nobs) memcpy(m->h_fin + kb + mb, obs, (size_t)nobs * 8);
    const uint8_t* d_k = m->d_fin; const uint8_t* d_m = d_k + kb; const uint8_t* d_o = d_m + mb; uint8_t* d_d = m->d_frameIn;
    SSVIO_HIP(hipMemcpyAsync(d_d, depth, (size_t)h * stride_px * 2, hipMemcpyHostToDevice, s));
    SSVIO_HIP(hipMemsetAsync(m->d_table, 0xFF, (size_t)2 * nkp * 4, s));
    {
        ProfScope p(m->prof, "k_first_index", s);
        launch_first_index((const int32_t*)d_m, nm, 4, 0, m->d_table, nkp, s);             // InMatchVec, src/Map.cpp:151-162
        launch_first_index((const int32_t*)d_o, nobs, 2, 1, m->d_table + nkp, nkp, s);     // GetMapPointID, src/Frame.cpp:138-147
        SSVIO_HIP(hipGetLastError());
    }
    ManageXform T; std::memcpy(T.m, world_trans, sizeof T.m);
    int32_t* d_mn = (int32_t*)m->d_fout; int32_t* d_mp = (int32_t*)(m->d_fout + ib); float* d_w = (float*)(m->d_fout + 2 * ib);
    {
        ProfScope p(m->prof, "k_manage", s);
        launch_manage((const ssvio_keypoint*)d_k, nkp, (const uint16_t*)d_d, c, T, (const ssvio_dmatch*)d_m, m->d_table,
                      (const int32_t*)d_o, m->d_table + nkp, nkp, d_mn, d_mp, d_w, s);
        SSVIO_HIP(hipGetLastError());
    }
    SSVIO_HIP(hipStreamSynchronize(s));
    memcpy(match_num, m->h_fout, (size_t)nkp * 4);
    memcpy(map_point, m->h_fout + ib, (size_t)nkp * 4);
    memcpy(world, m->h_fout + 2 * ib, (size_t)nkp * 12);
    m->prof.resolve();
    return SSVIO_OK;
}

}  // extern "C"

extern "C" int ssvio_rgb_cloud(ssvio_matcher* m, const uint16_t* depth, int depth_stride_px, const uint8_t* bgr, int bgr_stride,
                               int w, int h, const ssvio_camera* cam, double depth_scale, void* out)
{
    if (mt_check(m)) return SSVIO_ERR_INVALID;
    FrameCam c; int rc = cam_dev(cam, w, h, depth_stride_px, c); if (rc) return rc;
    if (!depth || !bgr || !out || bgr_stride < 3 * w) { set_error("bad cloud arguments"); return SSVIO_ERR_INVALID; }
    if ((size_t)w * h > ((size_t)1 << 28)) { set_error("image too large"); return SSVIO_ERR_INVALID; }
    SSVIO_HIP(hipSetDevice(m->device));
    hipStream_t s = m->stream;
    const size_t db = align256((size_t)h * depth_stride_px * 2), cb = align256((size_t)h * bgr_stride), ob = (size_t)w * h * 32;
    rc = grow(m->d_frameIn, m->frameInCap, db + cb, s); if (rc) return rc;
    rc = grow(m->d_frameOut, m->frameOutCap, ob, s); if (rc) return rc;
    SSVIO_HIP(hipMemcpyAsync(m->d_frameIn, depth, (size_t)h * depth_stride_px * 2, hipMemcpyHostToDevice, s));
    SSVIO_HIP(hipMemcpyAsync(m->d_frameIn + db, bgr, (size_t)h * bgr_stride, hipMemcpyHostToDevice, s));
    {
        ProfScope p(m->prof, "k_rgb_cloud", s);
        launch_rgb_cloud((const uint16_t*)m->d_frameIn, depth_stride_px, m->d_frameIn + db, bgr_stride, w, h, c.cx, c.cy, c.invFx, c.invFy,
                         depth_scale, m->d_frameOut, s);
        SSVIO_HIP(hipGetLastError());
    }
    SSVIO_HIP(hipMemcpyAsync(out, m->d_frameOut, ob, hipMemcpyDeviceToHost, s));
    SSVIO_HIP(hipStreamSynchronize(s));
    m->prof.resolve();
    return SSVIO_OK;
}
