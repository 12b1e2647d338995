// capi_voxel.hip -- Map::MapPointCloudFusion (src/Map.cpp:164-197; SURVEY.md 8-f4, second half) on gfx950: the point-cloud map
// stays in HBM inside the matcher handle; one call = one iteration of the reference's key-frame loop:
//   transform the frame's cloud (Affine3f::Identity().rotate(R).translate(t): p' = R*p + R*t, :175-181), append it to the map
//   (:183), pcl::VoxelGrid with leaf _VoxelSize over the whole (:185-190), the result is the new map (:192).
// PCL is restated (oracle/frame_oracle.c names the version and the one canonical choice: a voxel's points are summed in their
// order in the concatenated cloud, where PCL's unstable std::sort leaves the order open).  Parity unpinned: PCL is absent.
//
// Kernels: k_vox_bounds + k_vox_reduce (transform + append + min/max, no atomics), k_vox_keys (voxel index per point), rocPRIM radix
// sort of (index, position) pairs -- LSD radix sort is stable, which IS the canonical order --, k_vox_heads + scan + k_vox_segments
// (run boundaries), k_vox_centroid (sequential float sums per run in that order; a run longer than 32 points is gathered by the whole
// wave 512 points at a time and added channel by channel, still one by one).  HBM-bound gather / scatter work; no host round trip
// inside a call except the final count.
#include "capi_internal.h"
#include <rocprim/rocprim.hpp>
#include <cfloat>
#include <cmath>

namespace {

struct Pt { float x, y, z, w; uint32_t rgba; uint32_t pad[3]; };        // pcl::PointXYZRGB, 32 bytes (bytes of rgba: b, g, r, a)
static_assert(sizeof(Pt) == 32, "PointXYZRGB is 32 bytes");

struct VoxState {
    uint32_t mn[3], mx[3];      // min / max of the coordinates as order-preserving unsigned keys
    int nonFinite, overflow, total, pad;
};

__device__ __forceinline__ uint32_t fkey(float f) { const uint32_t u = __float_as_uint(f); return (u >> 31) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float funkey(uint32_t k) { return __uint_as_float((k >> 31) ? (k & 0x7FFFFFFFu) : ~k); }

struct Mat34 { float m[3][4]; };

// P[nf + i] = transform(cloud[i]) (src/Map.cpp:181, pcl::transformPointCloud on a dense cloud); min / max over all of P as one
// record of 7 words per workgroup (min xyz, max xyz, non-finite flag).  No atomics: the clouds arrive in column order, so a bound
// would improve wave after wave and every wave would queue on the same six addresses.
__global__ __launch_bounds__(256) void k_vox_bounds(Pt* __restrict__ P, int nf, const Pt* __restrict__ cloud, int nc, Mat34 M, uint32_t* __restrict__ part)
{
    __shared__ uint32_t red[4][7];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int n = nf + nc;
    float x = 0, y = 0, z = 0; bool live = i < n;
    if (live) {
        if (i >= nf) {
            Pt p = cloud[i - nf];
            x = ((M.m[0][0] * p.x + M.m[0][1] * p.y) + M.m[0][2] * p.z) + M.m[0][3];
            y = ((M.m[1][0] * p.x + M.m[1][1] * p.y) + M.m[1][2] * p.z) + M.m[1][3];
            z = ((M.m[2][0] * p.x + M.m[2][1] * p.y) + M.m[2][2] * p.z) + M.m[2][3];
            p.x = x; p.y = y; p.z = z;
            P[i] = p;
        } else { x = P[i].x; y = P[i].y; z = P[i].z; }
    }
    const bool bad = live && !(isfinite(x) && isfinite(y) && isfinite(z));
    uint32_t lo[3] = {live ? fkey(x) : 0xFFFFFFFFu, live ? fkey(y) : 0xFFFFFFFFu, live ? fkey(z) : 0xFFFFFFFFu};
    uint32_t hi[3] = {live ? fkey(x) : 0u, live ? fkey(y) : 0u, live ? fkey(z) : 0u};
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int k = 0; k < 3; ++k) { lo[k] = min(lo[k], (uint32_t)__shfl_xor((int)lo[k], o)); hi[k] = max(hi[k], (uint32_t)__shfl_xor((int)hi[k], o)); }
    const int w = threadIdx.x >> 6;
    const unsigned long long anyBad = __ballot(bad);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { red[w][k] = lo[k]; red[w][3 + k] = hi[k]; }
        red[w][6] = anyBad ? 1u : 0u;
    }
    __syncthreads();
    if (threadIdx.x < 7) {
        const int k = threadIdx.x;
        uint32_t v = red[0][k];
#pragma unroll
        for (int q = 1; q < 4; ++q) v = k < 3 ? min(v, red[q][k]) : max(v, red[q][k]);
        part[(size_t)blockIdx.x * 7 + k] = v;
    }
}

// the workgroups' records -> the state block (one workgroup; also resets the flags of the call)
__global__ __launch_bounds__(256) void k_vox_reduce(const uint32_t* __restrict__ part, int nb, VoxState* st)
{
    __shared__ uint32_t red[4][7];
    uint32_t v[7] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u, 0u};
    for (int b = threadIdx.x; b < nb; b += 256)
#pragma unroll
        for (int k = 0; k < 7; ++k) { const uint32_t t = part[(size_t)b * 7 + k]; v[k] = k < 3 ? min(v[k], t) : max(v[k], t); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int k = 0; k < 7; ++k) { const uint32_t t = (uint32_t)__shfl_xor((int)v[k], o); v[k] = k < 3 ? min(v[k], t) : max(v[k], t); }
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < 7; ++k) red[threadIdx.x >> 6][k] = v[k];
    __syncthreads();
    if (threadIdx.x < 7) {
        const int k = threadIdx.x;
        uint32_t r = red[0][k];
        for (int q = 1; q < 4; ++q) r = k < 3 ? min(r, red[q][k]) : max(r, red[q][k]);
        if (k < 3) st->mn[k] = r; else if (k < 6) st->mx[k - 3] = r; else { st->nonFinite = (int)r; st->overflow = 0; st->total = 0; st->pad = 0; }
    }
}

// pcl::VoxelGrid::applyFilter, first pass: the voxel index of every point
__global__ __launch_bounds__(256) void k_vox_keys(const Pt* __restrict__ P, int n, float inv, VoxState* st, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const float mn[3] = {funkey(st->mn[0]), funkey(st->mn[1]), funkey(st->mn[2])}, mx[3] = {funkey(st->mx[0]), funkey(st->mx[1]), funkey(st->mx[2])};
    const long long dx = (long long)((mx[0] - mn[0]) * inv) + 1, dy = (long long)((mx[1] - mn[1]) * inv) + 1, dz = (long long)((mx[2] - mn[2]) * inv) + 1;
    // "Leaf size is too small for the input dataset. Integer indices would overflow": the filter returns its input -- here every
    // point becomes its own voxel, in input order, which is the same cloud (a centroid of one point is the point)
    const bool overflow = dx * dy * dz > 2147483647ll;
    if (i == 0) st->overflow = overflow ? 1 : 0;
    if (i >= n) return;
    uint32_t idx = (uint32_t)i;
    if (!overflow) {
        const int b0 = (int)floorf(mn[0] * inv), b1 = (int)floorf(mn[1] * inv), b2 = (int)floorf(mn[2] * inv);
        const int d0 = (int)floorf(mx[0] * inv) - b0 + 1, d1 = (int)floorf(mx[1] * inv) - b1 + 1;
        const Pt p = P[i];
        const int i0 = (int)(floorf(p.x * inv) - (float)b0), i1 = (int)(floorf(p.y * inv) - (float)b1), i2 = (int)(floorf(p.z * inv) - (float)b2);
        idx = (uint32_t)(i0 + i1 * d0 + i2 * d0 * d1);
    }
    keys[i] = idx; vals[i] = (uint32_t)i;
}

__global__ __launch_bounds__(256) void k_vox_heads(const uint32_t* __restrict__ keys, int n, uint32_t* __restrict__ head)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < n) head[r] = (r == 0 || keys[r] != keys[r - 1]) ? 1u : 0u;
}

// run boundaries: segStart[v] = first sorted rank of output voxel v, segStart[total] = n
__global__ __launch_bounds__(256) void k_vox_segments(const uint32_t* __restrict__ head, const uint32_t* __restrict__ pos, int n,
                                                      uint32_t* __restrict__ segStart, VoxState* st, int* __restrict__ totalHost)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    if (head[r]) segStart[pos[r] - 1] = (uint32_t)r;
    if (r == n - 1) { const int total = (int)pos[r]; segStart[total] = (uint32_t)n; st->total = total; totalHost[0] = total; totalHost[1] = st->nonFinite; totalHost[2] = st->overflow; }
}

struct Acc { float x, y, z, r, g, b, a; };
__device__ __forceinline__ void acc_add(Acc& s, const Pt& p)
{
    s.x += p.x; s.y += p.y; s.z += p.z;
    s.b += (float)(p.rgba & 255u); s.g += (float)((p.rgba >> 8) & 255u); s.r += (float)((p.rgba >> 16) & 255u); s.a += (float)(p.rgba >> 24);
}
__device__ __forceinline__ Pt acc_get(const Acc& s, int n)
{
    const float c = (float)n;                                   // pcl::CentroidPoint: xyz / n, (uint32)(channel sum / n)
    Pt o; o.x = s.x / c; o.y = s.y / c; o.z = s.z / c; o.w = 1.0f;
    o.rgba = ((uint32_t)(s.a / c) & 255u) << 24 | ((uint32_t)(s.r / c) & 255u) << 16 | ((uint32_t)(s.g / c) & 255u) << 8 | ((uint32_t)(s.b / c) & 255u);
    o.pad[0] = o.pad[1] = o.pad[2] = 0;
    return o;
}

constexpr int kVoxShort = 32;
constexpr int kVoxDeep = 8;                      // points per lane and step of a long run: 512 points in flight per wave
// fourth pass: one output point per run, its points added one by one in sorted (= input) order.  One wave per workgroup.
__global__ __launch_bounds__(64) void k_vox_centroid(const Pt* __restrict__ P, const uint32_t* __restrict__ vals, const uint32_t* __restrict__ segStart,
                                                     const VoxState* __restrict__ st, Pt* __restrict__ out)
{
    const int total = st->total;
    const int v = blockIdx.x * 64 + threadIdx.x;
    if (__builtin_amdgcn_readfirstlane(v) >= total) return;            // the wave's first voxel: wave-uniform exit
    const int lane = threadIdx.x;
    int s = 0, e = 0;
    if (v < total) { s = (int)segStart[v]; e = (int)segStart[v + 1]; }
    const bool isLong = e - s > kVoxShort;
    if (v < total && !isLong) {
        Acc a{0, 0, 0, 0, 0, 0, 0};
        for (int j = s; j < e; ++j) acc_add(a, P[vals[j]]);
        out[v] = acc_get(a, e - s);
    }
    // long runs (e.g. every zero-depth pixel of a frame lands on one point): the wave gathers 512 points per step (coalesced index
    // loads, 8 independent point loads per lane), lays their seven summands out channel-major in LDS, and lane c < 7 adds channel
    // c's values in order -- the same sequence of float additions per channel as the short path.  The gathers of step k + 1 and the
    // index loads of step k + 2 are in flight while step k is summed: the chain of dependent adds is what the run costs.
    __shared__ float stage[7][64 * kVoxDeep];
    unsigned long long todo = __ballot(isLong);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1; todo &= todo - 1;
        const int ls = __shfl(s, src), le = __shfl(e, src);
        float acc = 0.0f;                                               // lane c < 7: running sum of channel c (x, y, z, b, g, r, a)
        constexpr int kStep = 64 * kVoxDeep;
        // every load is unconditional (positions past the run's end repeat its last point and are never summed): a load under a
        // lane predicate gets its own wait, and the eight gathers of a step would queue behind each other
        uint32_t idx[kVoxDeep]; float4 pv[kVoxDeep]; uint32_t pc[kVoxDeep];
#pragma unroll
        for (int t = 0; t < kVoxDeep; ++t) idx[t] = vals[min(ls + 64 * t + lane, le - 1)];
#pragma unroll
        for (int t = 0; t < kVoxDeep; ++t) { pv[t] = *(const float4*)&P[idx[t]]; pc[t] = P[idx[t]].rgba; }
#pragma unroll
        for (int t = 0; t < kVoxDeep; ++t) idx[t] = vals[min(ls + kStep + 64 * t + lane, le - 1)];
        for (int j0 = ls; j0 < le; j0 += kStep) {
            const int cnt = min(kStep, le - j0);
#pragma unroll
            for (int t = 0; t < kVoxDeep; ++t) {
                const int q = 64 * t + lane;
                stage[0][q] = pv[t].x; stage[1][q] = pv[t].y; stage[2][q] = pv[t].z;
                stage[3][q] = (float)(pc[t] & 255u); stage[4][q] = (float)((pc[t] >> 8) & 255u);
                stage[5][q] = (float)((pc[t] >> 16) & 255u); stage[6][q] = (float)(pc[t] >> 24);
            }
#pragma unroll
            for (int t = 0; t < kVoxDeep; ++t) { pv[t] = *(const float4*)&P[idx[t]]; pc[t] = P[idx[t]].rgba; }
#pragma unroll
            for (int t = 0; t < kVoxDeep; ++t) idx[t] = vals[min(j0 + 2 * kStep + 64 * t + lane, le - 1)];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane < 7) {
                const float* row = stage[lane];
                int l = 0;
                for (; l + 64 <= cnt; l += 64) {
#pragma unroll
                    for (int u = 0; u < 64; ++u) acc += row[l + u];
                }
                for (; l < cnt; ++l) acc += row[l];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        Acc a;
        a.x = __shfl(acc, 0); a.y = __shfl(acc, 1); a.z = __shfl(acc, 2); a.b = __shfl(acc, 3); a.g = __shfl(acc, 4); a.r = __shfl(acc, 5); a.a = __shfl(acc, 6);
        if (lane == src) out[v] = acc_get(a, le - ls);
    }
}

// keeps the first `keep` elements when a device array grows
template <typename T> int grow_keep(T*& p, size_t& cap, size_t need, size_t keep, hipStream_t s)
{
    if (need <= cap) return SSVIO_OK;
    SSVIO_HIP(hipStreamSynchronize(s));
    need += need / 2;
    if (alloc_fail_injected()) { set_error("hipMalloc: injected failure (SSVIO_TEST_FAIL_ALLOC)"); return SSVIO_ERR_HIP; }
    T* q = nullptr;
    SSVIO_HIP(hipMalloc(&q, need * sizeof(T)));
    if (p && keep) {
        const hipError_t e = hipMemcpy(q, p, keep * sizeof(T), hipMemcpyDeviceToDevice);
        if (e != hipSuccess) { (void)hipFree(q); return hip_fail(e, "hipMemcpy (map grow)", __FILE__, __LINE__); }
    }
    dfree(p); p = q; cap = need;
    return SSVIO_OK;
}

int vox_check(ssvio_matcher* m, const float* this_trans, float voxel_size, int n)
{
    if (mt_check(m)) return SSVIO_ERR_INVALID;
    if (n < 0) { set_error("negative point count"); return SSVIO_ERR_INVALID; }
    if (!this_trans) { set_error("null transform"); return SSVIO_ERR_INVALID; }
    if (!(voxel_size > 0) || !std::isfinite(voxel_size)) { set_error("voxel size must be positive"); return SSVIO_ERR_INVALID; }
    if ((size_t)m->mapCount + (size_t)n > ((size_t)1 << 30)) { set_error("more than 2^30 points"); return SSVIO_ERR_INVALID; }
    return SSVIO_OK;
}

// one iteration of the loop of src/Map.cpp:166-192 with the frame's cloud in device memory
int voxel_run(ssvio_matcher* m, const Pt* d_cloud, int nc, const float* T, float leaf)
{
    hipStream_t s = m->stream;
    const int nf = m->mapCount, n = nf + nc;
    if (n == 0) return SSVIO_OK;
    int rc = grow_keep(m->d_mapA, m->mapCapA, (size_t)n * 32, (size_t)nf * 32, s); if (rc) return rc;
    rc = grow(m->d_mapB, m->mapCapB, (size_t)n * 32, s); if (rc) return rc;
    rc = grow(m->d_voxKeys, m->voxKeysCap, (size_t)n * 5 + 8, s); if (rc) return rc;      // keys / vals in, keys / vals out, segStart[n + 1]
    if (!m->d_voxState) SSVIO_HIP(hipMalloc(&m->d_voxState, sizeof(VoxState)));
    if (!m->h_voxTotal) SSVIO_HIP(hipHostMalloc((void**)&m->h_voxTotal, 4 * sizeof(int), hipHostMallocMapped));      // (each on its own: a failure of one leaves no half-made pair behind)
    if (!m->d_voxTotal) SSVIO_HIP(hipHostGetDevicePointer((void**)&m->d_voxTotal, m->h_voxTotal, 0));
    uint32_t* keysIn = m->d_voxKeys; uint32_t* valsIn = keysIn + n; uint32_t* keysOut = valsIn + n; uint32_t* valsOut = keysOut + n;
    uint32_t* head = keysIn;            // keysIn / valsIn are dead after the sort: run-head flags and their inclusive scan
    uint32_t* pos = valsIn;
    uint32_t* segStart = valsOut + n;   // n + 1 entries
    size_t tmpSort = 0, tmpScan = 0;
    SSVIO_HIP(rocprim::radix_sort_pairs(nullptr, tmpSort, keysIn, keysOut, valsIn, valsOut, (size_t)n, 0, 32, s));
    SSVIO_HIP(rocprim::inclusive_scan(nullptr, tmpScan, head, pos, (size_t)n, rocprim::plus<uint32_t>(), s));
    rc = grow(m->d_voxTmp, m->voxTmpCap, std::max(tmpSort, tmpScan) + 256, s); if (rc) return rc;
    VoxState* st = (VoxState*)m->d_voxState;
    Pt* P = (Pt*)m->d_mapA; Pt* O = (Pt*)m->d_mapB;
    Mat34 M;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) M.m[r][c] = T[c * 4 + r];                      // rotation = this_trans.block<3,3>(0,0), :175
        M.m[r][3] = M.m[r][0] * T[12] + (M.m[r][1] * T[13] + M.m[r][2] * T[14]);   // translate(): linear * translation, :179
    }
    const int nb = (n + 255) / 256;
    m->h_voxTotal[0] = -1;
    {
        ProfScope p(m->prof, "k_vox_bounds", s);
        hipLaunchKernelGGL(k_vox_bounds, dim3(nb), dim3(256), 0, s, P, nf, d_cloud, nc, M, keysOut);      // 7 * nb <= 3 n + 8 words free behind valsIn
        hipLaunchKernelGGL(k_vox_reduce, dim3(1), dim3(256), 0, s, keysOut, nb, st);
    }
    {
        ProfScope p(m->prof, "k_vox_keys", s);
        hipLaunchKernelGGL(k_vox_keys, dim3(nb), dim3(256), 0, s, P, n, 1.0f / leaf, st, keysIn, valsIn);
    }
    {
        ProfScope p(m->prof, "vox_radix_sort", s);
        SSVIO_HIP(rocprim::radix_sort_pairs(m->d_voxTmp, tmpSort, keysIn, keysOut, valsIn, valsOut, (size_t)n, 0, 32, s));
    }
    {
        ProfScope p(m->prof, "k_vox_segments", s);
        hipLaunchKernelGGL(k_vox_heads, dim3(nb), dim3(256), 0, s, keysOut, n, head);
        SSVIO_HIP(rocprim::inclusive_scan(m->d_voxTmp, tmpScan, head, pos, (size_t)n, rocprim::plus<uint32_t>(), s));
        hipLaunchKernelGGL(k_vox_segments, dim3(nb), dim3(256), 0, s, head, pos, n, segStart, st, m->d_voxTotal);
    }
    {
        ProfScope p(m->prof, "k_vox_centroid", s);
        hipLaunchKernelGGL(k_vox_centroid, dim3((n + 63) / 64), dim3(64), 0, s, P, valsOut, segStart, st, O);
    }
    SSVIO_HIP(hipGetLastError());
    SSVIO_HIP(hipStreamSynchronize(s));                     // the count of the new map (mapped host memory) sizes the next call
    m->prof.resolve();
    const int total = m->h_voxTotal[0];
    if (total < 0 || total > n) { set_error("internal: voxel count %d of %d points", total, n); return SSVIO_ERR_HIP; }
    if (m->h_voxTotal[1]) { set_error("non-finite point coordinates: the reference's clouds are dense (is_dense = true), so is this path"); return SSVIO_ERR_INVALID; }
    std::swap(m->d_mapA, m->d_mapB); std::swap(m->mapCapA, m->mapCapB);
    m->mapCount = total;
    return SSVIO_OK;
}

}  // namespace

extern "C" {

int ssvio_map_set(ssvio_matcher* m, const void* points, int n)
{
    if (mt_check(m)) return SSVIO_ERR_INVALID;
    if (n < 0 || (n > 0 && !points)) { set_error("bad map arguments"); return SSVIO_ERR_INVALID; }
    SSVIO_HIP(hipSetDevice(m->device));
    m->mapCount = 0;
    if (n == 0) return SSVIO_OK;
    const int rc = grow(m->d_mapA, m->mapCapA, (size_t)n * 32, m->stream); if (rc) return rc;
    SSVIO_HIP(hipMemcpyAsync(m->d_mapA, points, (size_t)n * 32, hipMemcpyHostToDevice, m->stream));
    SSVIO_HIP(hipStreamSynchronize(m->stream));
    m->mapCount = n;
    return SSVIO_OK;
}

int ssvio_map_fuse_dev(ssvio_matcher* m, const void* d_cloud, int n, const float* this_trans, float voxel_size)
{
    int rc = vox_check(m, this_trans, voxel_size, n); if (rc) return rc;
    if (n > 0 && !d_cloud) { set_error("null cloud"); return SSVIO_ERR_INVALID; }
    SSVIO_HIP(hipSetDevice(m->device));
    return voxel_run(m, (const Pt*)d_cloud, n, this_trans, voxel_size);
}

int ssvio_map_fuse(ssvio_matcher* m, const void* cloud, int n, const float* this_trans, float voxel_size)
{
    int rc = vox_check(m, this_trans, voxel_size, n); if (rc) return rc;
    if (n > 0 && !cloud) { set_error("null cloud"); return SSVIO_ERR_INVALID; }
    SSVIO_HIP(hipSetDevice(m->device));
    if (n > 0) {
        rc = grow(m->d_frameOut, m->frameOutCap, (size_t)n * 32, m->stream); if (rc) return rc;
        SSVIO_HIP(hipMemcpyAsync(m->d_frameOut, cloud, (size_t)n * 32, hipMemcpyHostToDevice, m->stream));
    }
    return voxel_run(m, (const Pt*)m->d_frameOut, n, this_trans, voxel_size);
}

int ssvio_map_fuse_frame(ssvio_matcher* m, const uint16_t* depth, int depth_stride_px, const uint8_t* bgr, int bgr_stride, int w, int h,
                         const ssvio_camera* cam, double depth_scale, const float* this_trans, float voxel_size)
{
    if (mt_check(m)) return SSVIO_ERR_INVALID;
    if (!cam || w <= 0 || h <= 0 || depth_stride_px < w || !depth || !bgr || bgr_stride < 3 * w || (size_t)w * h > ((size_t)1 << 28)) { set_error("bad cloud arguments"); return SSVIO_ERR_INVALID; }
    int rc = vox_check(m, this_trans, voxel_size, w * h); if (rc) return rc;
    SSVIO_HIP(hipSetDevice(m->device));
    hipStream_t s = m->stream;
    const size_t db = align256((size_t)h * depth_stride_px * 2), cb = align256((size_t)h * bgr_stride), ob = (size_t)w * h * 32;
    rc = grow(m->d_frameIn, m->frameInCap, db + cb, s); if (rc) return rc;
    rc = grow(m->d_frameOut, m->frameOutCap, ob, s); if (rc) return rc;
    SSVIO_HIP(hipMemcpyAsync(m->d_frameIn, depth, (size_t)h * depth_stride_px * 2, hipMemcpyHostToDevice, s));
    SSVIO_HIP(hipMemcpyAsync(m->d_frameIn + db, bgr, (size_t)h * bgr_stride, hipMemcpyHostToDevice, s));
    {
        ProfScope p(m->prof, "k_rgb_cloud", s);
        launch_rgb_cloud((const uint16_t*)m->d_frameIn, depth_stride_px, m->d_frameIn + db, bgr_stride, w, h, cam->cx, cam->cy, 1 / cam->fx, 1 / cam->fy,
                         depth_scale, m->d_frameOut, s);
        SSVIO_HIP(hipGetLastError());
    }
    return voxel_run(m, (const Pt*)m->d_frameOut, w * h, this_trans, voxel_size);
}

int ssvio_map_fetch(ssvio_matcher* m, void* out, int cap, int* n)
{
    if (mt_check(m)) return SSVIO_ERR_INVALID;
    if (!n) { set_error("null count"); return SSVIO_ERR_INVALID; }
    *n = m->mapCount;
    if (!out || m->mapCount == 0) return SSVIO_OK;
    if (cap < m->mapCount) { set_error("output capacity %d < %d map points", cap, m->mapCount); return SSVIO_ERR_CAPACITY; }
    SSVIO_HIP(hipSetDevice(m->device));
    SSVIO_HIP(hipMemcpyAsync(out, m->d_mapA, (size_t)m->mapCount * 32, hipMemcpyDeviceToHost, m->stream));
    SSVIO_HIP(hipStreamSynchronize(m->stream));
    return SSVIO_OK;
}

int ssvio_map_device(ssvio_matcher* m, const void** d_points, int* n)
{
    if (mt_check(m)) return SSVIO_ERR_INVALID;
    if (d_points) *d_points = m->d_mapA;
    if (n) *n = m->mapCount;
    return SSVIO_OK;
}

}  // extern "C"
