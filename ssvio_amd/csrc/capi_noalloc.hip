// capi_noalloc.hip -- ssvio_match_batch_dev_noalloc: the batched matcher for callers that cannot allocate (code under HIP
// graph capture, tight serving loops, harnesses that forbid hipMalloc).  It runs the kernels of ssvio_match_batch_dev
// (k_hamming_knn2 / k_match_merge / k_match_finalize) with their partial 2-NN lists in a fixed device array instead of
// the handle's growing buffers; pair list, inputs and outputs are the caller's device memory.  No hipMalloc / hipFree /
// hipHostMalloc / event creation / copy command happens in the call.
#include "capi_internal.h"
#include <mutex>

// The scratch: [pair][chunk][max_rows] uint2 partial lists (both directions for the cross-check).  One copy per GPU the
// library is loaded on, held until the process ends.  Every call on a GPU shares it: calls that overlap on one GPU race.
// Nothing in it has to start at zero (k_hamming_knn2 writes every (pair, chunk, row < count) entry that is read later).
static __device__ __align__(256) unsigned char scratch[SSVIO_NOALLOC_SCRATCH_BYTES];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_mu;
void* g_scratchPtr[kMaxDevices];          // device address of `scratch` on each device (the array's name is only a host-side handle)
}

namespace ssvio {
// Set-up, once per device, with that device current (called from ssvio_matcher_create): the address of this device's copy.
int noalloc_scratch_init(int device)
{
    if (device < 0 || device >= kMaxDevices) { set_error("device %d beyond the %d the fixed scratch is tracked for", device, kMaxDevices); return SSVIO_ERR_INVALID; }
    std::lock_guard<std::mutex> l(g_mu);
    if (g_scratchPtr[device]) return SSVIO_OK;
    void* p = nullptr;
    SSVIO_HIP(hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch)));
    if (!p) { set_error("hipGetSymbolAddress returned a null scratch address"); return SSVIO_ERR_HIP; }
    g_scratchPtr[device] = p;
    return SSVIO_OK;
}
}  // namespace ssvio

extern "C" int ssvio_match_batch_dev_noalloc(ssvio_matcher* m, const void* d_desc, int set_stride_rows, int max_rows,
                                             const int32_t* d_counts, const int32_t* d_pairs, int npairs, int mode,
                                             void* d_matches, int32_t* d_nmatches)
{
    if (mt_check(m)) return SSVIO_ERR_INVALID;
    if (!d_desc || !d_counts || !d_pairs || !d_matches || !d_nmatches || npairs <= 0 || max_rows <= 0 || set_stride_rows < max_rows) {
        set_error("bad batch arguments"); return SSVIO_ERR_INVALID;
    }
    if (mode < 0 || mode == 3 || mode > SSVIO_MODE_CROSSCHECK_CV) { set_error("mode must be SSVIO_MODE_BRUTEFORCE, _KNN2_RATIO, _CROSSCHECK or _CROSSCHECK_CV"); return SSVIO_ERR_INVALID; }
    if (max_rows > (1 << 23)) { set_error("more than 2^23 rows per set"); return SSVIO_ERR_INVALID; }
    void* base = nullptr;
    {
        std::lock_guard<std::mutex> l(g_mu);
        if (m->device >= 0 && m->device < kMaxDevices) base = g_scratchPtr[m->device];
    }
    if (!base) { set_error("fixed scratch of device %d was not set up", m->device); return SSVIO_ERR_INVALID; }
    MatchPlan p;
    const int rc = match_plan(max_rows, max_rows, npairs, mode, m->kernel, 32, (size_t)SSVIO_NOALLOC_SCRATCH_BYTES / sizeof(uint2), p);
    if (rc) return rc;                                           // SSVIO_ERR_CAPACITY: beyond the fixed scratch, nothing was launched
    SSVIO_HIP(hipSetDevice(m->device));
    uint2* partial = (uint2*)base;
    uint2* partialRev = p.partialRevEntries ? partial + (size_t)SSVIO_NOALLOC_SCRATCH_BYTES / sizeof(uint2) / 2 : nullptr;
    return match_launch(m, p, (const uint8_t*)d_desc, set_stride_rows, d_counts, (const uint8_t*)d_desc, set_stride_rows, d_counts,
                        d_pairs, npairs, max_rows, max_rows, mode, partial, partialRev, (ssvio_dmatch*)d_matches, d_nmatches, nullptr,
                        /*profile=*/false);
}
