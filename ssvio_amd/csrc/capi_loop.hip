// capi_loop.hip -- the loop-candidate walk (BASELINE config 5) as one C-ABI call per rank: all unordered pairs of
// nsets key frames, MatchByBruteForce per pair-block (src/FeatureMatching.cpp:20-55), cut into super-blocks of sb x sb key frames
// that are dealt to the ranks.  The rank's super-blocks are walked natively: pair list built on the device, batched matcher,
// per-pair {pair, n_matches, sum_dist} records reduced on the device (k_pair_summary) -- nothing returns to the host and the
// stream is never waited for.  The one collective (all_gather of the records) stays with the caller.
#include "capi_internal.h"
#include <tuple>

namespace {

struct Tile { long n; int bi, bj; };

// Tiles (bi, bj), bj <= bi, of the lower triangle of the key-frame x key-frame block matrix, largest first, each to the rank with
// the fewest pair-blocks so far (ties: lowest rank) -- the same deterministic deal as ssvio_amd/dist.py:deal_super_blocks, so
// every rank (and the Python driver) computes the same plan.
void deal(int nsets, int sb, int world, std::vector<std::vector<Tile>>& plan, std::vector<long>& load)
{
    const int nb = (nsets + sb - 1) / sb;
    std::vector<Tile> tiles;
    for (int bi = 0; bi < nb; ++bi)
        for (int bj = 0; bj <= bi; ++bj) {
            const long ni = std::min((bi + 1) * sb, nsets) - bi * sb, nj = std::min((bj + 1) * sb, nsets) - bj * sb;
            const long n = bi == bj ? ni * (ni - 1) / 2 : ni * nj;
            if (n) tiles.push_back({n, bi, bj});
        }
    std::stable_sort(tiles.begin(), tiles.end(), [](const Tile& a, const Tile& b) { return std::make_tuple(-a.n, a.bi, a.bj) < std::make_tuple(-b.n, b.bi, b.bj); });
    plan.assign(world, {}); load.assign(world, 0);
    for (const Tile& t : tiles) {
        int r = 0;
        for (int k = 1; k < world; ++k) if (load[k] < load[r]) r = k;
        plan[r].push_back(t); load[r] += t.n;
    }
}

// the pair-blocks (i, j), j < i, of tile (bi, bj): i-major, j ascending (the order of dist.py:super_block_pairs)
__global__ __launch_bounds__(256) void k_super_block_pairs(int i0, int ni, int j0, int nj, int diagonal, int* __restrict__ pairs)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= ni * nj) return;
    const int li = t / nj, lj = t - li * nj;
    int at;
    if (diagonal) { if (lj >= li) return; at = li * (li - 1) / 2 + lj; }
    else at = t;
    pairs[2 * at] = i0 + li; pairs[2 * at + 1] = j0 + lj;
}

}  // namespace

extern "C" {

int ssvio_loop_candidates_plan(int nsets, int sb, int rank, int world, int64_t* n_mine, int64_t* n_max, int* n_tiles)
{
    if (nsets < 0 || sb <= 0 || world <= 0 || rank < 0 || rank >= world) { set_error("bad plan arguments"); return SSVIO_ERR_INVALID; }
    std::vector<std::vector<Tile>> plan; std::vector<long> load;
    deal(nsets, sb, world, plan, load);
    if (n_mine) *n_mine = load[rank];
    if (n_max) *n_max = *std::max_element(load.begin(), load.end());
    if (n_tiles) *n_tiles = (int)plan[rank].size();
    return SSVIO_OK;
}

int ssvio_loop_candidates_dev(ssvio_matcher* m, const void* d_desc, int set_stride_rows, int max_rows, const int32_t* d_counts,
                              int nsets, int sb, int rank, int world, int mode, void* d_records, int64_t records_cap, int64_t* n_records)
{
    if (mt_check(m)) return SSVIO_ERR_INVALID;
    if (!d_desc || !d_counts || !d_records || nsets < 0 || sb <= 0 || sb > 1024 || world <= 0 || rank < 0 || rank >= world || max_rows <= 0 || set_stride_rows < max_rows) {
        set_error("bad loop-candidate arguments"); return SSVIO_ERR_INVALID;
    }
    if (mode < 0 || mode == 3 || mode > SSVIO_MODE_CROSSCHECK_CV) { set_error("mode must be SSVIO_MODE_BRUTEFORCE, _KNN2_RATIO, _CROSSCHECK or _CROSSCHECK_CV"); return SSVIO_ERR_INVALID; }
    if (max_rows > (1 << 23)) { set_error("more than 2^23 rows per set"); return SSVIO_ERR_INVALID; }
    if ((long)nsets * (nsets - 1) / 2 > 0x7FFFFFFFL) { set_error("more than 2^31 key-frame pairs: the record's pair id is 32 bits"); return SSVIO_ERR_INVALID; }
    std::vector<std::vector<Tile>> plan; std::vector<long> load;
    deal(nsets, sb, world, plan, load);
    if (n_records) *n_records = load[rank];
    if (load[rank] > records_cap) { set_error("%ld records for this rank, the buffer holds %ld", load[rank], (long)records_cap); return SSVIO_ERR_CAPACITY; }
    SSVIO_HIP(hipSetDevice(m->device));
    hipStream_t s = m->stream;
    // one device pair list for the largest tile; the walk is stream-ordered, so a tile's list is rewritten only after the previous
    // tile's kernels have read theirs
    int rc = grow(m->d_pairs, m->pairsCap, (size_t)sb * sb * 2, s); if (rc) return rc;
    m->h_pairs.clear();                                   // the cache of upload_pairs no longer describes d_pairs
    ssvio_pair_summary* rec = (ssvio_pair_summary*)d_records;
    long o = 0;
    for (const Tile& t : plan[rank]) {
        const int i0 = t.bi * sb, j0 = t.bj * sb;
        const int ni = std::min(i0 + sb, nsets) - i0, nj = std::min(j0 + sb, nsets) - j0;
        hipLaunchKernelGGL(k_super_block_pairs, dim3((ni * nj + 255) / 256), dim3(256), 0, s, i0, ni, j0, nj, t.bi == t.bj ? 1 : 0, m->d_pairs);
        SSVIO_HIP(hipGetLastError());
        rc = match_batch_on_device_pairs(m, (const uint8_t*)d_desc, set_stride_rows, d_counts, (int)t.n, max_rows, mode); if (rc) return rc;
        {
            ProfScope p(m->prof, "k_pair_summary", s);
            launch_pair_summary(m->d_out, m->d_nout, m->lastMaxRows, m->d_pairs, (int)t.n, SSVIO_PAIR_ID_TRIANGULAR, 0, rec + o, s);
            SSVIO_HIP(hipGetLastError());
        }
        o += t.n;
    }
    return SSVIO_OK;
}

}  // extern "C"
