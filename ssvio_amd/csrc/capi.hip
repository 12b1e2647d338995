// capi.hip -- the C ABI of include/ssvio_gpu.h, part 1: library entry points, the profiler, and the extractor handle
// (device memory, streams, per-batch orchestration of the extraction kernels).  The matcher lives in capi_match.hip,
// MatchByDBoW in capi_bow.hip, the steps behind the matcher in capi_frame.hip.
#include "capi_internal.h"
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

namespace ssvio {
const char* last_error_cstr();

// ---------------------------------------------------------------- profiler
hipEvent_t KernelProfiler::get()
{
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e; (void)hipEventCreate(&e); return e;
}
void KernelProfiler::begin(const char* name, hipStream_t s)
{
    Rec r{name, get(), get()};
    (void)hipEventRecord(r.a, s);
    pending.push_back(r);
}
void KernelProfiler::end(hipStream_t s) { (void)hipEventRecord(pending.back().b, s); }
void KernelProfiler::resolve()
{
    for (auto& r : pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            auto it = std::find_if(stats.begin(), stats.end(), [&](const Stat& s) { return s.name == r.name; });
            if (it == stats.end()) stats.push_back(Stat{r.name, ms, 1});
            else { it->ms += ms; it->launches++; }
        }
        pool.push_back(r.a); pool.push_back(r.b);
    }
    pending.clear();
}
void KernelProfiler::clear() { resolve(); stats.clear(); }
void KernelProfiler::destroy()
{
    for (auto& r : pending) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto e : pool) (void)hipEventDestroy(e);
    pending.clear(); pool.clear(); stats.clear();
}

// ---------------------------------------------------------------- host worker pool (quadtree stage)
class WorkerPool {
public:
    explicit WorkerPool(int n) {
        for (int i = 0; i < n; ++i) th_.emplace_back([this] { loop(); });
    }
    ~WorkerPool() {
        { std::lock_guard<std::mutex> l(m_); stop_ = true; }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    // runs fn(i) for i in [0,n) on the pool plus the calling thread; returns when all are done
    // and no worker is still inside the claim loop (so the next run may reset the counters)
    void run(int n, const std::function<void(int)>& fn) {
        if (n <= 0) return;
        if (th_.empty() || n == 1) { for (int i = 0; i < n; ++i) fn(i); return; }
        {
            std::unique_lock<std::mutex> l(m_);
            dcv_.wait(l, [&] { return active_ == 0; });
            fn_ = &fn; n_ = n; next_.store(0); done_.store(0); ++gen_;
        }
        cv_.notify_all();
        work();
        std::unique_lock<std::mutex> l(m_);
        dcv_.wait(l, [&] { return done_.load() >= n_ && active_ == 0; });
        fn_ = nullptr;
    }
private:
    void work() {
        for (;;) {
            const int i = next_.fetch_add(1);
            if (i >= n_) break;
            (*fn_)(i);
            done_.fetch_add(1);
        }
    }
    void loop() {
        unsigned long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
                ++active_;
            }
            work();
            {
                std::lock_guard<std::mutex> l(m_);
                --active_;
            }
            dcv_.notify_all();
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_; std::condition_variable cv_, dcv_;
    const std::function<void(int)>* fn_ = nullptr;
    int n_ = 0; std::atomic<int> next_{0}, done_{0};
    int active_ = 0;
    unsigned long gen_ = 0; bool stop_ = false;
};

static int host_threads()
{
    if (const char* e = std::getenv("SSVIO_HOST_THREADS")) { const int v = std::atoi(e); if (v >= 1) return std::min(v, 64); }
    const unsigned hc = std::thread::hardware_concurrency();
    return (int)std::min<unsigned>(hc ? hc : 4, 16);
}

int pick_device(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        set_error("no HIP device visible: this library has no CPU fallback");
        return SSVIO_ERR_NOGPU;
    }
    if (device < 0 || device >= n) { set_error("device %d out of range (0..%d)", device, n - 1); return SSVIO_ERR_INVALID; }
    return SSVIO_OK;
}


// test hook (ssvio_test_fail_next_allocs): the next n growing device allocations of a handle report failure
static std::atomic<int> g_failAllocs{0};
bool alloc_fail_injected()
{
    int v = g_failAllocs.load();
    while (v > 0) if (g_failAllocs.compare_exchange_weak(v, v - 1)) return true;
    return false;
}

int prof_out(KernelProfiler& p, const char** names, double* total_ms, int32_t* launches, int cap, int* count)
{
    if (!count) { set_error("null count"); return SSVIO_ERR_INVALID; }
    p.resolve();
    const int n = (int)p.stats.size();
    *count = n;
    for (int i = 0; i < n && i < cap; ++i) {
        if (names) names[i] = p.stats[i].name;
        if (total_ms) total_ms[i] = p.stats[i].ms;
        if (launches) launches[i] = p.stats[i].launches;
    }
    return SSVIO_OK;
}

}  // namespace ssvio

using namespace ssvio;

// =====================================================================================
struct ssvio_orb {
    int device = 0;
    ExtractorTables tab{};
    PyramidGeom geom{};
    KLevels kl{};
    hipStream_t own = nullptr, stream = nullptr;
    hipStream_t side = nullptr;                       // k_blur runs here, beside k_fast / k_quadtree
    hipEvent_t evCand = nullptr, evFork = nullptr, evJoin = nullptr;
    bool pending = false;                             // an _async call has not been waited for yet
    int batchCap = 0;
    int kpCap = 0;                 // keypoints per frame
    int candCap = 0;               // candidates per frame (all levels)
    size_t inBytes = 0;
    uint8_t* d_in = nullptr; uint8_t* d_pyr = nullptr; uint8_t* d_blur = nullptr; uint8_t* d_score = nullptr;
    int16_t* d_tabs = nullptr; std::vector<size_t> xTabOff, yTabOff, qTabOff; std::vector<char> quadOk; std::vector<int> resizeRows;
    std::vector<char> fuse2;           // fuse2[l]: levels l and l + 1 can be built by one k_resize2 launch
    uint2* d_raw = nullptr; int* d_rawCount = nullptr; int* d_cellIni = nullptr;     // FAST output: unordered candidate lists
    uint32_t* d_kval = nullptr; uint32_t* d_kord = nullptr; uint32_t* d_knode = nullptr;   // quadtree per-key scratch
    uint2* d_sel = nullptr; int* d_selCount = nullptr;                                  // selected keypoints per level
    unsigned char* d_qtNode = nullptr; size_t qtNodeStride = 0;                         // k_quadtree node arrays in HBM (quotas > 2045 per level)
    int* d_kpCount = nullptr;
    ssvio_keypoint* d_kpOut = nullptr; uint8_t* d_desc = nullptr;
    int* h_kpCount = nullptr; int* d_kpCountHost = nullptr;   // pinned host array and its device address
    bool hostQuadtree = false;     // SSVIO_QUADTREE=host, or a level quota beyond the GPU kernel's LDS capacity
    int maxAlive = 0;
    // host copies for the host quadtree mode and the debug taps (lazily sized)
    std::vector<uint2> h_raw; std::vector<int> h_rawCount, h_cellIni; std::vector<uint2> h_sel; std::vector<int> h_selCount;
    bool hostRawValid = false;
    std::vector<uint8_t> stageIn;      // packed copy of a strided input frame
    // host entry points: k_describe stores a second copy of its outputs straight into this pinned, device-mapped area (keypoints,
    // then descriptors), so a call needs no copy commands and one stream wait (a copy to pinned memory costs more than k_describe)
    uint8_t* h_landing = nullptr; uint8_t* d_landing = nullptr; size_t landingCap = 0;
    ssvio_keypoint* hostKp = nullptr; uint8_t* hostDesc = nullptr;      // set around one call; device addresses inside d_landing
    uint8_t* h_inMapped = nullptr; uint8_t* d_inMapped = nullptr; size_t inMappedCap = 0;   // small inputs: k_level0 reads the frame from mapped host memory
    // mvImagePyramid of the host single-frame calls (ssvio_orb_set_export_pyramid): k_export_pyramid stores the rimmed levels here
    bool exportPyr = false, exportNow = false, exportValid = false;
    uint8_t* h_pyrExp = nullptr; uint8_t* d_pyrExp = nullptr; size_t pyrExpCap = 0; ExportLayout expLay{}; int expW = 0, expH = 0;
    hipEvent_t evExp = nullptr;
    KernelProfiler prof; std::string profOnly;
    std::unique_ptr<WorkerPool> pool;
    int lastBatch = 0;
    bool haveResults = false;
    // the launch sequence behind k_level0 as a hipGraph (one per batch size and buffer generation), replayed by later calls
    hipGraphExec_t gexec = nullptr; int gBatch = 0; unsigned gGen = 0, allocGen = 1; bool graphFailed = false;
    bool useGraph = false;         // ssvio_orb_set_graph: this handle replays its launch sequence whatever SSVIO_GRAPH says
};

static void orb_free_buffers(ssvio_orb* h)
{
    dfree(h->d_in); dfree(h->d_pyr); dfree(h->d_blur); dfree(h->d_tabs);
    dfree(h->d_raw); dfree(h->d_rawCount); dfree(h->d_cellIni); dfree(h->d_kval); dfree(h->d_kord); dfree(h->d_knode);
    dfree(h->d_sel); dfree(h->d_selCount); dfree(h->d_kpCount); dfree(h->d_kpOut); dfree(h->d_desc); dfree(h->d_qtNode);
    hfree(h->h_kpCount); hfree(h->h_landing); h->landingCap = 0; hfree(h->h_inMapped); h->inMappedCap = 0;
    h->batchCap = 0; h->inBytes = 0; h->haveResults = false;
    ++h->allocGen;                                   // a captured graph holds the old addresses
}

static void fill_klevels(const PyramidGeom& g, KLevels& k)
{
    memset(&k, 0, sizeof k);
    k.nlevels = g.nlevels;
    for (int l = 0; l < g.nlevels; ++l) {
        const LevelGeom& L = g.lv[l];
        KLevel& K = k.lv[l];
        K.w = L.w; K.h = L.h; K.pitch = L.pitch; K.offset = L.offset;
        K.nCols = L.nCols; K.nRows = L.nRows; K.wCell = L.wCell; K.hCell = L.hCell; K.cellBase = L.cellBase;
        K.fastTileBase = L.fastTileBase; K.fastTilesX = L.fastTilesX; K.fastTileBaseT = L.fastTileBaseT; K.fastTileH = L.fastTileH; K.fastTileHT = L.fastTileHT;
        K.blurStripBase = L.blurStripBase; K.blurStripsX = L.blurStripsX; K.blurIters = L.blurIters; K.blurPanelStride = L.blurPanelStride;
        K.quota = L.quota; K.kpSize = L.kpSize; K.scale = L.scale;
        K.rawBase = (unsigned)L.slotBase; K.rawCap = (unsigned)L.slotCap * L.nCols * L.nRows;
        K.selBase = L.selBase; K.nIni = L.nIni; K.hX = L.hX;
        K.wCellMagic = ((1 << 20) + L.wCell - 1) / L.wCell; K.hCellMagic = ((1 << 20) + L.hCell - 1) / L.hCell;
        for (int i = 0; i <= kMaxRoots; ++i) K.rootX[i] = L.rootX[i];
    }
    KLevel& E = k.lv[g.nlevels];
    E.cellBase = g.cellsPerFrame; E.fastTileBase = g.fastTilesPerFrame; E.fastTileBaseT = g.fastTilesPerFrameT; E.blurStripBase = g.blurStripsPerFrame;
    E.rawBase = (unsigned)g.slotsPerFrame;
}

// (re)allocate for a w x h x batch configuration
static int orb_ensure(ssvio_orb* h, int width, int height, int batch, size_t inBytes)
{
    SSVIO_HIP(hipSetDevice(h->device));
    const bool geomChange = width != h->geom.width || height != h->geom.height;
    if (geomChange) {
        PyramidGeom g;
        const int rc = build_geometry(h->tab, width, height, g);
        if (rc) return rc;
        SSVIO_HIP(hipStreamSynchronize(h->stream));
        orb_free_buffers(h);
        h->geom.width = h->geom.height = 0;          // committed below, once the tables are on the device: a failure in between must not
                                                     // leave the new size with a null table pointer for the next call to trust
        // resize tables
        std::vector<int16_t> all;
        h->xTabOff.assign(g.nlevels, 0); h->yTabOff.assign(g.nlevels, 0); h->qTabOff.assign(g.nlevels, 0); h->quadOk.assign(g.nlevels, 0); h->resizeRows.assign(g.nlevels, 1);
        h->fuse2.assign(g.nlevels + 1, 0);
        for (int l = 1; l < g.nlevels; ++l) {
            ResizeTabs t;
            build_resize_tabs(g.lv[l - 1].w, g.lv[l - 1].h, g.lv[l].w, g.lv[l].h, t);
            // the pair (l - 1, l) in one launch: neither level is the exact 2:1 INTER_AREA case and the tile slices fit
            if (l >= 2 && h->quadOk[l - 1] != 2 && !t.area2)
                h->fuse2[l - 1] = resize2_fits(t.xs.data(), t.ys.data(), g.lv[l - 1].w, g.lv[l - 1].h, g.lv[l].w, g.lv[l].h) ? 1 : 0;
            h->xTabOff[l] = all.size(); all.insert(all.end(), t.xs.begin(), t.xs.end());
            h->yTabOff[l] = all.size(); all.insert(all.end(), t.ys.begin(), t.ys.end());
            while (all.size() % 16) all.push_back(0);
            h->qTabOff[l] = all.size(); all.insert(all.end(), t.quads.begin(), t.quads.end()); h->quadOk[l] = t.area2 ? 2 : (t.quadOk ? 1 : 0); h->resizeRows[l] = t.rows;
        }
        if (all.empty()) all.resize(4);
        SSVIO_HIP(hipMalloc(&h->d_tabs, all.size() * sizeof(int16_t)));
        SSVIO_HIP(hipMemcpy(h->d_tabs, all.data(), all.size() * sizeof(int16_t), hipMemcpyHostToDevice));
        h->geom = g;
        fill_klevels(g, h->kl);
    }
    if (batch > h->batchCap) {
        SSVIO_HIP(hipStreamSynchronize(h->stream));
        int16_t* tabs = h->d_tabs; h->d_tabs = nullptr;     // keep the tables
        orb_free_buffers(h);
        h->d_tabs = tabs;
        const PyramidGeom& g = h->geom;
        const size_t B = (size_t)batch;
        h->kpCap = g.maxQuotaSum;                                  // sum of the levels' capacities
        h->maxAlive = 0;
        for (int l = 0; l < g.nlevels; ++l) h->maxAlive = std::max(h->maxAlive, std::max(g.lv[l].quota + 3, 4 * g.lv[l].nIni));
        if (h->maxAlive > kQtMaxGlobal && !h->hostQuadtree) {     // beyond every GPU form of k_quadtree: said out loud, once per handle
            h->hostQuadtree = true;
            fprintf(stderr, "ssvio: a pyramid level asks for %d keypoints, more than the GPU quadtree holds (%d); DistributeOctTree "
                            "runs in the library's host stage for this extractor (every other stage stays on the GPU)\n", h->maxAlive - 3, kQtMaxGlobal - 3);
        }
        if (h->maxAlive > kQtMaxLds && !h->hostQuadtree) {        // node arrays of the large-quota form, one block per (frame, level)
            h->qtNodeStride = (quadtree_lds_bytes(h->maxAlive <= 8192 ? 8192 : kQtMaxGlobal) + 255) & ~(size_t)255;
            SSVIO_HIP(hipMalloc(&h->d_qtNode, B * g.nlevels * h->qtNodeStride));
        }
        SSVIO_HIP(hipMalloc(&h->d_pyr, B * g.frameBytes));
        SSVIO_HIP(hipMalloc(&h->d_blur, B * g.frameBytes));
        SSVIO_HIP(hipMemsetAsync(h->d_pyr, 0, B * g.frameBytes, h->stream));
        SSVIO_HIP(hipMemsetAsync(h->d_blur, 0, B * g.frameBytes, h->stream));
        SSVIO_HIP(hipMalloc(&h->d_raw, B * g.slotsPerFrame * sizeof(uint2)));
        SSVIO_HIP(hipMalloc(&h->d_rawCount, B * kMaxLevels * sizeof(int)));
        SSVIO_HIP(hipMalloc(&h->d_cellIni, B * g.cellsPerFrame * sizeof(int)));
        SSVIO_HIP(hipMalloc(&h->d_kval, B * g.slotsPerFrame * sizeof(uint32_t)));
        SSVIO_HIP(hipMalloc(&h->d_kord, B * g.slotsPerFrame * sizeof(uint32_t)));
        SSVIO_HIP(hipMalloc(&h->d_knode, B * g.slotsPerFrame * sizeof(uint32_t)));
        SSVIO_HIP(hipMalloc(&h->d_sel, B * h->kpCap * sizeof(uint2)));
        // k_describe fetches a slot's patches before it knows whether the slot is in use: every slot holds a position that is valid
        // in every level from the start (the quadtree only ever overwrites a slot with a keypoint of the slot's own level)
        SSVIO_HIP(hipMemsetD32Async((hipDeviceptr_t)h->d_sel, (int)((unsigned)kEdge | ((unsigned)kEdge << 16)), B * h->kpCap * 2, h->stream));
        SSVIO_HIP(hipMalloc(&h->d_selCount, B * kMaxLevels * sizeof(int)));
        SSVIO_HIP(hipMalloc(&h->d_kpCount, B * sizeof(int)));
        SSVIO_HIP(hipMalloc(&h->d_kpOut, B * h->kpCap * sizeof(ssvio_keypoint)));
        SSVIO_HIP(hipMalloc(&h->d_desc, B * h->kpCap * 32));
        SSVIO_HIP(hipHostMalloc(&h->h_kpCount, B * sizeof(int), hipHostMallocMapped));
        SSVIO_HIP(hipHostGetDevicePointer((void**)&h->d_kpCountHost, h->h_kpCount, 0));
        h->batchCap = batch;
    }
    if (inBytes > h->inBytes) {
        SSVIO_HIP(hipStreamSynchronize(h->stream));
        dfree(h->d_in);
        SSVIO_HIP(hipMalloc(&h->d_in, inBytes));
        h->inBytes = inBytes;
    }
    return SSVIO_OK;
}

// copy the raw candidate lists of the last run to the host (host quadtree mode, debug taps)
static int orb_fetch_raw(ssvio_orb* h, int batch)
{
    if (h->hostRawValid) return SSVIO_OK;
    const PyramidGeom& g = h->geom;
    h->h_rawCount.resize((size_t)batch * kMaxLevels); h->h_cellIni.resize((size_t)batch * g.cellsPerFrame);
    h->h_raw.resize((size_t)batch * g.slotsPerFrame);
    SSVIO_HIP(hipMemcpyAsync(h->h_rawCount.data(), h->d_rawCount, h->h_rawCount.size() * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    SSVIO_HIP(hipMemcpyAsync(h->h_cellIni.data(), h->d_cellIni, h->h_cellIni.size() * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    SSVIO_HIP(hipStreamSynchronize(h->stream));
    for (int f = 0; f < batch; ++f)
        for (int l = 0; l < g.nlevels; ++l) {
            const int n = h->h_rawCount[(size_t)f * kMaxLevels + l];
            if (n > 0) SSVIO_HIP(hipMemcpyAsync(h->h_raw.data() + (size_t)f * g.slotsPerFrame + g.lv[l].slotBase,
                                                h->d_raw + (size_t)f * g.slotsPerFrame + g.lv[l].slotBase, (size_t)n * sizeof(uint2),
                                                hipMemcpyDeviceToHost, h->stream));
        }
    SSVIO_HIP(hipStreamSynchronize(h->stream));
    h->hostRawValid = true;
    return SSVIO_OK;
}

// candidates of (frame, level) in the reference's append order: resolve the per-cell 20->7 fallback, sort by key
static void orb_host_candidates(ssvio_orb* h, int f, int l, std::vector<uint32_t>& vals)
{
    const PyramidGeom& g = h->geom;
    const int n = h->h_rawCount[(size_t)f * kMaxLevels + l];
    const uint2* r = h->h_raw.data() + (size_t)f * g.slotsPerFrame + g.lv[l].slotBase;
    const int* ini = h->h_cellIni.data() + (size_t)f * g.cellsPerFrame + g.lv[l].cellBase;
    std::vector<std::pair<uint32_t, uint32_t>> kv;
    for (int i = 0; i < n; ++i) {
        const int cell = (r[i].x >> 12) & 0x3FFFF, flags = r[i].x >> 30;
        if (ini[cell] ? (flags & 1) : (flags & 2)) kv.emplace_back(r[i].x & 0x3FFFFFFFu, r[i].y);
    }
    std::sort(kv.begin(), kv.end());
    vals.resize(kv.size());
    for (size_t i = 0; i < kv.size(); ++i) vals[i] = kv[i].second;
}

// host quadtree mode: DistributeOctTree per (frame, level) on the worker pool, src/Feature.cpp:897-909
static int orb_host_quadtree(ssvio_orb* h, int batch)
{
    const PyramidGeom& g = h->geom;
    const int L = g.nlevels;
    int rc = orb_fetch_raw(h, batch);
    if (rc) return rc;
    h->h_sel.assign((size_t)batch * h->kpCap, make_uint2((unsigned)kEdge | ((unsigned)kEdge << 16), 0));     // unused slots: a valid position (k_describe)
    h->h_selCount.assign((size_t)batch * kMaxLevels, 0);
    std::atomic<int> fail{0};
    auto job = [&](int j) {
        const int f = j / L, l = j % L;
        std::vector<uint32_t> c;
        orb_host_candidates(h, f, l, c);
        const int n = (int)c.size();
        if (n == 0) return;
        std::vector<float> xs(n), ys(n), rs(n);
        for (int i = 0; i < n; ++i) { xs[i] = (float)(c[i] & 0xFFF); ys[i] = (float)((c[i] >> 12) & 0xFFF); rs[i] = (float)(c[i] >> 24); }
        const LevelGeom& G = g.lv[l];
        std::vector<int> sel;
        if (distribute_octtree(xs.data(), ys.data(), rs.data(), n, kMinBorder, G.maxBX, kMinBorder, G.maxBY, G.quota, sel)) { fail.store(1); return; }
        if ((int)sel.size() > level_sel_cap(G.quota)) { fail.store(2); return; }
        uint2* out = h->h_sel.data() + (size_t)f * h->kpCap + G.selBase;
        for (size_t i = 0; i < sel.size(); ++i) {
            const uint32_t e = c[sel[i]];
            const uint32_t x = (e & 0xFFF) + kMinBorder, y = ((e >> 12) & 0xFFF) + kMinBorder;   // src/Feature.cpp:905-906
            out[i] = make_uint2(x | (y << 16), (uint32_t)l | ((e >> 24) << 8));
        }
        h->h_selCount[(size_t)f * kMaxLevels + l] = (int)sel.size();
    };
    h->pool->run(batch * L, job);
    if (fail.load()) { set_error("host quadtree failed (%d)", fail.load()); return SSVIO_ERR_GEOMETRY; }
    SSVIO_HIP(hipMemcpyAsync(h->d_sel, h->h_sel.data(), h->h_sel.size() * sizeof(uint2), hipMemcpyHostToDevice, h->stream));
    SSVIO_HIP(hipMemcpyAsync(h->d_selCount, h->h_selCount.data(), h->h_selCount.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    return SSVIO_OK;
}

// the launch sequence for `batch` frames whose level-0 images are already in d_pyr
static int orb_enqueue(ssvio_orb* h, int batch)
{
    const PyramidGeom& g = h->geom;
    hipStream_t s = h->stream;
    // d_rawCount / d_cellIni were cleared by k_level0
    // small launches (a frame or two per call, or the upper levels of a large frame): the chain of dependent launches is the cost,
    // so two levels are built per launch where the tables allow.  Above ~600 k pixels per launch the streaming kernel's
    // bandwidth wins (measured: 4K single frame 336 us unfused, 363 us with every pair fused).
    static const bool noFuse = getenv("SSVIO_NO_RESIZE2") != nullptr;
    for (int l = 1; l < g.nlevels;) {
        ProfScope p(h->prof, "k_resize", s);
        static const long fuseMaxPx = [] { const char* e = getenv("SSVIO_RESIZE2_MAX_PX"); return e ? atol(e) : 600000l; }();
        const bool fuse = !noFuse && (long)g.lv[l].w * g.lv[l].h * batch <= fuseMaxPx;
        if (fuse && l + 1 < g.nlevels && h->fuse2[l]) {
            launch_resize2(h->d_pyr, g.frameBytes, h->kl, l, h->d_tabs, h->xTabOff[l], h->yTabOff[l], h->xTabOff[l + 1], h->yTabOff[l + 1], batch, s);
            l += 2;
        } else {
            launch_resize(h->d_pyr, g.frameBytes, h->kl, l, h->d_tabs, h->xTabOff[l], h->yTabOff[l], h->qTabOff[l], (int)h->quadOk[l], h->resizeRows[l], batch, s);
            l += 1;
        }
    }
    // tall FAST tiles halve the per-tile fixed work; short ones give a small batch enough workgroups to fill the chip
    const bool tall = (long)g.fastTilesPerFrameT * batch >= 2048;
    // the blur depends on the pyramid only: it runs on the side stream beside k_quadtree (one workgroup per frame-level, which
    // leaves most of the chip idle) and, for batches that fill the chip, beside k_fast as well (VALU-bound, where the blur moves
    // bytes: 272.8 k -> 275.2 k frames/s at 256-frame steps; a single frame loses 2.5 % that way - its k_fast is on the critical
    // path and gives up CUs - so small batches fork after it).  k_describe waits for both.
    static const bool lateForkEnv = getenv("SSVIO_BLUR_AFTER_FAST") != nullptr;   // the one order of round 2, kept for A/B runs
    // (with event pairs around every kernel the fork also stays behind k_fast, so that its pair times k_fast alone)
    const bool lateFork = lateForkEnv || !tall || (h->prof.enabled && !h->prof.only);
    auto fork_blur = [&]() -> int {
        SSVIO_HIP(hipEventRecord(h->evFork, s));
        SSVIO_HIP(hipStreamWaitEvent(h->side, h->evFork, 0));
        {
            ProfScope p(h->prof, "k_blur", h->side);
            launch_blur(h->d_pyr, h->d_blur, g.frameBytes, h->kl, g.blurStripsPerFrame, batch, h->side);
        }
        SSVIO_HIP(hipEventRecord(h->evJoin, h->side));
        if (h->exportNow) {      // behind the blur on the side stream: nothing on the device waits for it, the call's end does
            int maxRows = 0;
            for (int l = 0; l < g.nlevels; ++l) maxRows = std::max(maxRows, g.lv[l].h + 2 * kEdge);
            ProfScope p(h->prof, "k_export_pyramid", h->side);
            launch_export_pyramid(h->d_pyr, h->kl, h->expLay, maxRows, h->d_pyrExp, h->side);
            SSVIO_HIP(hipEventRecord(h->evExp, h->side));
        }
        return SSVIO_OK;
    };
    if (!lateFork) { const int rc = fork_blur(); if (rc) return rc; }
    {
        ProfScope p(h->prof, "k_fast", s);
        launch_fast(h->d_pyr, g.frameBytes, h->kl, tall ? g.fastTilesPerFrameT : g.fastTilesPerFrame, tall, g.cellsPerFrame, g.slotsPerFrame, h->tab.iniTh, h->tab.minTh,
                    h->d_raw, h->d_rawCount, h->d_cellIni, batch, s);
    }
    if (lateFork) { const int rc = fork_blur(); if (rc) return rc; }
    if (!h->hostQuadtree) {
        QtArgs a{};
        a.kl = h->kl; a.raw = h->d_raw; a.rawCount = h->d_rawCount; a.cellIni = h->d_cellIni;
        a.rawPerFrame = g.slotsPerFrame; a.cellsPerFrame = g.cellsPerFrame;
        a.kval = h->d_kval; a.kord = h->d_kord; a.knode = h->d_knode;
        a.sel = h->d_sel; a.selCount = h->d_selCount; a.selPerFrame = h->kpCap; a.batch = batch;
        a.nodeScratch = h->d_qtNode; a.nodeScratchStride = h->qtNodeStride;
        ProfScope p(h->prof, "k_quadtree", s);
        if (launch_quadtree(a, h->maxAlive, g.nlevels, batch, s)) { set_error("internal: quadtree capacity"); return SSVIO_ERR_INVALID; }
    }
    if (h->hostQuadtree) { const int rc = orb_host_quadtree(h, batch); if (rc) return rc; }
    SSVIO_HIP(hipStreamWaitEvent(s, h->evJoin, 0));
    {
        ProfScope p(h->prof, "k_describe", s);
        launch_describe(h->d_pyr, h->d_blur, g.frameBytes, h->kl, h->d_sel, h->d_selCount, h->kpCap, h->kpCap, h->d_kpOut, h->d_desc,
                        h->d_kpCount, h->d_kpCountHost, h->hostKp, h->hostDesc, batch, s);
    }
    if (h->exportNow) SSVIO_HIP(hipStreamWaitEvent(s, h->evExp, 0));
    SSVIO_HIP(hipGetLastError());                                 // a refused launch configuration shows up only here
    return SSVIO_OK;
}

// the mapped host area of the exported pyramid for the current geometry
static int orb_ensure_export(ssvio_orb* h)
{
    const PyramidGeom& g = h->geom;
    if (h->expW == g.width && h->expH == g.height && h->h_pyrExp) return SSVIO_OK;
    size_t off = 0;
    for (int l = 0; l < g.nlevels; ++l) {
        h->expLay.stride[l] = (g.lv[l].w + 2 * kEdge + 63) & ~63;
        h->expLay.offset[l] = off;
        off += (size_t)h->expLay.stride[l] * (g.lv[l].h + 2 * kEdge);
        off = (off + 255) & ~(size_t)255;
    }
    if (off > h->pyrExpCap) {
        SSVIO_HIP(hipStreamSynchronize(h->stream));
        hfree(h->h_pyrExp); h->pyrExpCap = 0; h->d_pyrExp = nullptr;
        SSVIO_HIP(hipHostMalloc((void**)&h->h_pyrExp, off, hipHostMallocMapped));
        SSVIO_HIP(hipHostGetDevicePointer((void**)&h->d_pyrExp, h->h_pyrExp, 0));
        h->pyrExpCap = off;
    }
    if (!h->evExp) SSVIO_HIP(hipEventCreateWithFlags(&h->evExp, hipEventDisableTiming));
    h->expW = g.width; h->expH = g.height;
    return SSVIO_OK;
}

static bool graphs_enabled()
{
    static const bool on = [] { const char* e = getenv("SSVIO_GRAPH"); return e && e[0] == '1'; }();
    return on;
}

// The whole pipeline behind k_level0; everything is enqueued on the stream, the host only waits for the per-frame keypoint
// counts at the end.  With SSVIO_GRAPH=1 the ~12 launches and the fork / join with the side stream are captured once per
// (batch size, buffer generation) into a hipGraph and replayed.  Off by default: on this stack (ROCm 7.2, MI355X) the replay
// is slower than the plain launches (one VGA frame per call 158 vs 150 us, 64-frame steps 181 k vs 193 k frames/s,
// DESIGN.md section 4).  Never with the host quadtree (it waits in the middle) or while kernels are bracketed by profiling events.
static int orb_run(ssvio_orb* h, int batch, bool wait = true)
{
    hipStream_t s = h->stream;
    h->hostRawValid = false;
    bool launched = false;
    if ((graphs_enabled() || h->useGraph) && !h->hostQuadtree && !h->prof.enabled && !h->graphFailed && !h->hostKp && !h->exportNow) {
        if (!h->gexec || h->gBatch != batch || h->gGen != h->allocGen) {
            if (h->gexec) { (void)hipGraphExecDestroy(h->gexec); h->gexec = nullptr; }
            quadtree_prepare();                                   // function attributes are set outside the capture
            hipGraph_t graph = nullptr;
            if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                const int rc = orb_enqueue(h, batch);
                const hipError_t e = hipStreamEndCapture(s, &graph);
                if (rc == SSVIO_OK && e == hipSuccess && graph && hipGraphInstantiate(&h->gexec, graph, nullptr, nullptr, 0) == hipSuccess) {
                    h->gBatch = batch; h->gGen = h->allocGen;
                } else { h->gexec = nullptr; h->graphFailed = true; (void)hipGetLastError(); }
                if (graph) (void)hipGraphDestroy(graph);
            } else { h->graphFailed = true; (void)hipGetLastError(); }
        }
        if (h->gexec) { SSVIO_HIP(hipGraphLaunch(h->gexec, s)); launched = true; }
    }
    if (!launched) { const int rc = orb_enqueue(h, batch); if (rc) return rc; }
    // the counts reach the host through k_describe's stores into the pinned, device-mapped h_kpCount
    h->lastBatch = batch; h->haveResults = true; h->pending = true;
    if (wait) { SSVIO_HIP(hipStreamSynchronize(s)); h->pending = false; }
    return SSVIO_OK;
}

// host-visible results (h_kpCount, profiler records) are complete after this
static int orb_wait(ssvio_orb* h)
{
    if (!h->pending) return SSVIO_OK;
    SSVIO_HIP(hipSetDevice(h->device));
    SSVIO_HIP(hipStreamSynchronize(h->stream));
    h->pending = false;
    h->prof.resolve();
    return SSVIO_OK;
}

static int orb_check(ssvio_orb* h) { if (!h) { set_error("null extractor handle"); return SSVIO_ERR_INVALID; } return SSVIO_OK; }

extern "C" {

const char* ssvio_last_error(void) { return last_error_cstr(); }
int ssvio_version(void) { return 200; }
void ssvio_test_fail_next_allocs(int n) { g_failAllocs.store(n > 0 ? n : 0); }
int ssvio_device_count(int* count)
{
    if (!count) { set_error("null count"); return SSVIO_ERR_INVALID; }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
    return SSVIO_OK;
}

int ssvio_orb_create(const ssvio_orb_params* params, int device, ssvio_orb** out)
{
    if (!params || !out) { set_error("null argument"); return SSVIO_ERR_INVALID; }
    *out = nullptr;
    ExtractorTables t;
    int rc = build_tables(*params, t);
    if (rc) return rc;
    rc = pick_device(device);
    if (rc) return rc;
    SSVIO_HIP(hipSetDevice(device));
    ssvio_orb* h = new ssvio_orb();
    h->device = device; h->tab = t;
    if (hipStreamCreateWithFlags(&h->own, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->evCand, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&h->evFork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->evJoin, hipEventDisableTiming) != hipSuccess) {
        set_error("cannot create HIP stream/event"); delete h; return SSVIO_ERR_HIP;
    }
    h->stream = h->own;
    h->pool.reset(new WorkerPool(host_threads() - 1));
    if (const char* e = std::getenv("SSVIO_QUADTREE")) h->hostQuadtree = std::string(e) == "host";
    *out = h;
    return SSVIO_OK;
}

void ssvio_orb_destroy(ssvio_orb* h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    // a borrowed stream (ssvio_orb_set_stream) may already be gone: wait for the device instead of touching it
    if (h->stream == h->own) (void)hipStreamSynchronize(h->stream); else (void)hipDeviceSynchronize();
    if (h->gexec) { (void)hipGraphExecDestroy(h->gexec); h->gexec = nullptr; }
    h->prof.resolve(); h->prof.destroy();
    orb_free_buffers(h);
    dfree(h->d_tabs);
    if (h->evCand) (void)hipEventDestroy(h->evCand);
    if (h->evFork) (void)hipEventDestroy(h->evFork);
    if (h->evJoin) (void)hipEventDestroy(h->evJoin);
    if (h->evExp) (void)hipEventDestroy(h->evExp);
    hfree(h->h_pyrExp);
    if (h->side) (void)hipStreamDestroy(h->side);
    if (h->own) (void)hipStreamDestroy(h->own);
    delete h;
}

int ssvio_orb_get_tables(ssvio_orb* h, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2, int32_t* fpl)
{
    if (orb_check(h)) return SSVIO_ERR_INVALID;
    for (int i = 0; i < h->tab.nlevels; ++i) {
        if (scale) scale[i] = h->tab.scale[i];
        if (inv_scale) inv_scale[i] = h->tab.invScale[i];
        if (sigma2) sigma2[i] = h->tab.sigma2[i];
        if (inv_sigma2) inv_sigma2[i] = h->tab.invSigma2[i];
        if (fpl) fpl[i] = h->tab.quota[i];
    }
    return SSVIO_OK;
}

int ssvio_orb_max_keypoints(ssvio_orb* h)
{
    if (orb_check(h)) return SSVIO_ERR_INVALID;
    int s = 0;
    for (int i = 0; i < h->tab.nlevels; ++i) s += level_sel_cap(h->tab.quota[i]);
    return s;
}

int ssvio_orb_level_geometry(ssvio_orb* h, int width, int height, int level, int32_t* lw, int32_t* lh,
                             int32_t* ncols, int32_t* nrows, int32_t* wcell, int32_t* hcell)
{
    if (orb_check(h)) return SSVIO_ERR_INVALID;
    if (level < 0 || level >= h->tab.nlevels) { set_error("level out of range"); return SSVIO_ERR_INVALID; }
    PyramidGeom g;
    const int rc = build_geometry(h->tab, width, height, g);
    if (rc) return rc;
    const LevelGeom& L = g.lv[level];
    if (lw) *lw = L.w; if (lh) *lh = L.h; if (ncols) *ncols = L.nCols; if (nrows) *nrows = L.nRows;
    if (wcell) *wcell = L.wCell; if (hcell) *hcell = L.hCell;
    return SSVIO_OK;
}

static int orb_extract_dev(ssvio_orb* h, const void* d_imgs, int batch, int channels,
                           int width, int height, int stride, size_t frame_stride, bool wait)
{
    if (orb_check(h)) return SSVIO_ERR_INVALID;
    if (channels != 1 && channels != 3) { set_error("channels must be 1 or 3"); return SSVIO_ERR_INVALID; }
    if (batch < 0) { set_error("negative batch"); return SSVIO_ERR_INVALID; }
    if (!d_imgs || width <= 0 || height <= 0 || batch == 0) { h->lastBatch = 0; h->haveResults = false; return SSVIO_OK; }   // src/Feature.cpp:507-508
    if (stride < width * channels) { set_error("stride %d < row bytes %d", stride, width * channels); return SSVIO_ERR_INVALID; }
    int rc = orb_ensure(h, width, height, batch, 0);
    if (rc) return rc;
    const PyramidGeom& g = h->geom;
    {
        ProfScope p(h->prof, channels == 3 ? "k_rgb2gray" : "k_copy_gray", h->stream);
        if (channels == 3) launch_rgb2gray((const uint8_t*)d_imgs, width, height, stride, frame_stride, batch, h->d_pyr + g.lv[0].offset, g.frameBytes, g.lv[0].pitch,
                                  h->d_rawCount, batch * kMaxLevels, h->d_cellIni, batch * g.cellsPerFrame, h->stream);
        else launch_copy_gray((const uint8_t*)d_imgs, width, height, stride, frame_stride, batch, h->d_pyr + g.lv[0].offset, g.frameBytes, g.lv[0].pitch,
                                  h->d_rawCount, batch * kMaxLevels, h->d_cellIni, batch * g.cellsPerFrame, h->stream);
        SSVIO_HIP(hipGetLastError());
    }
    rc = orb_run(h, batch, wait);
    if (rc) return rc;
    if (wait) h->prof.resolve();
    return SSVIO_OK;
}

int ssvio_orb_extract_batch_dev(ssvio_orb* h, const void* d_imgs, int batch, int channels,
                                int width, int height, int stride, size_t frame_stride)
{ return orb_extract_dev(h, d_imgs, batch, channels, width, height, stride, frame_stride, true); }

int ssvio_orb_extract_batch_dev_async(ssvio_orb* h, const void* d_imgs, int batch, int channels,
                                      int width, int height, int stride, size_t frame_stride)
{ return orb_extract_dev(h, d_imgs, batch, channels, width, height, stride, frame_stride, false); }

int ssvio_orb_sync(ssvio_orb* h)
{
    if (orb_check(h)) return SSVIO_ERR_INVALID;
    return orb_wait(h);
}

int ssvio_orb_get_stream(ssvio_orb* h, void** hip_stream)
{
    if (orb_check(h) || !hip_stream) { set_error("null argument"); return SSVIO_ERR_INVALID; }
    *hip_stream = (void*)h->stream;
    return SSVIO_OK;
}

int ssvio_orb_fetch_results(ssvio_orb* h, int batch, ssvio_keypoint* kps, uint8_t* desc, int cap, int* n)
{
    if (orb_check(h)) return SSVIO_ERR_INVALID;
    if (!n) { set_error("null count array"); return SSVIO_ERR_INVALID; }
    { const int rc = orb_wait(h); if (rc) return rc; }
    if (batch > h->lastBatch || !h->haveResults) {
        if (h->lastBatch == 0) { for (int f = 0; f < batch; ++f) n[f] = 0; return SSVIO_OK; }
        set_error("no results for %d frames (last call had %d)", batch, h->lastBatch); return SSVIO_ERR_INVALID;
    }
    SSVIO_HIP(hipSetDevice(h->device));
    for (int f = 0; f < batch; ++f) {
        const int c = h->h_kpCount[f];
        if (c > cap) { set_error("frame %d has %d keypoints, caller capacity is %d", f, c, cap); return SSVIO_ERR_CAPACITY; }
    }
    for (int f = 0; f < batch; ++f) {
        const int c = h->h_kpCount[f];
        n[f] = c;
        if (c == 0) continue;
        if (kps) SSVIO_HIP(hipMemcpyAsync(kps + (size_t)f * cap, h->d_kpOut + (size_t)f * h->kpCap, (size_t)c * sizeof(ssvio_keypoint), hipMemcpyDeviceToHost, h->stream));
        if (desc) SSVIO_HIP(hipMemcpyAsync(desc + (size_t)f * cap * 32, h->d_desc + (size_t)f * h->kpCap * 32, (size_t)c * 32, hipMemcpyDeviceToHost, h->stream));
    }
    SSVIO_HIP(hipStreamSynchronize(h->stream));
    return SSVIO_OK;
}

int ssvio_orb_last_counts(ssvio_orb* h, int batch, int* n)
{
    if (orb_check(h) || !n) return SSVIO_ERR_INVALID;
    { const int rc = orb_wait(h); if (rc) return rc; }
    for (int f = 0; f < batch; ++f) n[f] = (h->haveResults && f < h->lastBatch) ? h->h_kpCount[f] : 0;
    return SSVIO_OK;
}

int ssvio_orb_device_results(ssvio_orb* h, const void** d_kps, const void** d_desc, const int32_t** d_n, int* cap_per_frame)
{
    if (orb_check(h)) return SSVIO_ERR_INVALID;
    if (!h->haveResults) { set_error("no device results yet"); return SSVIO_ERR_INVALID; }
    if (d_kps) *d_kps = h->d_kpOut; if (d_desc) *d_desc = h->d_desc; if (d_n) *d_n = h->d_kpCount; if (cap_per_frame) *cap_per_frame = h->kpCap;
    return SSVIO_OK;
}

int ssvio_orb_extract_batch(ssvio_orb* h, const uint8_t* imgs, int batch, int channels, int width, int height, int stride,
                            size_t frame_stride, ssvio_keypoint* kps, uint8_t* desc, int cap, int* n)
{
    if (orb_check(h)) return SSVIO_ERR_INVALID;
    if (channels != 1 && channels != 3) { set_error("channels must be 1 or 3"); return SSVIO_ERR_INVALID; }
    if (!n) { set_error("null count array"); return SSVIO_ERR_INVALID; }
    if (batch < 0) { set_error("negative batch"); return SSVIO_ERR_INVALID; }
    if (!imgs || width <= 0 || height <= 0 || batch == 0) { for (int f = 0; f < batch; ++f) n[f] = 0; h->lastBatch = 0; h->haveResults = false; return SSVIO_OK; }
    if (stride < width * channels) { set_error("stride %d < row bytes %d", stride, width * channels); return SSVIO_ERR_INVALID; }
    const size_t rowBytes = (size_t)width * channels;
    int rc = orb_ensure(h, width, height, batch, (size_t)batch * rowBytes * height);
    if (rc) return rc;
    const PyramidGeom& g = h->geom;
    // Small inputs (a frame or a few): the rows are packed by the host into pinned, device-mapped memory and k_level0 reads them from
    // there -- no copy command, and a strided cv::Mat ROI costs nothing extra.  Larger batches: one copy per frame; strided rows are
    // packed on the host first, because a 2-D copy from pageable memory costs a transfer per row.
    const size_t inBytes = (size_t)batch * rowBytes * height;
    static const bool noMapped = getenv("SSVIO_NO_MAPPED_INPUT") != nullptr;
    const bool mappedIn = !noMapped && inBytes <= ((size_t)4 << 20);
    const uint8_t* d_src = h->d_in;
    if (mappedIn) {
        if (inBytes > h->inMappedCap) {
            SSVIO_HIP(hipStreamSynchronize(h->stream));
            hfree(h->h_inMapped); h->inMappedCap = 0; h->d_inMapped = nullptr;
            SSVIO_HIP(hipHostMalloc((void**)&h->h_inMapped, inBytes, hipHostMallocMapped));
            SSVIO_HIP(hipHostGetDevicePointer((void**)&h->d_inMapped, h->h_inMapped, 0));
            h->inMappedCap = inBytes;
        }
        for (int f = 0; f < batch; ++f) {
            const uint8_t* src = imgs + (size_t)f * frame_stride;
            uint8_t* dst = h->h_inMapped + (size_t)f * rowBytes * height;
            if ((size_t)stride == rowBytes) memcpy(dst, src, rowBytes * height);
            else for (int y = 0; y < height; ++y) memcpy(dst + (size_t)y * rowBytes, src + (size_t)y * stride, rowBytes);
        }
        d_src = h->d_inMapped;
    } else
    for (int f = 0; f < batch; ++f) {
        const uint8_t* src = imgs + (size_t)f * frame_stride;
        if ((size_t)stride != rowBytes) {
            h->stageIn.resize(rowBytes * height);
            for (int y = 0; y < height; ++y) memcpy(h->stageIn.data() + (size_t)y * rowBytes, src + (size_t)y * stride, rowBytes);
            src = h->stageIn.data();
        }
        SSVIO_HIP(hipMemcpyAsync(h->d_in + (size_t)f * rowBytes * height, src, rowBytes * height, hipMemcpyHostToDevice, h->stream));
        if ((size_t)stride != rowBytes) SSVIO_HIP(hipStreamSynchronize(h->stream));       // stageIn is reused by the next frame
    }
    {
        ProfScope p(h->prof, channels == 3 ? "k_rgb2gray" : "k_copy_gray", h->stream);
        if (channels == 3) launch_rgb2gray(d_src, width, height, (int)rowBytes, rowBytes * height, batch, h->d_pyr + g.lv[0].offset, g.frameBytes, g.lv[0].pitch,
                                  h->d_rawCount, batch * kMaxLevels, h->d_cellIni, batch * g.cellsPerFrame, h->stream);
        else launch_copy_gray(d_src, width, height, (int)rowBytes, rowBytes * height, batch, h->d_pyr + g.lv[0].offset, g.frameBytes, g.lv[0].pitch,
                                  h->d_rawCount, batch * kMaxLevels, h->d_cellIni, batch * g.cellsPerFrame, h->stream);
        SSVIO_HIP(hipGetLastError());
    }
    // Results come back with the kernels: k_describe also stores every frame's slot range into mapped host memory, the host waits
    // once and copies out what the counts say (a second device trip after reading the counts costs a wait more).
    const size_t kpBytes = align256((size_t)batch * h->kpCap * sizeof(ssvio_keypoint)), dBytes = (size_t)batch * h->kpCap * 32;
    if (!kps || !desc || kpBytes + dBytes > ((size_t)64 << 20)) {
        rc = orb_run(h, batch);
        if (rc) return rc;
        rc = ssvio_orb_fetch_results(h, batch, kps, desc, cap, n);
        h->prof.resolve();
        return rc;
    }
    if (kpBytes + dBytes > h->landingCap) {
        SSVIO_HIP(hipStreamSynchronize(h->stream));
        hfree(h->h_landing); h->landingCap = 0; h->d_landing = nullptr;
        SSVIO_HIP(hipHostMalloc((void**)&h->h_landing, kpBytes + dBytes, hipHostMallocMapped));
        SSVIO_HIP(hipHostGetDevicePointer((void**)&h->d_landing, h->h_landing, 0));
        h->landingCap = kpBytes + dBytes;
    }
    h->exportValid = false;
    if (h->exportPyr && batch == 1) { rc = orb_ensure_export(h); if (rc) return rc; h->exportNow = true; }
    h->hostKp = (ssvio_keypoint*)h->d_landing; h->hostDesc = h->d_landing + kpBytes;
    rc = orb_run(h, batch, false);
    h->hostKp = nullptr; h->hostDesc = nullptr;
    const bool exported = h->exportNow; h->exportNow = false;
    if (rc) return rc;
    rc = orb_wait(h);
    if (rc) return rc;
    h->exportValid = exported;
    for (int f = 0; f < batch; ++f) {
        const int c = h->h_kpCount[f];
        if (c > cap) { set_error("frame %d has %d keypoints, caller capacity is %d", f, c, cap); return SSVIO_ERR_CAPACITY; }
    }
    for (int f = 0; f < batch; ++f) {
        const int c = h->h_kpCount[f];
        n[f] = c;
        memcpy(kps + (size_t)f * cap, h->h_landing + (size_t)f * h->kpCap * sizeof(ssvio_keypoint), (size_t)c * sizeof(ssvio_keypoint));
        memcpy(desc + (size_t)f * cap * 32, h->h_landing + kpBytes + (size_t)f * h->kpCap * 32, (size_t)c * 32);
    }
    return SSVIO_OK;
}

int ssvio_orb_extract(ssvio_orb* h, const uint8_t* gray, int width, int height, int stride,
                      ssvio_keypoint* kps, uint8_t* desc, int cap, int* n)
{
    return ssvio_orb_extract_batch(h, gray, 1, 1, width, height, stride, 0, kps, desc, cap, n);
}

int ssvio_orb_extract_rgb(ssvio_orb* h, const uint8_t* rgb, int width, int height, int stride,
                          ssvio_keypoint* kps, uint8_t* desc, int cap, int* n)
{
    return ssvio_orb_extract_batch(h, rgb, 1, 3, width, height, stride, 0, kps, desc, cap, n);
}

int ssvio_orb_debug_level_image(ssvio_orb* h, int frame, int level, int blurred, uint8_t* dst, int dst_stride)
{
    if (orb_check(h)) return SSVIO_ERR_INVALID;
    if (!h->haveResults || frame < 0 || frame >= h->lastBatch || level < 0 || level >= h->geom.nlevels || !dst) { set_error("bad frame/level or no results"); return SSVIO_ERR_INVALID; }
    const LevelGeom& L = h->geom.lv[level];
    if (dst_stride < L.w) { set_error("dst_stride too small"); return SSVIO_ERR_INVALID; }
    SSVIO_HIP(hipSetDevice(h->device));
    if (blurred) {      // stored as 32-px-wide column panels (geometry.cpp): fetched whole, put back into rows here
        const int np = (L.w + 31) / 32;
        std::vector<uint8_t> tmp((size_t)np * L.blurPanelStride);
        SSVIO_HIP(hipMemcpyAsync(tmp.data(), h->d_blur + (size_t)frame * h->geom.frameBytes + (L.offset - 64), tmp.size(), hipMemcpyDeviceToHost, h->stream));
        SSVIO_HIP(hipStreamSynchronize(h->stream));
        for (int y = 0; y < L.h; ++y)
            for (int x = 0; x < L.w; ++x) dst[(size_t)y * dst_stride + x] = tmp[(size_t)(x >> 5) * L.blurPanelStride + 32 * y + (x & 31)];
        return SSVIO_OK;
    }
    const uint8_t* src = h->d_pyr + (size_t)frame * h->geom.frameBytes + L.offset;
    SSVIO_HIP(hipMemcpy2DAsync(dst, dst_stride, src, L.pitch, L.w, L.h, hipMemcpyDeviceToHost, h->stream));
    SSVIO_HIP(hipStreamSynchronize(h->stream));
    return SSVIO_OK;
}

static inline int refl101_host(int p, int n) { if (n == 1) return 0; while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p; return p; }

int ssvio_orb_pyramid_level(ssvio_orb* h, int frame, int level, uint8_t* dst, int dst_stride, int* lw, int* lh)
{
    if (orb_check(h)) return SSVIO_ERR_INVALID;
    if (!h->haveResults || frame < 0 || frame >= h->lastBatch || level < 0 || level >= h->geom.nlevels) { set_error("bad frame/level or no results"); return SSVIO_ERR_INVALID; }
    const LevelGeom& L = h->geom.lv[level];
    if (lw) *lw = L.w; if (lh) *lh = L.h;
    if (!dst) return SSVIO_OK;
    if (dst_stride < L.w + 2 * kEdge) { set_error("dst_stride too small"); return SSVIO_ERR_INVALID; }
    const int rc = ssvio_orb_debug_level_image(h, frame, level, 0, dst + (size_t)kEdge * dst_stride + kEdge, dst_stride);
    if (rc) return rc;
    // copyMakeBorder(BORDER_REFLECT_101), src/Feature.cpp:815-821
    for (int y = 0; y < L.h; ++y) {
        uint8_t* row = dst + (size_t)(y + kEdge) * dst_stride + kEdge;
        for (int i = 1; i <= kEdge; ++i) { row[-i] = row[refl101_host(-i, L.w)]; row[L.w - 1 + i] = row[refl101_host(L.w - 1 + i, L.w)]; }
    }
    for (int i = 1; i <= kEdge; ++i) {
        memcpy(dst + (size_t)(kEdge - i) * dst_stride, dst + (size_t)(kEdge + refl101_host(-i, L.h)) * dst_stride, (size_t)L.w + 2 * kEdge);
        memcpy(dst + (size_t)(kEdge + L.h - 1 + i) * dst_stride, dst + (size_t)(kEdge + refl101_host(L.h - 1 + i, L.h)) * dst_stride, (size_t)L.w + 2 * kEdge);
    }
    return SSVIO_OK;
}

int ssvio_orb_set_graph(ssvio_orb* h, int enable)
{
    if (orb_check(h)) return SSVIO_ERR_INVALID;
    h->useGraph = enable != 0;
    return SSVIO_OK;
}

int ssvio_orb_set_export_pyramid(ssvio_orb* h, int enable)
{
    if (orb_check(h)) return SSVIO_ERR_INVALID;
    h->exportPyr = enable != 0;
    if (!h->exportPyr) h->exportValid = false;
    return SSVIO_OK;
}

int ssvio_orb_exported_pyramid_level(ssvio_orb* h, int level, const uint8_t** base, int* stride, int* lw, int* lh)
{
    if (orb_check(h)) return SSVIO_ERR_INVALID;
    if (!h->exportValid || !h->haveResults || level < 0 || level >= h->geom.nlevels || !base || !stride) {
        set_error("no exported pyramid (ssvio_orb_set_export_pyramid + a single-frame ssvio_orb_extract[_rgb] call) or bad level"); return SSVIO_ERR_INVALID;
    }
    *base = h->h_pyrExp + h->expLay.offset[level]; *stride = h->expLay.stride[level];
    if (lw) *lw = h->geom.lv[level].w; if (lh) *lh = h->geom.lv[level].h;
    return SSVIO_OK;
}

int ssvio_orb_debug_candidates(ssvio_orb* h, int frame, int level, int32_t* xs, int32_t* ys, int32_t* resp, int cap, int* n)
{
    if (orb_check(h)) return SSVIO_ERR_INVALID;
    if (!h->haveResults || frame < 0 || frame >= h->lastBatch || level < 0 || level >= h->geom.nlevels || !n) { set_error("bad frame/level or no results"); return SSVIO_ERR_INVALID; }
    SSVIO_HIP(hipSetDevice(h->device));
    const int rc = orb_fetch_raw(h, h->lastBatch);
    if (rc) return rc;
    std::vector<uint32_t> e;
    orb_host_candidates(h, frame, level, e);
    const int c = (int)e.size();
    *n = c;
    if (c > cap) { set_error("capacity %d < %d candidates", cap, c); return SSVIO_ERR_CAPACITY; }
    for (int i = 0; i < c; ++i) { if (xs) xs[i] = e[i] & 0xFFF; if (ys) ys[i] = (e[i] >> 12) & 0xFFF; if (resp) resp[i] = e[i] >> 24; }
    return SSVIO_OK;
}

int ssvio_distribute_octtree(const float* xs, const float* ys, const float* resp, int n, int minX, int maxX, int minY, int maxY,
                             int N, int32_t* out_idx, int cap, int* nout)
{
    if (!nout || (n > 0 && (!xs || !ys || !resp))) { set_error("null argument"); return SSVIO_ERR_INVALID; }
    if (maxX <= minX || maxY <= minY) { set_error("empty region"); return SSVIO_ERR_INVALID; }
    std::vector<int> out;
    if (n > 0) { const int rc = distribute_octtree(xs, ys, resp, n, minX, maxX, minY, maxY, N, out); if (rc) return rc; }
    *nout = (int)out.size();
    if ((int)out.size() > cap) { set_error("capacity %d < %zu", cap, out.size()); return SSVIO_ERR_CAPACITY; }
    for (size_t i = 0; i < out.size(); ++i) out_idx[i] = out[i];
    return SSVIO_OK;
}

int ssvio_orb_set_profiling(ssvio_orb* h, int enable)
{
    if (orb_check(h)) return SSVIO_ERR_INVALID;
    SSVIO_HIP(hipSetDevice(h->device));
    SSVIO_HIP(hipStreamSynchronize(h->stream));
    h->prof.clear();
    h->prof.enabled = enable != 0;
    return SSVIO_OK;
}

int ssvio_orb_kernel_times(ssvio_orb* h, const char** names, double* total_ms, int32_t* launches, int cap, int* count)
{
    if (orb_check(h)) return SSVIO_ERR_INVALID;
    SSVIO_HIP(hipSetDevice(h->device));
    SSVIO_HIP(hipStreamSynchronize(h->stream));
    return prof_out(h->prof, names, total_ms, launches, cap, count);
}

int ssvio_orb_profile_only(ssvio_orb* h, const char* kernel)
{
    if (orb_check(h)) return SSVIO_ERR_INVALID;
    h->profOnly = kernel ? kernel : "";
    h->prof.only = h->profOnly.empty() ? nullptr : h->profOnly.c_str();
    return SSVIO_OK;
}

int ssvio_orb_set_stream(ssvio_orb* h, void* hip_stream)
{
    if (orb_check(h)) return SSVIO_ERR_INVALID;
    SSVIO_HIP(hipSetDevice(h->device));
    SSVIO_HIP(hipStreamSynchronize(h->stream));
    h->stream = hip_stream ? (hipStream_t)hip_stream : h->own;
    return SSVIO_OK;
}

}  // extern "C"
