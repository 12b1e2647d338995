// capi.hip -- the C ABI of include/ssvio_gpu.h: handles, device memory, streams and
// the per-batch orchestration of the gfx950 kernels.
#include "common.h"
#include "match.h"
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

static int pick_device(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        set_error("no HIP device visible: this library has no CPU fallback");
        return SSVIO_ERR_NOGPU;
    }
    if (device < 0 || device >= n) { set_error("device %d out of range (0..%d)", device, n - 1); return SSVIO_ERR_INVALID; }
    return SSVIO_OK;
}

template <typename T> static void dfree(T*& p) { if (p) { (void)hipFree(p); p = nullptr; } }
template <typename T> static void hfree(T*& p) { if (p) { (void)hipHostFree(p); p = nullptr; } }

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
    KernelProfiler prof; std::string profOnly;
    std::unique_ptr<WorkerPool> pool;
    int lastBatch = 0;
    bool haveResults = false;
    // the launch sequence behind k_level0 as a hipGraph (one per batch size and buffer generation), replayed by later calls
    hipGraphExec_t gexec = nullptr; int gBatch = 0; unsigned gGen = 0, allocGen = 1; bool graphFailed = false;
};

static void orb_free_buffers(ssvio_orb* h)
{
    dfree(h->d_in); dfree(h->d_pyr); dfree(h->d_blur); dfree(h->d_tabs);
    dfree(h->d_raw); dfree(h->d_rawCount); dfree(h->d_cellIni); dfree(h->d_kval); dfree(h->d_kord); dfree(h->d_knode);
    dfree(h->d_sel); dfree(h->d_selCount); dfree(h->d_kpCount); dfree(h->d_kpOut); dfree(h->d_desc); dfree(h->d_qtNode);
    hfree(h->h_kpCount);
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
        K.fastTileBase = L.fastTileBase; K.fastTilesX = L.fastTilesX; K.fastTileBaseT = L.fastTileBaseT;
        K.blurStripBase = L.blurStripBase; K.blurStripsX = L.blurStripsX; K.blurIters = L.blurIters;
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
        h->geom = g;
        fill_klevels(g, h->kl);
        // resize tables
        std::vector<int16_t> all;
        h->xTabOff.assign(g.nlevels, 0); h->yTabOff.assign(g.nlevels, 0); h->qTabOff.assign(g.nlevels, 0); h->quadOk.assign(g.nlevels, 0); h->resizeRows.assign(g.nlevels, 1);
        for (int l = 1; l < g.nlevels; ++l) {
            ResizeTabs t;
            build_resize_tabs(g.lv[l - 1].w, g.lv[l - 1].h, g.lv[l].w, g.lv[l].h, t);
            h->xTabOff[l] = all.size(); all.insert(all.end(), t.xs.begin(), t.xs.end());
            h->yTabOff[l] = all.size(); all.insert(all.end(), t.ys.begin(), t.ys.end());
            while (all.size() % 16) all.push_back(0);
            h->qTabOff[l] = all.size(); all.insert(all.end(), t.quads.begin(), t.quads.end()); h->quadOk[l] = t.area2 ? 2 : (t.quadOk ? 1 : 0); h->resizeRows[l] = t.rows;
        }
        if (all.empty()) all.resize(4);
        SSVIO_HIP(hipMalloc(&h->d_tabs, all.size() * sizeof(int16_t)));
        SSVIO_HIP(hipMemcpy(h->d_tabs, all.data(), all.size() * sizeof(int16_t), hipMemcpyHostToDevice));
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
    h->h_sel.assign((size_t)batch * h->kpCap, make_uint2(0, 0));
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
    for (int l = 1; l < g.nlevels; ++l) {
        ProfScope p(h->prof, "k_resize", s);
        launch_resize(h->d_pyr, g.frameBytes, h->kl, l, h->d_tabs, h->xTabOff[l], h->yTabOff[l], h->qTabOff[l], (int)h->quadOk[l], h->resizeRows[l], batch, s);
    }
    {
        ProfScope p(h->prof, "k_fast", s);
        // tall tiles halve the per-tile fixed work; short ones give a small batch enough workgroups to fill the chip
        const bool tall = (long)g.fastTilesPerFrameT * batch >= 2048;
        launch_fast(h->d_pyr, g.frameBytes, h->kl, tall ? g.fastTilesPerFrameT : g.fastTilesPerFrame, tall, g.cellsPerFrame, g.slotsPerFrame, h->tab.iniTh, h->tab.minTh,
                    h->d_raw, h->d_rawCount, h->d_cellIni, batch, s);
    }
    // the blur depends on the pyramid only: it runs on the side stream beside k_quadtree (one workgroup per frame-level,
    // which leaves most of the chip idle), and k_describe waits for both
    SSVIO_HIP(hipEventRecord(h->evFork, s));
    SSVIO_HIP(hipStreamWaitEvent(h->side, h->evFork, 0));
    {
        ProfScope p(h->prof, "k_blur", h->side);
        launch_blur(h->d_pyr, h->d_blur, g.frameBytes, h->kl, g.blurStripsPerFrame, batch, h->side);
    }
    SSVIO_HIP(hipEventRecord(h->evJoin, h->side));
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
                        h->d_kpCount, h->d_kpCountHost, batch, s);
    }
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
    if (graphs_enabled() && !h->hostQuadtree && !h->prof.enabled && !h->graphFailed) {
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
int ssvio_version(void) { return 100; }
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
    if (h->gexec) { (void)hipSetDevice(h->device); (void)hipStreamSynchronize(h->stream); (void)hipGraphExecDestroy(h->gexec); h->gexec = nullptr; }
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    h->prof.resolve(); h->prof.destroy();
    orb_free_buffers(h);
    dfree(h->d_tabs);
    if (h->evCand) (void)hipEventDestroy(h->evCand);
    if (h->evFork) (void)hipEventDestroy(h->evFork);
    if (h->evJoin) (void)hipEventDestroy(h->evJoin);
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
    // continuous frames (the usual cv::Mat) are one copy each; strided rows (an ROI) are packed on the host first, because a
    // 2-D copy from pageable memory costs a transfer per row
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
        if (channels == 3) launch_rgb2gray(h->d_in, width, height, (int)rowBytes, rowBytes * height, batch, h->d_pyr + g.lv[0].offset, g.frameBytes, g.lv[0].pitch,
                                  h->d_rawCount, batch * kMaxLevels, h->d_cellIni, batch * g.cellsPerFrame, h->stream);
        else launch_copy_gray(h->d_in, width, height, (int)rowBytes, rowBytes * height, batch, h->d_pyr + g.lv[0].offset, g.frameBytes, g.lv[0].pitch,
                                  h->d_rawCount, batch * kMaxLevels, h->d_cellIni, batch * g.cellsPerFrame, h->stream);
    }
    rc = orb_run(h, batch);
    if (rc) return rc;
    rc = ssvio_orb_fetch_results(h, batch, kps, desc, cap, n);
    h->prof.resolve();
    return rc;
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
    const uint8_t* src = (blurred ? h->d_blur : h->d_pyr) + (size_t)frame * h->geom.frameBytes + L.offset;
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

static int prof_out(KernelProfiler& p, const char** names, double* total_ms, int32_t* launches, int cap, int* count)
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

// =====================================================================================
struct ssvio_matcher {
    int device = 0;
    int threshold = 50; float nnRatio = 0.6f;
    hipStream_t own = nullptr, stream = nullptr;
    uint8_t* d_q = nullptr; size_t qRows = 0;
    uint8_t* d_t = nullptr; size_t tRows = 0;
    int* d_counts2 = nullptr;            // {nq, nt} of the single-pair entry points
    uint2* d_partial = nullptr; size_t partialCap = 0;
    uint2* d_partialRev = nullptr; size_t partialRevCap = 0;
    ssvio_dmatch* d_out = nullptr; size_t outCap = 0;
    int32_t* d_raw = nullptr; size_t rawCap = 0;
    int* d_nout = nullptr; size_t noutCap = 0;
    int* d_pairs = nullptr; size_t pairsCap = 0;
    double* d_p3d = nullptr; size_t p3dCap = 0; double* d_p2d = nullptr; size_t p2dCap = 0;     // MatchByProject inputs
    ssvio_dmatch* d_cand = nullptr; size_t candCap = 0; int* d_keep = nullptr; size_t keepCap = 0;
    // MatchByDBoW: per-feature transform results (query rows first, then train rows) and the grouped-match scratch
    int32_t* d_bowWord = nullptr; size_t bowWordCap = 0; int32_t* d_bowNode = nullptr; size_t bowNodeCap = 0;
    double* d_bowWeight = nullptr; size_t bowWeightCap = 0; int32_t* d_gnode = nullptr; size_t gnodeCap = 0;
    int32_t* d_keepNode = nullptr; size_t keepNodeCap = 0;
    int32_t* d_sortIdx = nullptr; size_t sortIdxCap = 0; int32_t* d_sortNode = nullptr; size_t sortNodeCap = 0;
    int lastBowQ = 0, lastBowT = 0;
    // behind the matcher (f2, f3)
    uint8_t* d_frameIn = nullptr; size_t frameInCap = 0;      // uploaded matches / keypoints / depth of the host entry points
    uint8_t* d_frameOut = nullptr; size_t frameOutCap = 0;    // assembled lists
    uint32_t* d_table = nullptr; size_t tableCap = 0;
    std::vector<uint8_t> stageQ, stageT;              // packed copies of strided descriptor rows
    std::vector<int32_t> h_pairs;                     // what d_pairs holds (an unchanged pair list is not uploaded again)
    KernelProfiler prof; std::string profOnly;
    int lastPairs = 0, lastMaxRows = 0;
};

template <typename T> static int grow(T*& p, size_t& cap, size_t need, hipStream_t s)
{
    if (need <= cap) return SSVIO_OK;
    SSVIO_HIP(hipStreamSynchronize(s));
    dfree(p);
    need += need / 4;
    SSVIO_HIP(hipMalloc(&p, need * sizeof(T)));
    cap = need;
    return SSVIO_OK;
}

// descriptor rows host -> device.  Continuous rows (the usual cv::Mat) are one copy; strided rows are packed on the
// host first: a 2-D copy of 32-byte rows costs a transfer per row (18 ms for 1000 rows on this stack).
static int upload_rows(uint8_t* dst, const uint8_t* src, int n, int stride, std::vector<uint8_t>& staging, hipStream_t s)
{
    if (n <= 0) return SSVIO_OK;
    if (stride == 32) { SSVIO_HIP(hipMemcpyAsync(dst, src, (size_t)n * 32, hipMemcpyHostToDevice, s)); return SSVIO_OK; }
    staging.resize((size_t)n * 32);
    for (int i = 0; i < n; ++i) memcpy(staging.data() + (size_t)i * 32, src + (size_t)i * stride, 32);
    SSVIO_HIP(hipMemcpyAsync(dst, staging.data(), (size_t)n * 32, hipMemcpyHostToDevice, s));
    return SSVIO_OK;
}

static int mt_check(ssvio_matcher* m) { if (!m) { set_error("null matcher handle"); return SSVIO_ERR_INVALID; } return SSVIO_OK; }

// choose how many pieces the train set is cut into so that the launch has >= ~2048 workgroups (two per SIMD)
static void pick_chunks(int maxQ, int maxT, int npairs, int& nChunks, int& chunkRows)
{
    static const int forcedRows = [] { const char* e = getenv("SSVIO_MATCH_CHUNKROWS"); return e ? atoi(e) : 0; }();   // tuning knob
    const int R = knn2_queries_per_lane(maxQ);
    const long qBlocks = (long)((maxQ + 256 * R - 1) / (256 * R)) * npairs;
    long want = (2048 + qBlocks - 1) / qBlocks;
    const long maxChunks = std::max(1, (maxT + 127) / 128);
    want = std::min<long>(std::min<long>(want, maxChunks), 64);
    nChunks = (int)std::max<long>(want, 1);
    chunkRows = ((maxT + nChunks - 1) / nChunks + 127) / 128 * 128;
    if (forcedRows >= 128) chunkRows = forcedRows / 128 * 128;
    nChunks = std::max(1, (maxT + chunkRows - 1) / chunkRows);
}

// the pair list of the batched entry points, on the device (an unchanged list is not uploaded again)
static int upload_pairs(ssvio_matcher* m, const int32_t* pairs, int npairs)
{
    if (m->h_pairs.size() == (size_t)npairs * 2 && memcmp(m->h_pairs.data(), pairs, (size_t)npairs * 2 * sizeof(int32_t)) == 0) return SSVIO_OK;
    int rc = grow(m->d_pairs, m->pairsCap, (size_t)npairs * 2, m->stream); if (rc) return rc;
    SSVIO_HIP(hipStreamSynchronize(m->stream));    // earlier launches may still read the old list
    m->h_pairs.assign(pairs, pairs + (size_t)npairs * 2);
    SSVIO_HIP(hipMemcpyAsync(m->d_pairs, m->h_pairs.data(), (size_t)npairs * 2 * sizeof(int), hipMemcpyHostToDevice, m->stream));
    SSVIO_HIP(hipStreamSynchronize(m->stream));
    return SSVIO_OK;
}

// shared driver: descriptors on the device; pairs == nullptr means the single pair (0,0)
static int match_run(ssvio_matcher* m, const uint8_t* d_qdesc, int qStrideRows, const int* d_qcounts,
                     const uint8_t* d_tdesc, int tStrideRows, const int* d_tcounts,
                     const int* d_pairs, int npairs, int maxQ, int maxT, int mode)
{
    hipStream_t s = m->stream;
    const int maxRows = std::max(maxQ, maxT);
    int nChunks, chunkRows;
    pick_chunks(maxQ, maxT, npairs, nChunks, chunkRows);
    int rc = grow(m->d_partial, m->partialCap, (size_t)npairs * nChunks * maxRows, s); if (rc) return rc;
    rc = grow(m->d_nout, m->noutCap, (size_t)npairs, s); if (rc) return rc;
    if (mode == 3) { rc = grow(m->d_raw, m->rawCap, (size_t)npairs * 3 * maxRows, s); if (rc) return rc; }
    else { rc = grow(m->d_out, m->outCap, (size_t)npairs * maxRows, s); if (rc) return rc; }
    MatchArgs a{};
    a.qdesc = (const uint4*)d_qdesc; a.tdesc = (const uint4*)d_tdesc;
    a.qSetStrideRows = qStrideRows; a.tSetStrideRows = tStrideRows;
    a.qcounts = d_qcounts; a.tcounts = d_tcounts; a.pairs = d_pairs; a.swap = 0;
    a.nChunks = nChunks; a.chunkRows = chunkRows; a.partial = m->d_partial; a.maxRows = maxRows;
    {
        ProfScope p(m->prof, "k_hamming_knn2", s);
        launch_knn2(a, maxQ, npairs, s);
    }
    FinalizeArgs f{};
    f.partial = m->d_partial; f.nChunks = nChunks; f.maxRows = maxRows;
    f.qcounts = d_qcounts; f.tcounts = d_tcounts; f.pairs = d_pairs;
    f.mode = mode; f.threshold = m->threshold; f.nnRatio = m->nnRatio;
    f.out = m->d_out; f.nout = m->d_nout; f.raw = m->d_raw;
    if (mode == SSVIO_MODE_CROSSCHECK) {
        int nChunksR, chunkRowsR;
        pick_chunks(maxT, maxQ, npairs, nChunksR, chunkRowsR);
        rc = grow(m->d_partialRev, m->partialRevCap, (size_t)npairs * nChunksR * maxRows, s); if (rc) return rc;
        MatchArgs r = a;
        r.qdesc = (const uint4*)d_tdesc; r.tdesc = (const uint4*)d_qdesc;
        r.qSetStrideRows = tStrideRows; r.tSetStrideRows = qStrideRows;
        r.qcounts = d_tcounts; r.tcounts = d_qcounts; r.swap = 1;
        r.nChunks = nChunksR; r.chunkRows = chunkRowsR; r.partial = m->d_partialRev;
        {
            ProfScope p(m->prof, "k_hamming_knn2", s);
            launch_knn2(r, maxT, npairs, s);
        }
        f.partialRev = m->d_partialRev; f.nChunksRev = nChunksR;
    }
    {
        ProfScope p(m->prof, "k_match_finalize", s);
        launch_finalize(f, npairs, maxQ, maxT, s);
    }
    m->lastPairs = npairs; m->lastMaxRows = maxRows;
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
    size_t qcap = m->qRows * 32, tcap = m->tRows * 32;
    int rc = grow(m->d_q, qcap, (size_t)nq * 32, s); if (rc) return rc; m->qRows = qcap / 32;
    rc = grow(m->d_t, tcap, (size_t)nt * 32, s); if (rc) return rc; m->tRows = tcap / 32;
    rc = upload_rows(m->d_q, q, nq, qstride, m->stageQ, s); if (rc) return rc;
    rc = upload_rows(m->d_t, t, nt, tstride, m->stageT, s); if (rc) return rc;
    const int counts[2] = {nq, nt};
    SSVIO_HIP(hipMemcpyAsync(m->d_counts2, counts, sizeof counts, hipMemcpyHostToDevice, s));
    SSVIO_HIP(hipStreamSynchronize(s));                                // `counts` is a stack array
    rc = match_run(m, m->d_q, 0, m->d_counts2, m->d_t, 0, m->d_counts2 + 1, nullptr, 1, nq, nt, mode);
    if (rc) return rc;
    const int maxRows = std::max(nq, nt);
    if (mode == 3) {
        std::vector<int32_t> raw((size_t)3 * maxRows);
        SSVIO_HIP(hipMemcpyAsync(raw.data(), m->d_raw, raw.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        SSVIO_HIP(hipStreamSynchronize(s));
        for (int i = 0; i < nq; ++i) { if (idx) idx[i] = raw[i]; if (d1) d1[i] = raw[maxRows + i]; if (d2) d2[i] = raw[2 * (size_t)maxRows + i]; }
    } else {
        int cnt = 0;
        SSVIO_HIP(hipMemcpyAsync(&cnt, m->d_nout, sizeof(int), hipMemcpyDeviceToHost, s));
        SSVIO_HIP(hipStreamSynchronize(s));
        if (cnt > 0) SSVIO_HIP(hipMemcpy(out, m->d_out, (size_t)cnt * sizeof(ssvio_dmatch), hipMemcpyDeviceToHost));
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
    if (hipStreamCreateWithFlags(&m->own, hipStreamNonBlocking) != hipSuccess || hipMalloc(&m->d_counts2, 2 * sizeof(int)) != hipSuccess) {
        set_error("cannot create HIP stream / allocate"); delete m; return SSVIO_ERR_HIP;
    }
    m->stream = m->own;
    *out = m;
    return SSVIO_OK;
}

void ssvio_matcher_destroy(ssvio_matcher* m)
{
    if (!m) return;
    (void)hipSetDevice(m->device);
    (void)hipStreamSynchronize(m->stream);
    m->prof.resolve(); m->prof.destroy();
    dfree(m->d_q); dfree(m->d_t); dfree(m->d_counts2); dfree(m->d_partial); dfree(m->d_partialRev);
    dfree(m->d_out); dfree(m->d_raw); dfree(m->d_nout); dfree(m->d_pairs);
    dfree(m->d_p3d); dfree(m->d_p2d); dfree(m->d_cand); dfree(m->d_keep);
    dfree(m->d_bowWord); dfree(m->d_bowNode); dfree(m->d_bowWeight); dfree(m->d_gnode); dfree(m->d_keepNode); dfree(m->d_sortIdx); dfree(m->d_sortNode);
    dfree(m->d_frameIn); dfree(m->d_frameOut); dfree(m->d_table);
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
    size_t qcap = m->qRows * 32, tcap = m->tRows * 32;
    int rc = grow(m->d_q, qcap, (size_t)nq * 32, s); if (rc) return rc; m->qRows = qcap / 32;
    rc = grow(m->d_t, tcap, (size_t)std::max(nt, 1) * 32, s); if (rc) return rc; m->tRows = tcap / 32;
    rc = grow(m->d_p3d, m->p3dCap, (size_t)nq * 3, s); if (rc) return rc;
    rc = grow(m->d_p2d, m->p2dCap, (size_t)std::max(nt, 1) * 2, s); if (rc) return rc;
    rc = grow(m->d_cand, m->candCap, (size_t)nq, s); if (rc) return rc;
    rc = grow(m->d_keep, m->keepCap, (size_t)nq, s); if (rc) return rc;
    rc = grow(m->d_out, m->outCap, (size_t)nq, s); if (rc) return rc;
    rc = grow(m->d_nout, m->noutCap, (size_t)1, s); if (rc) return rc;
    rc = upload_rows(m->d_q, q, nq, qstride, m->stageQ, s); if (rc) return rc;
    SSVIO_HIP(hipMemcpyAsync(m->d_p3d, p3d, (size_t)nq * 3 * sizeof(double), hipMemcpyHostToDevice, s));
    if (nt > 0) {
        rc = upload_rows(m->d_t, t, nt, tstride, m->stageT, s); if (rc) return rc;
        SSVIO_HIP(hipMemcpyAsync(m->d_p2d, p2d, (size_t)nt * 2 * sizeof(double), hipMemcpyHostToDevice, s));
    }
    ProjectArgs a{};
    a.p3d = m->d_p3d; a.qdesc = (const uint4*)m->d_q; a.nq = nq; a.p2d = m->d_p2d; a.tdesc = (const uint4*)m->d_t; a.nt = nt;
    a.radian = radian;
    for (int i = 0; i < 9; ++i) { a.K[i] = K[i]; a.R[i] = R[i]; }
    for (int i = 0; i < 3; ++i) a.t[i] = tr[i];
    a.threshold = m->threshold; a.nnRatio = m->nnRatio;
    a.cand = m->d_cand; a.keep = m->d_keep; a.out = m->d_out; a.nout = m->d_nout;
    {
        ProfScope p(m->prof, "k_match_project", s);
        launch_project(a, s);
    }
    int cnt = 0;
    SSVIO_HIP(hipMemcpyAsync(&cnt, m->d_nout, sizeof(int), hipMemcpyDeviceToHost, s));
    SSVIO_HIP(hipStreamSynchronize(s));
    if (cnt > 0) SSVIO_HIP(hipMemcpy(out, m->d_out, (size_t)cnt * sizeof(ssvio_dmatch), hipMemcpyDeviceToHost));
    *n = cnt;
    m->prof.resolve();
    return SSVIO_OK;
}

int ssvio_match_batch_dev(ssvio_matcher* m, const void* d_desc, int set_stride_rows, int max_rows, const int32_t* d_counts,
                          const int32_t* pairs, int npairs, int mode, const void** d_matches, const int32_t** d_nmatches)
{
    if (mt_check(m)) return SSVIO_ERR_INVALID;
    if (!d_desc || !d_counts || !pairs || npairs <= 0 || max_rows <= 0 || set_stride_rows < max_rows) { set_error("bad batch arguments"); return SSVIO_ERR_INVALID; }
    if (mode < 0 || mode > 2) { set_error("mode must be 0..2"); return SSVIO_ERR_INVALID; }
    if (max_rows > (1 << 23)) { set_error("more than 2^23 rows per set"); return SSVIO_ERR_INVALID; }
    SSVIO_HIP(hipSetDevice(m->device));
    int rc = SSVIO_OK;
    rc = upload_pairs(m, pairs, npairs); if (rc) return rc;
    rc = match_run(m, (const uint8_t*)d_desc, set_stride_rows, d_counts, (const uint8_t*)d_desc, set_stride_rows, d_counts,
                   m->d_pairs, npairs, max_rows, max_rows, mode);
    if (rc) return rc;
    if (d_matches) *d_matches = m->d_out;
    if (d_nmatches) *d_nmatches = m->d_nout;
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
                    int maxRows, int levelsup, size_t rowBase)
{
    BowArgs a{};
    a.v = vocab_dev(v); a.nidLevel = v->L - levelsup;
    a.desc = (const uint4*)d_desc; a.setStrideRows = setStrideRows; a.counts = d_counts; a.nsets = nsets; a.maxRows = maxRows;
    a.word = m->d_bowWord + rowBase; a.node = m->d_bowNode + rowBase; a.weight = m->d_bowWeight + rowBase; a.gnode = m->d_gnode + rowBase;
    ProfScope p(m->prof, "k_bow_transform", m->stream);
    launch_bow_transform(a, maxRows, m->stream);
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
    size_t qcap = m->qRows * 32;
    rc = grow(m->d_q, qcap, (size_t)n * 32, s); if (rc) return rc; m->qRows = qcap / 32;
    rc = bow_grow(m, (size_t)n); if (rc) return rc;
    rc = upload_rows(m->d_q, desc, n, stride, m->stageQ, s); if (rc) return rc;
    const int counts[2] = {n, 0};
    SSVIO_HIP(hipMemcpyAsync(m->d_counts2, counts, sizeof counts, hipMemcpyHostToDevice, s));
    SSVIO_HIP(hipStreamSynchronize(s));
    bow_run(m, v, m->d_q, 0, m->d_counts2, 1, n, levelsup, 0);
    if (word) SSVIO_HIP(hipMemcpyAsync(word, m->d_bowWord, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    if (node) SSVIO_HIP(hipMemcpyAsync(node, m->d_bowNode, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    if (weight) SSVIO_HIP(hipMemcpyAsync(weight, m->d_bowWeight, (size_t)n * 8, hipMemcpyDeviceToHost, s));
    SSVIO_HIP(hipStreamSynchronize(s));
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
    size_t qcap = m->qRows * 32, tcap = m->tRows * 32;
    rc = grow(m->d_q, qcap, (size_t)nq * 32, s); if (rc) return rc; m->qRows = qcap / 32;
    rc = grow(m->d_t, tcap, (size_t)nt * 32, s); if (rc) return rc; m->tRows = tcap / 32;
    rc = bow_grow(m, (size_t)nq + nt); if (rc) return rc;
    rc = grow(m->d_cand, m->candCap, (size_t)nq, s); if (rc) return rc;
    rc = grow(m->d_keepNode, m->keepNodeCap, (size_t)nq, s); if (rc) return rc;
    rc = grow(m->d_out, m->outCap, (size_t)nq, s); if (rc) return rc;
    rc = grow(m->d_nout, m->noutCap, (size_t)1, s); if (rc) return rc;
    rc = upload_rows(m->d_q, q, nq, qstride, m->stageQ, s); if (rc) return rc;
    rc = upload_rows(m->d_t, t, nt, tstride, m->stageT, s); if (rc) return rc;
    const int counts[2] = {nq, nt};
    SSVIO_HIP(hipMemcpyAsync(m->d_counts2, counts, sizeof counts, hipMemcpyHostToDevice, s));
    SSVIO_HIP(hipStreamSynchronize(s));
    bow_run(m, v, m->d_q, 0, m->d_counts2, 1, nq, levelsup, 0);               // ComputeBoWVector(desp1), :68
    bow_run(m, v, m->d_t, 0, m->d_counts2 + 1, 1, nt, levelsup, (size_t)nq);  // ComputeBoWVector(desp2), :69
    if (std::max(nq, nt) <= kBowSortMax && !getenv("SSVIO_DBOW_UNSORTED")) {
        bow_sort(m, 0, m->d_counts2, 1, nq, 0);
        bow_sort(m, 0, m->d_counts2 + 1, 1, nt, (size_t)nq);
        SortedMatchArgs g{};
        g.qdesc = (const uint4*)m->d_q; g.tdesc = (const uint4*)m->d_t;
        g.qSortedIdx = m->d_sortIdx; g.qSortedNode = m->d_sortNode; g.tSortedIdx = m->d_sortIdx + nq; g.tSortedNode = m->d_sortNode + nq;
        g.qSetStrideRows = 0; g.tSetStrideRows = 0; g.qcounts = m->d_counts2; g.tcounts = m->d_counts2 + 1; g.pairs = nullptr;
        g.threshold = m->threshold; g.nnRatio = m->nnRatio; g.maxRows = nq; g.maxT = nt;
        g.cand = m->d_cand; g.keep = m->d_keepNode; g.out = m->d_out; g.nout = m->d_nout;
        ProfScope p(m->prof, "k_match_sorted", s);
        launch_match_sorted(g, nq, nt, 1, s);
    } else {                                                           // sets beyond the LDS sort: wave per query over all train rows
        GroupedArgs g{};
        g.qdesc = (const uint4*)m->d_q; g.tdesc = (const uint4*)m->d_t; g.qgnode = m->d_gnode; g.tgnode = m->d_gnode + nq;
        g.qSetStrideRows = 0; g.tSetStrideRows = 0; g.qcounts = m->d_counts2; g.tcounts = m->d_counts2 + 1; g.pairs = nullptr;
        g.threshold = m->threshold; g.nnRatio = m->nnRatio; g.maxRows = nq; g.maxT = nt;
        g.cand = m->d_cand; g.keepNode = m->d_keepNode; g.out = m->d_out; g.nout = m->d_nout;
        ProfScope p(m->prof, "k_match_grouped", s);
        launch_match_grouped(g, nq, 1, s);
    }
    int cnt = 0;
    SSVIO_HIP(hipMemcpyAsync(&cnt, m->d_nout, sizeof(int), hipMemcpyDeviceToHost, s));
    SSVIO_HIP(hipStreamSynchronize(s));
    if (cnt > 0) SSVIO_HIP(hipMemcpy(out, m->d_out, (size_t)cnt * sizeof(ssvio_dmatch), hipMemcpyDeviceToHost));
    *n = cnt;
    m->prof.resolve();
    m->lastBowQ = nq; m->lastBowT = nt;
    return SSVIO_OK;
}

int ssvio_bow_last_transform(ssvio_matcher* m, int side, int32_t* word, int32_t* node, double* weight, int cap, int* n)
{
    if (mt_check(m)) return SSVIO_ERR_INVALID;
    if (side != 0 && side != 1) { set_error("side must be 0 (desp1) or 1 (desp2)"); return SSVIO_ERR_INVALID; }
    const int rows = side ? m->lastBowT : m->lastBowQ;
    const size_t base = side ? (size_t)m->lastBowQ : 0;
    if (n) *n = rows;
    if (rows == 0) return SSVIO_OK;
    if (cap < rows) { set_error("capacity %d < %d rows", cap, rows); return SSVIO_ERR_CAPACITY; }
    SSVIO_HIP(hipSetDevice(m->device));
    hipStream_t s = m->stream;
    if (word) SSVIO_HIP(hipMemcpyAsync(word, m->d_bowWord + base, (size_t)rows * 4, hipMemcpyDeviceToHost, s));
    if (node) SSVIO_HIP(hipMemcpyAsync(node, m->d_bowNode + base, (size_t)rows * 4, hipMemcpyDeviceToHost, s));
    if (weight) SSVIO_HIP(hipMemcpyAsync(weight, m->d_bowWeight + base, (size_t)rows * 8, hipMemcpyDeviceToHost, s));
    SSVIO_HIP(hipStreamSynchronize(s));
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
    } else {
        GroupedArgs g{};
        g.qdesc = (const uint4*)d_desc; g.tdesc = (const uint4*)d_desc; g.qgnode = m->d_gnode; g.tgnode = m->d_gnode;
        g.qSetStrideRows = set_stride_rows; g.tSetStrideRows = set_stride_rows; g.qcounts = d_counts; g.tcounts = d_counts; g.pairs = m->d_pairs;
        g.threshold = m->threshold; g.nnRatio = m->nnRatio; g.maxRows = max_rows; g.maxT = max_rows;
        g.cand = m->d_cand; g.keepNode = m->d_keepNode; g.out = m->d_out; g.nout = m->d_nout;
        ProfScope p(m->prof, "k_match_grouped", s);
        launch_match_grouped(g, max_rows, npairs, s);
    }
    m->lastPairs = npairs; m->lastMaxRows = max_rows;
    m->lastBowQ = 0; m->lastBowT = 0;
    if (d_matches) *d_matches = m->d_out;
    if (d_nmatches) *d_nmatches = m->d_nout;
    return SSVIO_OK;
}

}  // extern "C"

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

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// carve the assembled-list block of the handle into its seven arrays + counts
static int assemble_outputs(ssvio_matcher* m, int npairs, int maxRows, AssembleArgs& a)
{
    const size_t rows = (size_t)npairs * maxRows;
    const size_t s3 = align256(rows * 12), s2 = align256(rows * 8), sm = align256(rows * 16), sc = align256((size_t)npairs * 8);
    int rc = grow(m->d_frameOut, m->frameOutCap, 4 * s3 + s2 + 2 * sm + sc, m->stream); if (rc) return rc;
    uint8_t* p = m->d_frameOut;
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
    const int kpStride = std::max(std::max(nlast, ncur), 1);
    const size_t depthPx = (size_t)h * stride_px;
    const size_t mb = align256((size_t)nm * 16), kb = align256((size_t)2 * kpStride * sizeof(ssvio_keypoint)), db = align256(2 * depthPx * 2);
    rc = grow(m->d_frameIn, m->frameInCap, mb + kb + db + 256, s); if (rc) return rc;
    uint8_t* d_m = m->d_frameIn; uint8_t* d_k = d_m + mb; uint8_t* d_d = d_k + kb; int* d_c = (int*)(d_d + db);   // {nm, ncur, nlast}
    SSVIO_HIP(hipMemcpyAsync(d_m, matches, (size_t)nm * 16, hipMemcpyHostToDevice, s));
    if (ncur) SSVIO_HIP(hipMemcpyAsync(d_k, cur_kps, (size_t)ncur * sizeof(ssvio_keypoint), hipMemcpyHostToDevice, s));       // set 0 = current
    if (nlast) SSVIO_HIP(hipMemcpyAsync(d_k + (size_t)kpStride * sizeof(ssvio_keypoint), last_kps, (size_t)nlast * sizeof(ssvio_keypoint), hipMemcpyHostToDevice, s));
    SSVIO_HIP(hipMemcpyAsync(d_d, cur_depth, depthPx * 2, hipMemcpyHostToDevice, s));
    SSVIO_HIP(hipMemcpyAsync(d_d + depthPx * 2, last_depth, depthPx * 2, hipMemcpyHostToDevice, s));
    const int cnts[3] = {nm, ncur, nlast};
    SSVIO_HIP(hipMemcpyAsync(d_c, cnts, sizeof cnts, hipMemcpyHostToDevice, s));
    SSVIO_HIP(hipStreamSynchronize(s));                             // `cnts` is a stack array
    AssembleArgs a{};
    a.matches = (const ssvio_dmatch*)d_m; a.nmatches = d_c; a.maxRows = nm; a.pairs = nullptr;
    a.kps = (const ssvio_keypoint*)d_k; a.kpStride = kpStride; a.kpCounts = d_c + 1;
    a.depth = (const uint16_t*)d_d; a.depthFrameStride = depthPx; a.cam = c;
    rc = assemble_outputs(m, 1, nm, a); if (rc) return rc;
    {
        ProfScope p(m->prof, "k_assemble", s);
        launch_assemble(a, 1, s);
    }
    int got[2] = {0, 0};
    SSVIO_HIP(hipMemcpyAsync(got, a.counts, sizeof got, hipMemcpyDeviceToHost, s));
    SSVIO_HIP(hipStreamSynchronize(s));
    const size_t np = (size_t)got[0], nf = (size_t)got[1];
    if (np) {
        SSVIO_HIP(hipMemcpyAsync(last3d, a.last3d, np * 12, hipMemcpyDeviceToHost, s));
        SSVIO_HIP(hipMemcpyAsync(cur2d, a.cur2d, np * 8, hipMemcpyDeviceToHost, s));
        SSVIO_HIP(hipMemcpyAsync(cur3d, a.cur3d, np * 12, hipMemcpyDeviceToHost, s));
        SSVIO_HIP(hipMemcpyAsync(pre_good, a.preGood, np * 16, hipMemcpyDeviceToHost, s));
    }
    if (nf) {
        SSVIO_HIP(hipMemcpyAsync(edge_last3d, a.edgeLast3d, nf * 12, hipMemcpyDeviceToHost, s));
        SSVIO_HIP(hipMemcpyAsync(edge_cur3d, a.edgeCur3d, nf * 12, hipMemcpyDeviceToHost, s));
        SSVIO_HIP(hipMemcpyAsync(final_good, a.finalGood, nf * 16, hipMemcpyDeviceToHost, s));
    }
    SSVIO_HIP(hipStreamSynchronize(s));
    *n_pre = got[0]; *n_final = got[1];
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
    launch_first_index(d_v, n, useTrain, m->d_table, tableSize, s);
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
    int rc = grow(m->d_frameIn, m->frameInCap, lb + cb, s); if (rc) return rc;
    rc = grow(m->d_out, m->outCap, (size_t)nlast, s); if (rc) return rc;
    rc = grow(m->d_nout, m->noutCap, (size_t)1, s); if (rc) return rc;
    SSVIO_HIP(hipMemcpyAsync(m->d_frameIn, last, (size_t)nlast * 16, hipMemcpyHostToDevice, s));
    SSVIO_HIP(hipMemcpyAsync(m->d_frameIn + lb, cur, (size_t)ncur * 16, hipMemcpyHostToDevice, s));
    rc = first_index_table(m, (const ssvio_dmatch*)(m->d_frameIn + lb), ncur, 1, maxKey + 1); if (rc) return rc;
    {
        ProfScope p(m->prof, "k_track", s);
        launch_track((const ssvio_dmatch*)m->d_frameIn, nlast, (const ssvio_dmatch*)(m->d_frameIn + lb), m->d_table, maxKey + 1, m->d_out, m->d_nout, s);
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
    }
    SSVIO_HIP(hipMemcpyAsync(out, m->d_frameOut, ob, hipMemcpyDeviceToHost, s));
    SSVIO_HIP(hipStreamSynchronize(s));
    m->prof.resolve();
    return SSVIO_OK;
}
