// geometry.cpp -- host-side tables of the extractor: scale pyramid, per-level
// feature quotas, FAST cell grid, bilinear-resize coefficient tables, and the
// quadtree keypoint distribution.  Product code (no dependency on oracle/).
#include "common.h"
#include <cstdlib>
#include <cmath>
#include <cstdarg>
#include <algorithm>

namespace ssvio {

static thread_local std::string g_err;
void set_error(const char* fmt, ...)
{
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf;
}
const char* last_error_cstr() { return g_err.c_str(); }
int hip_fail(hipError_t e, const char* what, const char* file, int line)
{
    set_error("HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
    (void)hipGetLastError();                  // reported once: the launch checks read the runtime's last-error slot
    return SSVIO_ERR_HIP;
}

static inline int round_half_even(double v) { return (int)std::lrint(v); }   // cvRound

// ORBextractor::ORBextractor, src/Feature.cpp:379-417.  The member scaleFactor is a
// double initialised from the float argument (include/Feature.h:89), so the per-level
// factor is float(double(prev) * double(float(sf))).
int build_tables(const ssvio_orb_params& p, ExtractorTables& t)
{
    if (p.nlevels < 1 || p.nlevels > kMaxLevels) { set_error("nlevels %d outside 1..%d", p.nlevels, kMaxLevels); return SSVIO_ERR_INVALID; }
    if (!(p.scale_factor > 1.0f)) { set_error("scale_factor must be > 1"); return SSVIO_ERR_INVALID; }
    if (p.nfeatures < 1) { set_error("nfeatures must be >= 1"); return SSVIO_ERR_INVALID; }
    t.nfeatures = p.nfeatures; t.nlevels = p.nlevels; t.iniTh = p.ini_th_fast; t.minTh = p.min_th_fast;
    const double sf = (double)p.scale_factor;
    t.scale[0] = 1.0f; t.sigma2[0] = 1.0f;
    for (int i = 1; i < p.nlevels; ++i) {
        t.scale[i] = (float)((double)t.scale[i - 1] * sf);
        t.sigma2[i] = t.scale[i] * t.scale[i];
    }
    for (int i = 0; i < p.nlevels; ++i) { t.invScale[i] = 1.0f / t.scale[i]; t.invSigma2[i] = 1.0f / t.sigma2[i]; }
    const float factor = (float)(1.0f / sf);
    float want = p.nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)p.nlevels));
    int sum = 0;
    for (int l = 0; l < p.nlevels - 1; ++l) {
        t.quota[l] = round_half_even(want);
        sum += t.quota[l];
        want *= factor;
    }
    t.quota[p.nlevels - 1] = std::max(p.nfeatures - sum, 0);
    return SSVIO_OK;
}

// ComputePyramid sizes (src/Feature.cpp:804-805) and the FAST grid (:835-849).
int build_geometry(const ExtractorTables& t, int width, int height, PyramidGeom& g)
{
    g = PyramidGeom();
    g.nlevels = t.nlevels; g.width = width; g.height = height;
    size_t off = 0, slot = 0; int cell = 0, tiles = 0, ftiles = 0, ftilesT = 0;
    for (int l = 0; l < t.nlevels; ++l) {
        LevelGeom& L = g.lv[l];
        L.w = round_half_even((float)width * t.invScale[l]);
        L.h = round_half_even((float)height * t.invScale[l]);
        if (L.w > 4096 + 2 * kMinBorder || L.h > 4096 + 2 * kMinBorder) {
            set_error("level %d is %dx%d: candidate packing supports levels up to 4128 px", l, L.w, L.h);
            return SSVIO_ERR_GEOMETRY;
        }
        L.maxBX = L.w - kEdge + 3; L.maxBY = L.h - kEdge + 3;
        const float fw = (float)(L.maxBX - kMinBorder), fh = (float)(L.maxBY - kMinBorder);
        L.nCols = (int)(fw / 30.f); L.nRows = (int)(fh / 30.f);
        if (L.nCols < 1 || L.nRows < 1) {
            set_error("pyramid level %d is %dx%d: smaller than one 30-px FAST cell plus borders "
                      "(the reference divides by zero here, src/Feature.cpp:846-849)", l, L.w, L.h);
            return SSVIO_ERR_GEOMETRY;
        }
        L.wCell = (int)std::ceil(fw / L.nCols); L.hCell = (int)std::ceil(fh / L.nRows);
        const int nIni = (int)std::round(fw / fh);                    // src/Feature.cpp:574
        if (nIni > kMaxRoots) { set_error("level %d aspect ratio %g needs %d quadtree roots (max %d supported)", l, fw / fh, nIni, kMaxRoots); return SSVIO_ERR_GEOMETRY; }
        if (nIni < 1) { set_error("level %d aspect ratio %g rounds to zero quadtree roots (src/Feature.cpp:574)", l, fw / fh); return SSVIO_ERR_GEOMETRY; }
        if (L.wCell > kMaxCellDim || L.hCell > kMaxCellDim) { set_error("internal: cell %dx%d exceeds %d", L.wCell, L.hCell, kMaxCellDim); return SSVIO_ERR_GEOMETRY; }
        L.pitch = ((L.w + 63) / 64) * 64 + 128;         // 64-B aligned rows, 64 B of slack on both sides (pad columns, vector reads)
        L.offset = off + 64;                            // x = 0 of row 0
        off += (size_t)L.pitch * (L.h + 1);
        off = (off + 255) & ~(size_t)255;
        L.cellBase = cell; cell += L.nCols * L.nRows;
        L.slotCap = ((L.wCell + 1) / 2) * ((L.hCell + 1) / 2);        // NMS leaves no two 8-adjacent maxima
        L.slotBase = slot; slot += (size_t)L.slotCap * L.nCols * L.nRows;
        L.selBase = g.maxQuotaSum;                                      // running sum of level_sel_cap
        L.nIni = nIni; L.hX = fw / nIni;                                 // :576
        for (int i = 0; i <= nIni; ++i) L.rootX[i] = (int)(L.hX * (float)i);   // :585-586
        L.quota = t.quota[l]; L.scale = t.scale[l]; L.kpSize = (int)(kHalfPatch * t.scale[l]);   // src/Feature.cpp:900
        // k_blur strips: 120 output px wide (32 lanes x 4 px, first/last lane are halo), 7*iters-6 rows tall
        L.blurStripsX = (L.w + 119) / 120;
        {
            static const int target = [] { const char* e = getenv("SSVIO_BLUR_ROWS"); const int v = e ? atoi(e) : 0; return v >= 8 ? v : 29; }();   // rows per strip (SSVIO_BLUR_ROWS: tuning knob; 57 -> 29 rows: 79 -> 62 us per 64 VGA frames)
            const int nsY = (L.h + target - 1) / target, rows = (L.h + nsY - 1) / nsY;
            L.blurIters = (rows + 6 + 6) / 7;
            L.blurStripsY = (L.h + (7 * L.blurIters - 6) - 1) / (7 * L.blurIters - 6);
        }
        L.blurStripBase = tiles; tiles += L.blurStripsX * L.blurStripsY;
        // the blurred level in 32-px-wide column panels (a 128-byte line = 4 rows of a panel): the 37 x 37 window k_describe samples touches
        // ~23 lines instead of ~48 in row-major order.  Columns up to w + 2 and rows up to h are addressed (alignment, three-row passes).
        L.blurPanelStride = ((L.h + 2) * 32 + 127) & ~127;
        if ((size_t)((L.w + 3 + 31) / 32) * (size_t)L.blurPanelStride > (size_t)L.pitch * (L.h + 1)) {
            set_error("level %d is %dx%d: the blurred level's panel layout does not fit its slot", l, L.w, L.h);
            return SSVIO_ERR_GEOMETRY;
        }
        // k_fast tiles cover the tested region only: columns from kFastX0 = 16 (aligned) to w - 20, rows 19 .. h - 20 split evenly
        {
            const int cols = L.w - kEdge - kFastX0, rows = L.h - 2 * kEdge;          // both >= 24: a level holds at least one 30-px cell
            L.fastTilesX = (cols + 63) / 64;
            L.fastTilesY = (rows + 31) / 32; L.fastTileH = (rows + L.fastTilesY - 1) / L.fastTilesY;
            L.fastTileBase = ftiles; ftiles += L.fastTilesX * L.fastTilesY;
            L.fastTilesYT = (rows + 63) / 64; L.fastTileHT = (rows + L.fastTilesYT - 1) / L.fastTilesYT;
            L.fastTileBaseT = ftilesT; ftilesT += L.fastTilesX * L.fastTilesYT;
        }
        g.maxQuotaSum += level_sel_cap(L.quota);
    }
    g.frameBytes = off + 256;
    g.cellsPerFrame = cell; g.slotsPerFrame = slot; g.blurStripsPerFrame = tiles; g.fastTilesPerFrame = ftiles; g.fastTilesPerFrameT = ftilesT;
    return SSVIO_OK;
}

// cv::resize(INTER_LINEAR) coefficient tables, 8UC1: 11-bit weights,
// source index clamped with the weight moved to the valid tap.
static inline short sat_short(float v) { int i = round_half_even(v); return (short)std::min(32767, std::max(-32768, i)); }
void build_resize_tabs(int sw, int sh, int dw, int dh, ResizeTabs& t)
{
    const double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
    t.xs.assign((size_t)dw * 4, 0); t.ys.assign((size_t)dh * 4, 0);
    t.area2 = scale_x == 2.0 && scale_y == 2.0;     // cv::resize turns INTER_LINEAR into INTER_AREA here
    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)std::floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        t.xs[dx * 4 + 0] = (int16_t)sx;
        t.xs[dx * 4 + 1] = (int16_t)std::min(sx + 1, sw - 1);
        t.xs[dx * 4 + 2] = sat_short((1.f - fx) * 2048);
        t.xs[dx * 4 + 3] = sat_short(fx * 2048);
    }
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)std::floor(fy);
        fy -= sy;
        t.ys[dy * 4 + 0] = (int16_t)std::min(std::max(sy, 0), sh - 1);
        t.ys[dy * 4 + 1] = (int16_t)std::min(std::max(sy + 1, 0), sh - 1);
        t.ys[dy * 4 + 2] = sat_short((1.f - fy) * 2048);
        t.ys[dy * 4 + 3] = sat_short(fy * 2048);
    }
    // quad table for k_resize_quad: the 8 source taps of 4 adjacent outputs lie in a window of <= 8 bytes
    // starting at sx(dx0) when the scale is <= ~2; then two v_perm selectors per row replace 8 byte loads
    const int nq = (dw + 2 * kPadX + 3) / 4;            // quad q covers dst x = 4q-4 .. 4q-1 (pad columns included)
    t.quads.assign((size_t)nq * 16, 0);
    t.quadOk = true;
    for (int q = 0; q < nq; ++q) {
        ResizeQuad e; memset(&e, 0, sizeof e);
        int dxs[4], s0 = 1 << 30;
        for (int i = 0; i < 4; ++i) {
            int p = 4 * q - kPadX + i;
            p = p < 0 ? -p : p; p = p >= dw ? 2 * dw - 2 - p : p;          // reflect-101 pad columns
            dxs[i] = std::min(std::max(p, 0), dw - 1);
            s0 = std::min(s0, (int)t.xs[dxs[i] * 4 + 0]);
        }
        e.bOff = (uint32_t)(s0 >> 2) | ((uint32_t)(s0 & 3) << 24);
        for (int i = 0; i < 4; ++i) {
            const int dx = dxs[i];
            const int p0 = t.xs[dx * 4 + 0] - s0, p1 = t.xs[dx * 4 + 1] - s0;
            if (p0 < 0 || p0 > 7 || p1 < 0 || p1 > 7) t.quadOk = false;
            e.sel0 |= (uint32_t)(p0 & 7) << (8 * i); e.sel1 |= (uint32_t)(p1 & 7) << (8 * i);
            e.a0[i] = t.xs[dx * 4 + 2]; e.a1[i] = t.xs[dx * 4 + 3];
        }
        memcpy(&t.quads[(size_t)q * 16], &e, sizeof e);
    }
    // output rows per wave: the strip's source rows must fit the kernel's 12-row register window
    t.rows = std::min(8, std::max(1, (int)(10.0 / scale_y)));
    for (bool ok = false; !ok && t.rows >= 1;) {
        ok = true;
        for (int y = 0; y < dh && ok; y += t.rows) {
            const int ye = std::min(y + t.rows, dh) - 1;
            if (t.ys[ye * 4 + 1] - t.ys[y * 4 + 0] + 1 > 12) ok = false;
        }
        if (!ok && --t.rows < 1) { t.rows = 1; t.quadOk = false; break; }
    }
}

// ---------------------------------------------------------------------------
// DistributeOctTree (src/Feature.cpp:570-798) in array form.
//
// Observations that remove the linked list and the per-node key vectors:
//  * every node is an axis-aligned rectangle [x0,x1) x [y0,y1) (DivideNode, :17-47);
//  * children are pushed at the FRONT and parents erased, so the list is always
//    "nodes in reverse creation order" (roots: root 0 first);
//  * DivideNode keeps keys in their parent's order, so vKeys is always in input
//    order and "first key with the maximum response" (:783-793) is the
//    max-response key with the lowest input index;
//  * sort() on pair<int, ExtractorNode*> (:716) breaks size ties by heap address,
//    which no restatement can reproduce.  Canonical tie-break here and in
//    the oracle: creation order (later-created node sorts higher).
// What remains per round is: count keys per child, pick the nodes to split (all
// expandable nodes, or the size-sorted prefix that reaches N), create children.
// The GPU version of this stage uses the same formulation.
struct QNode { int x0, y0, x1, y1; int n; long seq; };

int distribute_octtree(const float* xs, const float* ys, const float* resp, int n,
                       int minX, int maxX, int minY, int maxY, int N, std::vector<int>& out)
{
    out.clear();
    const int nIni = (int)std::round((float)(maxX - minX) / (maxY - minY));
    if (nIni < 1) { set_error("quadtree: aspect ratio rounds to zero roots"); return SSVIO_ERR_GEOMETRY; }
    const float hX = (float)(maxX - minX) / nIni;
    std::vector<QNode> list;             // front of the reference's std::list == index 0
    std::vector<int> knode(n);           // key -> index in `list`
    long seq = 0;
    for (int i = 0; i < nIni; ++i)
        list.push_back(QNode{(int)(hX * (float)i), 0, (int)(hX * (float)(i + 1)), maxY - minY, 0, -(long)i});
    for (int k = 0; k < n; ++k) {
        int b = (int)(xs[k] / hX);
        b = std::min(std::max(b, 0), nIni - 1);
        knode[k] = b; list[b].n++;
    }
    {   // erase empty roots (:611-614)
        std::vector<int> remap(list.size(), -1); std::vector<QNode> keep;
        for (size_t i = 0; i < list.size(); ++i) if (list[i].n > 0) { remap[i] = (int)keep.size(); keep.push_back(list[i]); }
        for (int k = 0; k < n; ++k) knode[k] = remap[knode[k]];
        list.swap(keep);
    }
    std::vector<int> cnt, order, childOf, quad(n);
    std::vector<char> split;
    bool careful = false, finish = false;
    while (!finish) {
        const int prevSize = (int)list.size();
        // children key counts for every expandable node
        cnt.assign(list.size() * 4, 0);
        std::vector<int> hx(list.size()), hy(list.size());
        for (size_t i = 0; i < list.size(); ++i) {
            hx[i] = list[i].x0 + (int)std::ceil((float)(list[i].x1 - list[i].x0) / 2);
            hy[i] = list[i].y0 + (int)std::ceil((float)(list[i].y1 - list[i].y0) / 2);
        }
        for (int k = 0; k < n; ++k) {
            const int p = knode[k];
            if (list[p].n < 2) continue;
            const int q = (xs[k] < (float)hx[p] ? 0 : 1) + (ys[k] < (float)hy[p] ? 0 : 2);   // n1,n2,n3,n4 = 0,1,2,3
            quad[k] = q; cnt[p * 4 + q]++;
        }
        // processing order
        order.clear();
        for (size_t i = 0; i < list.size(); ++i) if (list[i].n > 1) order.push_back((int)i);
        int nToExpand = 0;
        if (careful) {
            // largest first; ties: latest created first (iterating the ascending sort from the back)
            std::sort(order.begin(), order.end(), [&](int a, int b) {
                if (list[a].n != list[b].n) return list[a].n > list[b].n;
                return list[a].seq > list[b].seq;
            });
        }
        split.assign(list.size(), 0);
        int size = prevSize;
        std::vector<QNode> created;
        childOf.assign(list.size() * 4, -1);
        for (int p : order) {
            split[p] = 1;
            const QNode& P = list[p];
            const int xs4[4][2] = {{P.x0, hx[p]}, {hx[p], P.x1}, {P.x0, hx[p]}, {hx[p], P.x1}};
            const int ys4[4][2] = {{P.y0, hy[p]}, {P.y0, hy[p]}, {hy[p], P.y1}, {hy[p], P.y1}};
            for (int q = 0; q < 4; ++q) {
                const int c = cnt[p * 4 + q];
                if (c > 0) {
                    childOf[p * 4 + q] = (int)created.size();
                    created.push_back(QNode{xs4[q][0], ys4[q][0], xs4[q][1], ys4[q][1], c, ++seq});
                    ++size;
                    if (c > 1) ++nToExpand;
                }
            }
            --size;
            if (careful && size >= N) break;          // :762
        }
        // rebuild the list: reversed creations, then the survivors in their old order
        const int nc = (int)created.size();
        std::vector<QNode> next; next.reserve(nc + list.size());
        for (int i = nc - 1; i >= 0; --i) next.push_back(created[i]);
        std::vector<int> remap(list.size(), -1);
        for (size_t i = 0; i < list.size(); ++i) if (!split[i]) { remap[i] = (int)next.size(); next.push_back(list[i]); }
        for (int k = 0; k < n; ++k) {
            const int p = knode[k];
            knode[k] = split[p] ? (nc - 1 - childOf[p * 4 + quad[k]]) : remap[p];
        }
        list.swap(next);
        const int newSize = (int)list.size();
        if (newSize >= N || newSize == prevSize) finish = true;              // :703 / :769
        else if (!careful && newSize + nToExpand * 3 > N) careful = true;    // :707
    }
    // best key per node: max response, lowest input index (:778-796)
    std::vector<int> best(list.size(), -1);
    for (int k = 0; k < n; ++k) {
        int& b = best[knode[k]];
        if (b < 0 || resp[k] > resp[b]) b = k;
    }
    out.assign(best.begin(), best.end());
    return SSVIO_OK;
}

}  // namespace ssvio
