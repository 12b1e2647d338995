/* orb_oracle.c -- CPU restatement of SSVIO's ORB extractor (TEST INFRASTRUCTURE ONLY).
 *
 * PARITY UNPINNED (see ssvio_oracle.h): no reference fixture exists; pinned by
 * hand-derived known-answer tests only.
 *
 * Reference-authored steps follow /root/reference/src/Feature.cpp line by line
 * (cited per function).  OpenCV primitives follow SURVEY.md Appendix A, i.e. the
 * OpenCV 3.x generic C++ paths (integer fixed point, no SIMD/IPP variants).
 * Build with -ffp-contract=off: every float op below is a single IEEE operation.
 */
#include "ssvio_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define HALF_PATCH 15      /* src/Feature.cpp:13 */
#define EDGE_TH    19      /* src/Feature.cpp:14 */
#define MAXLV      32

static const int8_t k_pattern[1024] = {
#include "orb_pattern.inc"
};

/* ------------------------------------------------------------------ A0 */
int orc_cvround(double v) { return (int)lrint(v); }           /* round-half-even under the default mode */
static int cv_floor(double v) { int i = (int)v; return i - (v < i); }
static int cv_ceil(double v)  { int i = (int)v; return i + (v > i); }
static short sat_short(float v) { int i = orc_cvround(v); return (short)(i < -32768 ? -32768 : i > 32767 ? 32767 : i); }

/* ------------------------------------------------------------------ A1 */
/* cvtColor(.., CV_RGB2GRAY), 8U, OpenCV 3.x: 14-bit fixed point, weights on
 * memory-order bytes (0.299, 0.587, 0.114). src/Frame.cpp:7 */
void orc_rgb2gray(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride)
{
    for (int y = 0; y < h; ++y) {
        const uint8_t* s = src + (size_t)y * sstride;
        uint8_t* d = dst + (size_t)y * dstride;
        for (int x = 0; x < w; ++x, s += 3)
            d[x] = (uint8_t)((s[0] * 4899 + s[1] * 9617 + s[2] * 1868 + 8192) >> 14);
    }
}

/* ------------------------------------------------------------------ A2 */
/* resize(.., INTER_LINEAR) 8UC1: 11-bit coefficients, 32-bit row buffer,
 * vertical pass ((b0*(T0>>4))>>16 + (b1*(T1>>4))>>16 + 2)>>2.  src/Feature.cpp:813 */
void orc_resize_linear(const uint8_t* src, int sw, int sh, int sstride,
                       uint8_t* dst, int dw, int dh, int dstride)
{
    double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    if (scale_x == 2.0 && scale_y == 2.0) {
        /* OpenCV 3.x: INTER_LINEAR with an exact 2:1 shrink in both directions silently becomes
         * INTER_AREA, whose 8U 2x2 fast path is (a+b+c+d+2)>>2 (SURVEY App. A2 note) */
        for (int dy = 0; dy < dh; ++dy) {
            const uint8_t* S0 = src + (size_t)(2 * dy) * sstride; const uint8_t* S1 = S0 + sstride;
            for (int dx = 0; dx < dw; ++dx)
                dst[(size_t)dy * dstride + dx] = (uint8_t)((S0[2 * dx] + S0[2 * dx + 1] + S1[2 * dx] + S1[2 * dx + 1] + 2) >> 2);
        }
        return;
    }
    int* xofs = (int*)malloc(sizeof(int) * dw);
    short* ialpha = (short*)malloc(sizeof(short) * 2 * dw);
    int* T0 = (int*)malloc(sizeof(int) * dw);
    int* T1 = (int*)malloc(sizeof(int) * dw);
    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        ialpha[2 * dx]     = sat_short((1.f - fx) * 2048);
        ialpha[2 * dx + 1] = sat_short(fx * 2048);
    }
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor(fy);
        fy -= sy;
        short b0 = sat_short((1.f - fy) * 2048), b1 = sat_short(fy * 2048);
        int sy0 = sy < 0 ? 0 : sy >= sh ? sh - 1 : sy;          /* row clip */
        int sy1 = sy + 1 < 0 ? 0 : sy + 1 >= sh ? sh - 1 : sy + 1;
        const uint8_t* S0 = src + (size_t)sy0 * sstride;
        const uint8_t* S1 = src + (size_t)sy1 * sstride;
        for (int dx = 0; dx < dw; ++dx) {
            int sx = xofs[dx], sx1 = sx + 1 < sw ? sx + 1 : sw - 1;
            int a0 = ialpha[2 * dx], a1 = ialpha[2 * dx + 1];
            T0[dx] = S0[sx] * a0 + S0[sx1] * a1;
            T1[dx] = S1[sx] * a0 + S1[sx1] * a1;
        }
        uint8_t* D = dst + (size_t)dy * dstride;
        for (int dx = 0; dx < dw; ++dx)
            D[dx] = (uint8_t)((((b0 * (T0[dx] >> 4)) >> 16) + ((b1 * (T1[dx] >> 4)) >> 16) + 2) >> 2);
    }
    free(xofs); free(ialpha); free(T0); free(T1);
}

/* ------------------------------------------------------------------ A7 */
static int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) { if (p < 0) p = -p; else p = 2 * len - 2 - p; }
    return p;
}
/* copyMakeBorder(.., BORDER_REFLECT_101 [+ISOLATED]) in place. src/Feature.cpp:815-821 */
void orc_border_reflect101(uint8_t* buf, int w, int h, int stride, int pad)
{
    for (int y = 0; y < h; ++y) {
        uint8_t* row = buf + (size_t)(y + pad) * stride + pad;
        for (int i = 1; i <= pad; ++i) {
            row[-i] = row[reflect101(-i, w)];
            row[w - 1 + i] = row[reflect101(w - 1 + i, w)];
        }
    }
    for (int i = 1; i <= pad; ++i) {
        memcpy(buf + (size_t)(pad - i) * stride, buf + (size_t)(pad + reflect101(-i, h)) * stride, (size_t)w + 2 * pad);
        memcpy(buf + (size_t)(pad + h - 1 + i) * stride, buf + (size_t)(pad + reflect101(h - 1 + i, h)) * stride, (size_t)w + 2 * pad);
    }
}

/* ------------------------------------------------------------------ A3 */
static const int k_ring[16][2] = { {0,3},{1,3},{2,2},{3,1},{3,0},{3,-1},{2,-2},{1,-3},
                                   {0,-3},{-1,-3},{-2,-2},{-3,-1},{-3,0},{-3,1},{-2,2},{-1,3} };
static void make_offsets(int pixel[25], int stride)
{
    for (int k = 0; k < 16; ++k) pixel[k] = k_ring[k][0] + k_ring[k][1] * stride;
    for (int k = 16; k < 25; ++k) pixel[k] = pixel[k - 16];
}
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* cornerScore<16> of OpenCV's FAST */
static int corner_score16(const uint8_t* ptr, const int pixel[25], int threshold)
{
    int d[25], v = ptr[0];
    for (int k = 0; k < 25; ++k) d[k] = (short)(v - ptr[pixel[k]]);
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = imin(d[k + 1], d[k + 2]);
        a = imin(a, d[k + 3]);
        if (a <= a0) continue;
        a = imin(a, d[k + 4]); a = imin(a, d[k + 5]); a = imin(a, d[k + 6]);
        a = imin(a, d[k + 7]); a = imin(a, d[k + 8]);
        a0 = imax(a0, imin(a, d[k]));
        a0 = imax(a0, imin(a, d[k + 9]));
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = imax(d[k + 1], d[k + 2]);
        b = imax(b, d[k + 3]); b = imax(b, d[k + 4]); b = imax(b, d[k + 5]);
        if (b >= b0) continue;
        b = imax(b, d[k + 6]); b = imax(b, d[k + 7]); b = imax(b, d[k + 8]);
        b0 = imin(b0, imax(b, d[k]));
        b0 = imin(b0, imax(b, d[k + 9]));
    }
    return -b0 - 1;
}
int orc_fast_score_ref(const uint8_t* p, int stride, int threshold)
{
    int pixel[25]; make_offsets(pixel, stride);
    return corner_score16(p, pixel, threshold);
}

/* cv::FAST(img, kps, threshold, nonmaxSuppression=true), TYPE_9_16.
 * Rows/cols 3..n-4 of the Mat passed are tested; NMS is strict '>' over the 8
 * neighbours' stored scores (0 where not a corner / not tested); raster order. */
int orc_fast9(const uint8_t* img, int w, int h, int stride, int threshold,
              int* xs, int* ys, int* scores, int cap)
{
    const int K = 8, N = 25;
    int pixel[25]; make_offsets(pixel, stride);
    if (threshold < 0) threshold = 0;
    if (threshold > 255) threshold = 255;
    uint8_t tab[512];
    for (int i = -255; i <= 255; ++i) tab[i + 255] = (uint8_t)(i < -threshold ? 1 : i > threshold ? 2 : 0);
    if (w < 1 || h < 1) return 0;
    uint8_t* sc = (uint8_t*)calloc((size_t)w * h, 1);     /* full score map instead of 3 rolling rows */
    for (int i = 3; i < h - 3; ++i) {
        const uint8_t* ptr = img + (size_t)i * stride + 3;
        for (int j = 3; j < w - 3; ++j, ++ptr) {
            int v = ptr[0];
            const uint8_t* t = &tab[0] - v + 255;
            int d = t[ptr[pixel[0]]] | t[ptr[pixel[8]]];
            if (d == 0) continue;
            d &= t[ptr[pixel[2]]] | t[ptr[pixel[10]]];
            d &= t[ptr[pixel[4]]] | t[ptr[pixel[12]]];
            d &= t[ptr[pixel[6]]] | t[ptr[pixel[14]]];
            if (d == 0) continue;
            d &= t[ptr[pixel[1]]] | t[ptr[pixel[9]]];
            d &= t[ptr[pixel[3]]] | t[ptr[pixel[11]]];
            d &= t[ptr[pixel[5]]] | t[ptr[pixel[13]]];
            d &= t[ptr[pixel[7]]] | t[ptr[pixel[15]]];
            int is_corner = 0;
            if (d & 1) {
                int vt = v - threshold, count = 0;
                for (int k = 0; k < N; ++k) {
                    if (ptr[pixel[k]] < vt) { if (++count > K) { is_corner = 1; break; } }
                    else count = 0;
                }
            }
            if (!is_corner && (d & 2)) {
                int vt = v + threshold, count = 0;
                for (int k = 0; k < N; ++k) {
                    if (ptr[pixel[k]] > vt) { if (++count > K) { is_corner = 1; break; } }
                    else count = 0;
                }
            }
            if (is_corner) sc[(size_t)i * w + j] = (uint8_t)corner_score16(ptr, pixel, threshold);
        }
    }
    int n = 0;
    for (int i = 3; i < h - 3; ++i)
        for (int j = 3; j < w - 3; ++j) {
            const uint8_t* c = sc + (size_t)i * w + j;
            int s = c[0];
            if (!s) continue;      /* a corner's score is >= threshold-1+1 > 0 for threshold >= 1; see note */
            if (s > c[-1] && s > c[1] && s > c[-w - 1] && s > c[-w] && s > c[-w + 1] &&
                s > c[w - 1] && s > c[w] && s > c[w + 1]) {
                if (n < cap) { xs[n] = j; ys[n] = i; scores[n] = s; }
                ++n;
            }
        }
    /* note: with threshold 0 a corner can score 0 and OpenCV would still list it when all
     * neighbours are 0 ("0 > 0" is false, so it would NOT): identical outcome. */
    free(sc);
    return n;
}

/* ------------------------------------------------------------------ A4 */
void orc_gauss_kernel_fixed(int k[7])
{
    /* getGaussianKernel(7, 2, CV_32F): double exp, float store, normalise; then
     * convertTo(CV_32S, 256) = cvRound(k*256). */
    float cf[7]; double sum = 0;
    double scale2X = -0.5 / (2.0 * 2.0);
    for (int i = 0; i < 7; ++i) { double x = i - 3.0; double t = exp(scale2X * x * x); cf[i] = (float)t; sum += cf[i]; }
    sum = 1. / sum;
    for (int i = 0; i < 7; ++i) { cf[i] = (float)(cf[i] * sum); k[i] = orc_cvround((double)cf[i] * 256.0); }
}
/* GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) on an isolated w x h image:
 * row pass int32 sum k[i]*I, column pass (sum k[j]*R + 32768) >> 16, saturated. src/Feature.cpp:547-548 */
void orc_gaussian7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride)
{
    int k[7]; orc_gauss_kernel_fixed(k);
    int* R = (int*)malloc(sizeof(int) * (size_t)w * h);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int s = 0;
            for (int i = -3; i <= 3; ++i) s += k[i + 3] * src[(size_t)y * sstride + reflect101(x + i, w)];
            R[(size_t)y * w + x] = s;
        }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int s = 0;
            for (int j = -3; j <= 3; ++j) s += k[j + 3] * R[(size_t)reflect101(y + j, h) * w + x];
            s = (s + 32768) >> 16;
            dst[(size_t)y * dstride + x] = (uint8_t)(s < 0 ? 0 : s > 255 ? 255 : s);
        }
    free(R);
}

/* ------------------------------------------------------------------ A5 */
float orc_fast_atan2(float y, float x)
{
    static const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
    static const float p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
    static const float p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
    static const float p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* Canonical cosf/sinf for src/Feature.cpp:449: evaluate in double with a fixed
 * sequence of IEEE operations (Cody-Waite reduction by pi/2, fdlibm kernel
 * polynomials, explicit fma) and round once to float.  The GPU kernel runs the
 * same sequence, so both sides agree bit for bit; tests pin it to libm's
 * cosf/sinf on a dense sample. */
void orc_sincosf(float xf, float* s, float* c)
{
    static const double TWO_OVER_PI = 6.36619772367581382433e-01;
    static const double PIO2_1 = 1.57079632673412561417e+00, PIO2_1T = 6.07710050650619224932e-11;
    static const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
        S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
        S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    static const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
        C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
        C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double x = (double)xf;
    double kd = rint(x * TWO_OVER_PI);
    int k = (int)kd;
    double r = fma(-kd, PIO2_1, x);
    r = fma(-kd, PIO2_1T, r);
    double z = r * r;
    double ps = fma(z, S6, S5); ps = fma(z, ps, S4); ps = fma(z, ps, S3); ps = fma(z, ps, S2); ps = fma(z, ps, S1);
    double sn = fma(r * z, ps, r);
    double pc = fma(z, C6, C5); pc = fma(z, pc, C4); pc = fma(z, pc, C3); pc = fma(z, pc, C2); pc = fma(z, pc, C1);
    double cs = fma(z * z, pc, fma(z, -0.5, 1.0));
    double so, co;
    switch (k & 3) {
        case 0: so = sn; co = cs; break;
        case 1: so = cs; co = -sn; break;
        case 2: so = -sn; co = -cs; break;
        default: so = -cs; co = sn; break;
    }
    *s = (float)so; *c = (float)co;
}

/* ------------------------------------------------------------------ IC_Angle, src/Feature.cpp:84-113 */
static int g_umax[16];
static void build_umax(int umax[16])   /* src/Feature.cpp:425-443 */
{
    int v, v0, vmax = cv_floor(HALF_PATCH * sqrt(2.f) / 2 + 1);
    int vmin = cv_ceil(HALF_PATCH * sqrt(2.f) / 2);
    const double hp2 = HALF_PATCH * HALF_PATCH;
    for (v = 0; v <= vmax; ++v) umax[v] = orc_cvround(sqrt(hp2 - v * v));
    for (v = HALF_PATCH, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
}
float orc_ic_angle(const uint8_t* img, int stride, int x, int y, int* pm01, int* pm10)
{
    if (!g_umax[0]) build_umax(g_umax);
    int m_01 = 0, m_10 = 0;
    const uint8_t* center = img + (size_t)y * stride + x;
    for (int u = -HALF_PATCH; u <= HALF_PATCH; ++u) m_10 += u * center[u];
    for (int v = 1; v < HALF_PATCH; ++v) {           /* v = 1..14 as written (:98) */
        int v_sum = 0, d = g_umax[v];
        for (int u = -d; u < d; ++u) {               /* u = -d..d-1 as written (:102) */
            int val_plus = center[u + v * stride], val_minus = center[u - v * stride];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    if (pm01) *pm01 = m_01;
    if (pm10) *pm10 = m_10;
    return orc_fast_atan2((float)m_01, (float)m_10);
}

/* ------------------------------------------------------------------ rBRIEF, src/Feature.cpp:446-482 */
void orc_orb_descriptor(const uint8_t* img, int stride, int x, int y, float angle_deg, uint8_t desc[32])
{
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);   /* :116 */
    float angle = (float)angle_deg * factorPI;
    float a, b;
    orc_sincosf(angle, &b, &a);                     /* a = cos, b = sin (:449) */
    const uint8_t* center = img + (size_t)y * stride + x;
    const int8_t* p = k_pattern;
    for (int i = 0; i < 32; ++i) {
        int val = 0;
        for (int k = 0; k < 8; ++k, p += 4) {
            float x0 = (float)p[0], y0 = (float)p[1], x1 = (float)p[2], y1 = (float)p[3];
            int t0 = center[orc_cvround(x0 * b + y0 * a) * stride + orc_cvround(x0 * a - y0 * b)];
            int t1 = center[orc_cvround(x1 * b + y1 * a) * stride + orc_cvround(x1 * a - y1 * b)];
            val |= (t0 < t1) << k;
        }
        desc[i] = (uint8_t)val;
    }
}

/* Same descriptor with the host libm's cosf/sinf -- what the reference binary would call
 * (src/Feature.cpp:449) on THIS machine.  glibc's cosf/sinf are faithfully but not correctly
 * rounded (and differ between its FMA and non-FMA ifunc variants), so they are no portable
 * definition; this entry point exists so tests can count how many descriptors the choice of
 * libm can change at all. */
void orc_orb_descriptor_libm(const uint8_t* img, int stride, int x, int y, float angle_deg, uint8_t desc[32])
{
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    float angle = (float)angle_deg * factorPI;
    float a = cosf(angle), b = sinf(angle);
    const uint8_t* center = img + (size_t)y * stride + x;
    const int8_t* p = k_pattern;
    for (int i = 0; i < 32; ++i) {
        int val = 0;
        for (int k = 0; k < 8; ++k, p += 4) {
            float x0 = (float)p[0], y0 = (float)p[1], x1 = (float)p[2], y1 = (float)p[3];
            int t0 = center[orc_cvround(x0 * b + y0 * a) * stride + orc_cvround(x0 * a - y0 * b)];
            int t1 = center[orc_cvround(x1 * b + y1 * a) * stride + orc_cvround(x1 * a - y1 * b)];
            val |= (t0 < t1) << k;
        }
        desc[i] = (uint8_t)val;
    }
}

/* ------------------------------------------------------------------ quadtree, src/Feature.cpp:17-81, 570-798 */
typedef struct QNode {
    int ulx, uly, urx, ury, blx, bly, brx, bry;
    int* keys; int nkeys;
    int no_more;
    long seq;                       /* creation order: the canonical stand-in for the heap-pointer tie-break */
    struct QNode *prev, *next;
} QNode;
typedef struct { QNode *head, *tail; int size; } QList;

static void ql_push_front(QList* l, QNode* n) { n->prev = 0; n->next = l->head; if (l->head) l->head->prev = n; else l->tail = n; l->head = n; l->size++; }
static void ql_push_back(QList* l, QNode* n)  { n->next = 0; n->prev = l->tail; if (l->tail) l->tail->next = n; else l->head = n; l->tail = n; l->size++; }
static QNode* ql_erase(QList* l, QNode* n)
{
    QNode* nx = n->next;
    if (n->prev) n->prev->next = n->next; else l->head = n->next;
    if (n->next) n->next->prev = n->prev; else l->tail = n->prev;
    l->size--; free(n->keys); free(n);
    return nx;
}
static QNode* qn_new(int cap) { QNode* n = (QNode*)calloc(1, sizeof(QNode)); n->keys = (int*)malloc(sizeof(int) * (cap > 0 ? cap : 1)); return n; }

/* ExtractorNode::DivideNode (:17-81) */
static void divide_node(const QNode* p, const float* xs, const float* ys, QNode* c[4])
{
    const int halfX = (int)ceilf((float)(p->urx - p->ulx) / 2);
    const int halfY = (int)ceilf((float)(p->bry - p->uly) / 2);
    for (int i = 0; i < 4; ++i) c[i] = qn_new(p->nkeys);
    QNode *n1 = c[0], *n2 = c[1], *n3 = c[2], *n4 = c[3];
    n1->ulx = p->ulx; n1->uly = p->uly; n1->urx = p->ulx + halfX; n1->ury = p->uly;
    n1->blx = p->ulx; n1->bly = p->uly + halfY; n1->brx = p->ulx + halfX; n1->bry = p->uly + halfY;
    n2->ulx = n1->urx; n2->uly = n1->ury; n2->urx = p->urx; n2->ury = p->ury;
    n2->blx = n1->brx; n2->bly = n1->bry; n2->brx = p->urx; n2->bry = p->uly + halfY;
    n3->ulx = n1->blx; n3->uly = n1->bly; n3->urx = n1->brx; n3->ury = n1->bry;
    n3->blx = p->blx; n3->bly = p->bly; n3->brx = n1->brx; n3->bry = p->bly;
    n4->ulx = n3->urx; n4->uly = n3->ury; n4->urx = n2->brx; n4->ury = n2->bry;
    n4->blx = n3->brx; n4->bly = n3->bry; n4->brx = p->brx; n4->bry = p->bry;
    for (int i = 0; i < p->nkeys; ++i) {
        int k = p->keys[i];
        if (xs[k] < (float)n1->urx) {
            if (ys[k] < (float)n1->bry) n1->keys[n1->nkeys++] = k; else n3->keys[n3->nkeys++] = k;
        } else if (ys[k] < (float)n1->bry) n2->keys[n2->nkeys++] = k;
        else n4->keys[n4->nkeys++] = k;
    }
    for (int i = 0; i < 4; ++i) if (c[i]->nkeys == 1) c[i]->no_more = 1;
}

typedef struct { int size; long seq; QNode* node; } SizeNode;
static int cmp_sizenode(const void* a, const void* b)
{
    const SizeNode* x = (const SizeNode*)a; const SizeNode* y = (const SizeNode*)b;
    if (x->size != y->size) return x->size < y->size ? -1 : 1;
    return x->seq < y->seq ? -1 : x->seq > y->seq ? 1 : 0;
}

/* push the non-empty children in order n1..n4 at the front (:655-694 / :723-758) */
static void push_children(QList* l, QNode* c[4], long* seq, SizeNode* vec, int* nvec, int* nToExpand)
{
    for (int i = 0; i < 4; ++i) {
        if (c[i]->nkeys > 0) {
            c[i]->seq = (*seq)++;
            ql_push_front(l, c[i]);
            if (c[i]->nkeys > 1) {
                if (nToExpand) (*nToExpand)++;
                vec[*nvec].size = c[i]->nkeys; vec[*nvec].seq = c[i]->seq; vec[*nvec].node = c[i]; (*nvec)++;
            }
        } else { free(c[i]->keys); free(c[i]); }
    }
}

int orc_distribute_octtree(const float* xs, const float* ys, const float* resp, int n,
                           int minX, int maxX, int minY, int maxY, int N, int* out_idx, int cap, int* tie_flag)
{
    if (tie_flag) *tie_flag = 0;
    const int nIni = (int)roundf((float)(maxX - minX) / (maxY - minY));           /* :574 */
    if (nIni < 1) return -1;                                                      /* reference divides by zero here */
    const float hX = (float)(maxX - minX) / nIni;                                 /* :576 */
    QList L = {0, 0, 0};
    long seq = 1;
    QNode** init = (QNode**)malloc(sizeof(QNode*) * nIni);
    for (int i = 0; i < nIni; ++i) {                                              /* :582-593 */
        QNode* ni = qn_new(n);
        ni->ulx = (int)(hX * (float)i); ni->uly = 0;
        ni->urx = (int)(hX * (float)(i + 1)); ni->ury = 0;
        ni->blx = ni->ulx; ni->bly = maxY - minY;
        ni->brx = ni->urx; ni->bry = maxY - minY;
        ni->seq = -(long)i;
        ql_push_back(&L, ni);
        init[i] = ni;
    }
    for (int i = 0; i < n; ++i) {                                                 /* :596-600 */
        int b = (int)(xs[i] / hX);
        if (b < 0 || b >= nIni) { b = b < 0 ? 0 : nIni - 1; }                     /* out of contract; reference would index out of range */
        init[b]->keys[init[b]->nkeys++] = i;
    }
    free(init);
    for (QNode* it = L.head; it;) {                                               /* :604-619 */
        if (it->nkeys == 1) { it->no_more = 1; it = it->next; }
        else if (it->nkeys == 0) it = ql_erase(&L, it);
        else it = it->next;
    }
    int bFinish = 0;
    SizeNode* vec = (SizeNode*)malloc(sizeof(SizeNode) * (size_t)(4 * (n + nIni) + 4));
    SizeNode* prevvec = (SizeNode*)malloc(sizeof(SizeNode) * (size_t)(4 * (n + nIni) + 4));
    int nvec = 0;
    while (!bFinish) {                                                            /* :628 */
        int prevSize = L.size, nToExpand = 0;
        nvec = 0;
        for (QNode* it = L.head; it;) {                                           /* :640-699 */
            if (it->no_more) { it = it->next; continue; }
            QNode* c[4];
            divide_node(it, xs, ys, c);
            push_children(&L, c, &seq, vec, &nvec, &nToExpand);
            it = ql_erase(&L, it);
        }
        if (L.size >= N || L.size == prevSize) bFinish = 1;                       /* :703 */
        else if (L.size + nToExpand * 3 > N) {                                    /* :707 */
            while (!bFinish) {
                prevSize = L.size;
                int nprev = nvec;
                memcpy(prevvec, vec, sizeof(SizeNode) * (size_t)nprev);
                nvec = 0;
                qsort(prevvec, (size_t)nprev, sizeof(SizeNode), cmp_sizenode);    /* :716 (size, then creation order) */
                for (int j = nprev - 1; j >= 0; --j) {
                    QNode* c[4];
                    divide_node(prevvec[j].node, xs, ys, c);
                    push_children(&L, c, &seq, vec, &nvec, 0);
                    ql_erase(&L, prevvec[j].node);
                    if (L.size >= N) {                                            /* :762 */
                        if (tie_flag && j > 0 && prevvec[j - 1].size == prevvec[j].size) *tie_flag = 1;
                        break;
                    }
                }
                if (L.size >= N || L.size == prevSize) bFinish = 1;               /* :769 */
            }
        }
    }
    int nout = 0;
    for (QNode* it = L.head; it; it = it->next) {                                 /* :778-796 */
        int best = it->keys[0];
        float maxResponse = resp[best];
        for (int k = 1; k < it->nkeys; ++k)
            if (resp[it->keys[k]] > maxResponse) { best = it->keys[k]; maxResponse = resp[best]; }
        if (nout < cap) out_idx[nout] = best;
        ++nout;
    }
    while (L.head) ql_erase(&L, L.head);
    free(vec); free(prevvec);
    return nout;
}

/* ------------------------------------------------------------------ extractor */
struct orc_extractor {
    int nfeatures, nlevels, iniTh, minTh;
    double scaleFactor;                       /* include/Feature.h:89 (double member, float ctor arg) */
    float sf[MAXLV], isf[MAXLV], sig2[MAXLV], isig2[MAXLV];
    int quota[MAXLV];
    int umax[16];
    /* state of the last extract */
    int lw[MAXLV], lh[MAXLV], lstride[MAXLV];
    uint8_t* lbuf[MAXLV];                     /* padded pyramid level */
    uint8_t* blur[MAXLV]; int bstride[MAXLV];
    float *cx[MAXLV], *cy[MAXLV], *cr[MAXLV]; int ncand[MAXLV];
    orc_keypoint* kps[MAXLV]; int nkps[MAXLV];
    int tie[MAXLV];
};

orc_extractor* orc_create(int nfeatures, float scaleFactorF, int nlevels, int iniTh, int minTh)
{
    if (nlevels < 1 || nlevels > MAXLV) return 0;
    orc_extractor* e = (orc_extractor*)calloc(1, sizeof(*e));
    e->nfeatures = nfeatures; e->nlevels = nlevels; e->iniTh = iniTh; e->minTh = minTh;
    e->scaleFactor = (double)scaleFactorF;
    e->sf[0] = 1.0f; e->sig2[0] = 1.0f;
    for (int i = 1; i < nlevels; ++i) {                                            /* :388-392 */
        e->sf[i] = (float)((double)e->sf[i - 1] * e->scaleFactor);
        e->sig2[i] = e->sf[i] * e->sf[i];
    }
    for (int i = 0; i < nlevels; ++i) { e->isf[i] = 1.0f / e->sf[i]; e->isig2[i] = 1.0f / e->sig2[i]; }   /* :397-401 */
    float factor = (float)(1.0f / e->scaleFactor);                                 /* :407 */
    float nDesired = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));   /* :408 */
    int sum = 0;
    for (int level = 0; level < nlevels - 1; ++level) {                            /* :411-416 */
        e->quota[level] = orc_cvround(nDesired);
        sum += e->quota[level];
        nDesired *= factor;
    }
    e->quota[nlevels - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0;             /* :417 */
    build_umax(e->umax);
    if (!g_umax[0]) build_umax(g_umax);
    return e;
}
static void free_state(orc_extractor* e)
{
    for (int l = 0; l < MAXLV; ++l) {
        free(e->lbuf[l]); free(e->blur[l]); free(e->cx[l]); free(e->cy[l]); free(e->cr[l]); free(e->kps[l]);
        e->lbuf[l] = e->blur[l] = 0; e->cx[l] = e->cy[l] = e->cr[l] = 0; e->kps[l] = 0;
        e->ncand[l] = e->nkps[l] = 0; e->tie[l] = 0;
    }
}
void orc_destroy(orc_extractor* e) { if (e) { free_state(e); free(e); } }
const float* orc_scale_factors(orc_extractor* e) { return e->sf; }
const float* orc_inv_scale_factors(orc_extractor* e) { return e->isf; }
const float* orc_level_sigma2(orc_extractor* e) { return e->sig2; }
const float* orc_inv_level_sigma2(orc_extractor* e) { return e->isig2; }
const int* orc_features_per_level(orc_extractor* e) { return e->quota; }
const int* orc_umax(orc_extractor* e) { return e->umax; }

int orc_cell_grid(int cols, int rows, orc_grid* g)          /* :835-849 */
{
    g->minBX = EDGE_TH - 3; g->minBY = g->minBX;
    g->maxBX = cols - EDGE_TH + 3; g->maxBY = rows - EDGE_TH + 3;
    const float width = (float)(g->maxBX - g->minBX), height = (float)(g->maxBY - g->minBY);
    g->nCols = (int)(width / 30.f); g->nRows = (int)(height / 30.f);
    if (g->nCols < 1 || g->nRows < 1) return -1;            /* reference divides by zero */
    g->wCell = (int)ceilf(width / g->nCols); g->hCell = (int)ceilf(height / g->nRows);
    return 0;
}

int orc_extract(orc_extractor* e, const uint8_t* image, int cols, int rows, int stride,
                orc_keypoint* out_kps, uint8_t* out_desc, int cap)
{
    if (!image || cols <= 0 || rows <= 0) return 0;                               /* :507-508 */
    free_state(e);
    const int L = e->nlevels;
    /* geometry check first (the reference has undefined behaviour on degenerate levels) */
    for (int l = 0; l < L; ++l) {
        float scale = e->isf[l];
        e->lw[l] = orc_cvround((float)cols * scale); e->lh[l] = orc_cvround((float)rows * scale);   /* :805 */
        orc_grid g; if (orc_cell_grid(e->lw[l], e->lh[l], &g)) return -1;
        int nIni = (int)roundf((float)(g.maxBX - g.minBX) / (g.maxBY - g.minBY));
        if (nIni < 1) return -1;
    }
    /* ComputePyramid :800-824 */
    for (int l = 0; l < L; ++l) {
        int w = e->lw[l], h = e->lh[l], st = w + 2 * EDGE_TH;
        e->lstride[l] = st;
        e->lbuf[l] = (uint8_t*)calloc((size_t)st * (h + 2 * EDGE_TH), 1);
        uint8_t* roi = e->lbuf[l] + (size_t)EDGE_TH * st + EDGE_TH;
        if (l == 0) for (int y = 0; y < h; ++y) memcpy(roi + (size_t)y * st, image + (size_t)y * stride, (size_t)w);
        else orc_resize_linear(e->lbuf[l - 1] + (size_t)EDGE_TH * e->lstride[l - 1] + EDGE_TH, e->lw[l - 1], e->lh[l - 1], e->lstride[l - 1],
                               roi, w, h, st);
        orc_border_reflect101(e->lbuf[l], w, h, st, EDGE_TH);
    }
    /* ComputeKeyPointsOctTree :826-916 */
    for (int l = 0; l < L; ++l) {
        const int w = e->lw[l], h = e->lh[l], st = e->lstride[l];
        const uint8_t* roi = e->lbuf[l] + (size_t)EDGE_TH * st + EDGE_TH;
        orc_grid g; orc_cell_grid(w, h, &g);
        int ccap = 1024, nc = 0;
        float *cx = (float*)malloc(sizeof(float) * ccap), *cy = (float*)malloc(sizeof(float) * ccap), *cr = (float*)malloc(sizeof(float) * ccap);
        int tmpcap = 4096; int *tx = (int*)malloc(sizeof(int) * tmpcap), *ty = (int*)malloc(sizeof(int) * tmpcap), *ts = (int*)malloc(sizeof(int) * tmpcap);
        for (int i = 0; i < g.nRows; ++i) {
            const float iniY = (float)(g.minBY + i * g.hCell);
            float maxY = iniY + g.hCell + 6;
            if (iniY >= g.maxBY - 3) continue;                                     /* :856 */
            if (maxY > g.maxBY) maxY = (float)g.maxBY;
            for (int j = 0; j < g.nCols; ++j) {
                const float iniX = (float)(g.minBX + j * g.wCell);
                float maxX = iniX + g.wCell + 6;
                if (iniX >= g.maxBX - 6) continue;                                 /* :866 */
                if (maxX > g.maxBX) maxX = (float)g.maxBX;
                int x0 = (int)iniX, y0 = (int)iniY, ww = (int)maxX - x0, wh = (int)maxY - y0;
                const uint8_t* win = roi + (size_t)y0 * st + x0;
                int nk = orc_fast9(win, ww, wh, st, e->iniTh, tx, ty, ts, tmpcap);        /* :872 */
                if (nk == 0) nk = orc_fast9(win, ww, wh, st, e->minTh, tx, ty, ts, tmpcap); /* :875-879 */
                if (nk > tmpcap) nk = tmpcap;   /* cannot happen: <= 64*64/4 */
                for (int k = 0; k < nk; ++k) {                                      /* :883-889 */
                    if (nc == ccap) { ccap *= 2; cx = (float*)realloc(cx, sizeof(float) * ccap); cy = (float*)realloc(cy, sizeof(float) * ccap); cr = (float*)realloc(cr, sizeof(float) * ccap); }
                    cx[nc] = (float)tx[k] + j * g.wCell; cy[nc] = (float)ty[k] + i * g.hCell; cr[nc] = (float)ts[k]; ++nc;
                }
            }
        }
        free(tx); free(ty); free(ts);
        e->cx[l] = cx; e->cy[l] = cy; e->cr[l] = cr; e->ncand[l] = nc;
        int* sel = (int*)malloc(sizeof(int) * (size_t)(nc + 8));
        int tie = 0;
        int nsel = orc_distribute_octtree(cx, cy, cr, nc, g.minBX, g.maxBX, g.minBY, g.maxBY, e->quota[l], sel, nc + 8, &tie);   /* :897 */
        if (nsel < 0) nsel = 0;
        e->tie[l] = tie;
        const int scalePatchSize = (int)(HALF_PATCH * e->sf[l]);                    /* :900 */
        e->kps[l] = (orc_keypoint*)malloc(sizeof(orc_keypoint) * (size_t)(nsel + 1));
        e->nkps[l] = nsel;
        for (int i = 0; i < nsel; ++i) {                                            /* :903-909 */
            orc_keypoint* kp = &e->kps[l][i];
            kp->x = cx[sel[i]] + g.minBX; kp->y = cy[sel[i]] + g.minBY;
            kp->size = (float)scalePatchSize; kp->angle = -1.f; kp->response = cr[sel[i]];
            kp->octave = l; kp->class_id = -1;
        }
        free(sel);
    }
    for (int l = 0; l < L; ++l) {                                                   /* :912-915 */
        const uint8_t* roi = e->lbuf[l] + (size_t)EDGE_TH * e->lstride[l] + EDGE_TH;
        for (int i = 0; i < e->nkps[l]; ++i)
            e->kps[l][i].angle = orc_ic_angle(roi, e->lstride[l], orc_cvround(e->kps[l][i].x), orc_cvround(e->kps[l][i].y), 0, 0);
    }
    /* Extract :519-567 */
    int total = 0;
    for (int l = 0; l < L; ++l) total += e->nkps[l];
    int n = 0;
    for (int l = 0; l < L; ++l) {
        if (e->nkps[l] == 0) continue;                                              /* :543-544 */
        int w = e->lw[l], h = e->lh[l];
        e->bstride[l] = w;
        e->blur[l] = (uint8_t*)malloc((size_t)w * h);
        orc_gaussian7(e->lbuf[l] + (size_t)EDGE_TH * e->lstride[l] + EDGE_TH, w, h, e->lstride[l], e->blur[l], w);   /* :547-548 */
        for (int i = 0; i < e->nkps[l]; ++i, ++n) {
            const orc_keypoint* kp = &e->kps[l][i];
            if (n >= cap) continue;
            orc_orb_descriptor(e->blur[l], w, orc_cvround(kp->x), orc_cvround(kp->y), kp->angle, out_desc + (size_t)n * 32);   /* :552 */
            out_kps[n] = *kp;
            if (l != 0) { float scale = e->sf[l]; out_kps[n].x = kp->x * scale; out_kps[n].y = kp->y * scale; }   /* :557-563 */
        }
    }
    (void)total;
    return n;
}

int orc_extract_rgb(orc_extractor* e, const uint8_t* rgb, int w, int h, int stride,
                    orc_keypoint* kps, uint8_t* desc, int cap)                      /* src/Frame.cpp:3-9 */
{
    if (!rgb || w <= 0 || h <= 0) return 0;
    uint8_t* gray = (uint8_t*)malloc((size_t)w * h);
    orc_rgb2gray(rgb, w, h, stride, gray, w);
    int n = orc_extract(e, gray, w, h, w, kps, desc, cap);
    free(gray);
    return n;
}

int orc_level_size(orc_extractor* e, int l, int* w, int* h) { if (l < 0 || l >= e->nlevels || !e->lbuf[l]) return -1; *w = e->lw[l]; *h = e->lh[l]; return 0; }
const uint8_t* orc_level_roi(orc_extractor* e, int l, int* stride) { if (!e->lbuf[l]) return 0; *stride = e->lstride[l]; return e->lbuf[l] + (size_t)EDGE_TH * e->lstride[l] + EDGE_TH; }
const uint8_t* orc_level_blur(orc_extractor* e, int l, int* stride) { if (!e->blur[l]) return 0; *stride = e->bstride[l]; return e->blur[l]; }
int orc_level_candidates(orc_extractor* e, int l, const float** xs, const float** ys, const float** resp) { *xs = e->cx[l]; *ys = e->cy[l]; *resp = e->cr[l]; return e->ncand[l]; }
int orc_level_keypoints(orc_extractor* e, int l, const orc_keypoint** kps) { *kps = e->kps[l]; return e->nkps[l]; }
int orc_level_tie_at_cut(orc_extractor* e, int l) { return e->tie[l]; }
