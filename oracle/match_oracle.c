/* match_oracle.c -- CPU restatement of SSVIO's Hamming matching (TEST INFRASTRUCTURE ONLY).
 *
 * PARITY UNPINNED (see ssvio_oracle.h).  Follows /root/reference/src/FeatureMatching.cpp:
 *   :211-228 ComputeMatchingScore   -> orc_hamming256
 *   :85-112  2-NN + Threshold/NNRatio acceptance -> orc_knn2_raw / orc_match_knn2_ratio
 *   :20-55   MatchByBruteForce      -> orc_match_bruteforce (BFMatcher per SURVEY App. A6)
 *   :128-189 MatchByProject         -> orc_match_project (the DBoW3-free half)
 * Cross-check has no reference code; mutual nearest neighbour as defined in App. A6.
 */
#include "ssvio_oracle.h"
#include <string.h>
#include <limits.h>
#include <stdlib.h>

int orc_hamming256(const uint8_t* a, const uint8_t* b)          /* :211-228, SWAR popcount over 8 x int32 */
{
    int dist = 0;
    for (int i = 0; i < 8; ++i) {
        uint32_t x, y; memcpy(&x, a + 4 * i, 4); memcpy(&y, b + 4 * i, 4);
        uint32_t v = x ^ y;
        v = v - ((v >> 1) & 0x55555555u);
        v = (v & 0x33333333u) + ((v >> 2) & 0x33333333u);
        dist += (int)((((v + (v >> 4)) & 0xF0F0F0Fu) * 0x1010101u) >> 24);
    }
    return dist;
}

void orc_knn2_raw(const uint8_t* q, int nq, int qstride, const uint8_t* t, int nt, int tstride,
                  int32_t* idx, int32_t* d1, int32_t* d2)       /* :85-102 */
{
    for (int i = 0; i < nq; ++i) {
        int bestDist1 = 256, bestDist2 = 256, bestIdx = -1;
        for (int j = 0; j < nt; ++j) {
            int dist = orc_hamming256(q + (size_t)i * qstride, t + (size_t)j * tstride);
            if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx = j; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        idx[i] = bestIdx; d1[i] = bestDist1; d2[i] = bestDist2;
    }
}

void orc_bf_nn(const uint8_t* q, int nq, int qstride, const uint8_t* t, int nt, int tstride,
               int32_t* idx, int32_t* dist)                     /* App. A6: K=1, 'd < best' strict, ascending j */
{
    for (int i = 0; i < nq; ++i) {
        int best = INT_MAX, bi = -1;
        for (int j = 0; j < nt; ++j) {
            int d = orc_hamming256(q + (size_t)i * qstride, t + (size_t)j * tstride);
            if (d < best) { best = d; bi = j; }
        }
        idx[i] = bi; dist[i] = best;
    }
}

int orc_match_bruteforce(const uint8_t* q, int nq, int qstride, const uint8_t* t, int nt, int tstride,
                         orc_dmatch* out)                       /* :20-55 */
{
    if (!(nq > 0 && nt > 0)) return 0;                          /* :25-27 */
    int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * nq);
    int32_t* dist = (int32_t*)malloc(sizeof(int32_t) * nq);
    orc_bf_nn(q, nq, qstride, t, nt, tstride, idx, dist);       /* :30-32 */
    double min_dist = 10000.0, max_dist = 0.;                   /* :35-44 */
    for (int i = 0; i < nq; ++i) {
        double d = (double)(float)dist[i];
        if (d < min_dist) min_dist = d;
        if (d > max_dist) max_dist = d;
    }
    double lim = min_dist * 2 > 30.0 ? min_dist * 2 : 30.0;     /* :49 */
    int n = 0;
    for (int i = 0; i < nq; ++i) {
        double d = (double)(float)dist[i];
        if (d < lim) { out[n].queryIdx = i; out[n].trainIdx = idx[i]; out[n].imgIdx = 0; out[n].distance = (float)dist[i]; ++n; }
    }
    free(idx); free(dist);
    return n;
}

int orc_match_knn2_ratio(const uint8_t* q, int nq, int qstride, const uint8_t* t, int nt, int tstride,
                         int threshold, float nn_ratio, orc_dmatch* out)   /* :85-112 over every row of both sets */
{
    if (!(nq > 0 && nt > 0)) return 0;
    int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * nq);
    int32_t* d1 = (int32_t*)malloc(sizeof(int32_t) * nq);
    int32_t* d2 = (int32_t*)malloc(sizeof(int32_t) * nq);
    orc_knn2_raw(q, nq, qstride, t, nt, tstride, idx, d1, d2);
    int n = 0;
    for (int i = 0; i < nq; ++i) {
        if (d1[i] <= threshold) {                                           /* :104 */
            if ((float)d1[i] < nn_ratio * (float)d2[i]) {                   /* :105 */
                out[n].queryIdx = i; out[n].trainIdx = idx[i]; out[n].imgIdx = -1; out[n].distance = (float)d1[i]; ++n;
            }
        }
    }
    free(idx); free(d1); free(d2);
    return n;
}

/* BFMatcher(NORM_HAMMING, crossCheck = true) as OpenCV 3.x implements it (SURVEY App. A6 "OpenCV-compat", recalled from
 * batchDistance(..., crosscheck = true); unverified here: OpenCV is absent).  Only the REVERSE search is run: every train row t
 * finds its nearest query row tq[t] (lowest query index on ties); then, for t ascending, `if tdist[t] < dist[q] { dist[q] =
 * tdist[t]; nidx[q] = t; }` with dist[] starting at INT_MAX and nidx[] at -1; queries with nidx >= 0 are emitted in ascending
 * order.  A query thus gets the best train AMONG THE TRAINS THAT VOTED FOR IT, which differs from the mutual nearest neighbour
 * when the query's own nearest train prefers another query. */
int orc_match_crosscheck_cv(const uint8_t* q, int nq, int qstride, const uint8_t* t, int nt, int tstride, orc_dmatch* out)
{
    if (!(nq > 0 && nt > 0)) return 0;
    int32_t* ti = (int32_t*)malloc(sizeof(int32_t) * nt); int32_t* td = (int32_t*)malloc(sizeof(int32_t) * nt);
    int32_t* nidx = (int32_t*)malloc(sizeof(int32_t) * nq); int32_t* dist = (int32_t*)malloc(sizeof(int32_t) * nq);
    orc_bf_nn(t, nt, tstride, q, nq, qstride, ti, td);
    for (int i = 0; i < nq; ++i) { nidx[i] = -1; dist[i] = 0x7FFFFFFF; }
    for (int j = 0; j < nt; ++j) {
        const int i = ti[j];
        if (i >= 0 && td[j] < dist[i]) { dist[i] = td[j]; nidx[i] = j; }
    }
    int n = 0;
    for (int i = 0; i < nq; ++i)
        if (nidx[i] >= 0) { out[n].queryIdx = i; out[n].trainIdx = nidx[i]; out[n].imgIdx = 0; out[n].distance = (float)dist[i]; ++n; }
    free(ti); free(td); free(nidx); free(dist);
    return n;
}

int orc_match_crosscheck(const uint8_t* q, int nq, int qstride, const uint8_t* t, int nt, int tstride,
                         orc_dmatch* out)                                   /* App. A6 mutual-NN, lowest-index ties both ways */
{
    if (!(nq > 0 && nt > 0)) return 0;
    int32_t* qi = (int32_t*)malloc(sizeof(int32_t) * nq); int32_t* qd = (int32_t*)malloc(sizeof(int32_t) * nq);
    int32_t* ti = (int32_t*)malloc(sizeof(int32_t) * nt); int32_t* td = (int32_t*)malloc(sizeof(int32_t) * nt);
    orc_bf_nn(q, nq, qstride, t, nt, tstride, qi, qd);
    orc_bf_nn(t, nt, tstride, q, nq, qstride, ti, td);
    int n = 0;
    for (int i = 0; i < nq; ++i)
        if (qi[i] >= 0 && ti[qi[i]] == i) { out[n].queryIdx = i; out[n].trainIdx = qi[i]; out[n].imgIdx = 0; out[n].distance = (float)qd[i]; ++n; }
    free(qi); free(qd); free(ti); free(td);
    return n;
}

/* MatchByProject, src/FeatureMatching.cpp:128-189.  p3d[n1][3], p2d[n2][2], K/R row-major 3x3, t[3], all double.
 * OpenCV arithmetic behind the cv::Mat expressions (canonical choices, unverifiable here like the rest):
 *   R*p3d + t      one gemm with C = t: (r0*x + r1*y + r2*z) summed left to right, then + t, no contraction (:137)
 *   p /= p(2,0)    Mat::operator/=(double) is convertTo(alpha = 1./s): every element is MULTIPLIED by (1.0 / z) (:138)
 * The bounds test against K(0,2) / K(1,2) (the principal point, not the image size) is the reference's (:141). */
int orc_match_project(const double* p3d, const uint8_t* q, int n1, int qstride,
                      const double* p2d, const uint8_t* t, int n2, int tstride,
                      double radian, const double* K, const double* R, const double* tr,
                      int threshold, float nn_ratio, orc_dmatch* out)
{
    int n = 0;
    for (int i = 0; i < n1; ++i) {
        const double x = p3d[3 * i], y = p3d[3 * i + 1], z = p3d[3 * i + 2];
        double px = R[0] * x + R[1] * y + R[2] * z + tr[0];                   /* :137 */
        double py = R[3] * x + R[4] * y + R[5] * z + tr[1];
        const double pz = R[6] * x + R[7] * y + R[8] * z + tr[2];
        const double inv = 1. / pz;                                             /* :138 */
        px = px * inv; py = py * inv;
        const double u = K[0] * px + K[2];                                      /* :139 */
        const double v = K[4] * py + K[5];                                      /* :140 */
        if (u < 0 || u > K[2] || v < 0 || v > K[5]) continue;                   /* :141 (also drops NaN-free out-of-range only) */
        int min_dist = 256, sec_min_dist = 256, best_id = -1;                   /* :160-162 */
        for (int j = 0; j < n2; ++j) {
            const double x2 = p2d[2 * j], y2 = p2d[2 * j + 1];
            if ((u - radian) < x2 && (u + radian) > x2 && (v - radian) < y2 && (v + radian) > y2) {   /* :151-152 */
                const int dist = orc_hamming256(q + (size_t)i * qstride, t + (size_t)j * tstride);
                if (dist < min_dist) { sec_min_dist = min_dist; min_dist = dist; best_id = j; }
                else if (dist < sec_min_dist) sec_min_dist = dist;
            }
        }
        if (min_dist < threshold) {                                             /* :177 (strict, unlike :104) */
            if ((float)min_dist < nn_ratio * (float)sec_min_dist) {             /* :178 */
                orc_dmatch m; m.queryIdx = i; m.trainIdx = best_id; m.imgIdx = -1; m.distance = (float)min_dist;
                out[n++] = m;
            }
        }
    }
    return n;
}
