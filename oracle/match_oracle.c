/* match_oracle.c -- CPU restatement of SSVIO's Hamming matching (TEST INFRASTRUCTURE ONLY).
 *
 * PARITY UNPINNED (see ssvio_oracle.h).  Follows /root/reference/src/FeatureMatching.cpp:
 *   :211-228 ComputeMatchingScore   -> orc_hamming256
 *   :85-112  2-NN + Threshold/NNRatio acceptance -> orc_knn2_raw / orc_match_knn2_ratio
 *   :20-55   MatchByBruteForce      -> orc_match_bruteforce (BFMatcher per SURVEY App. A6)
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
