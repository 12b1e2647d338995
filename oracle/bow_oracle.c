/* bow_oracle.c -- CPU restatement of SSVIO's MatchByDBoW (TEST INFRASTRUCTURE ONLY).
 *
 * PARITY UNPINNED (see ssvio_oracle.h).  Follows /root/reference/src/FeatureMatching.cpp:
 *   :58-125   MatchByDBoW       -> orc_match_dbow
 *   :192-208  ComputeBoWVector  -> orc_bow_transform + orc_bow_vector
 * ComputeBoWVector is one call into DBoW3 (`mVoc_->transform(allDesp, bowVec, featVec, 4)`, :206).
 * DBoW3 (github.com/rmsalinas/DBow3) is linked as /usr/local/lib/libDBoW3.a with no version pin
 * (CMakeLists.txt:54-57) and is absent from the reference tree and from this image; what follows
 * restates its published algorithm (Vocabulary::transform, DBoW3 1.0 src/Vocabulary.cpp):
 *   per feature: start at the root; at every level take the child whose descriptor has the smallest
 *   Hamming distance, the first such child in the parent's children order on ties ('d < best_d');
 *   remember the node reached at level L - levelsup (node 0 when that level is <= 0); stop at a leaf;
 *   the leaf's word id and weight are the result.  A feature whose word has weight 0 ("stopped") enters
 *   neither vector.  BowVector: TF / TF-IDF add the weight per word in feature order, IDF / BINARY keep
 *   the first; with a normalising score (L1 default) the vector is divided by its L1 (or L2) norm,
 *   otherwise TF vectors by their size.  FeatureVector: map node id -> feature indices in ascending order.
 * The vocabulary is a plain node table here (the .dbow3 blob is missing from the reference tree):
 * node 0 is the root, parent[n] < n, the children of a node in ascending node id (the order DBoW3
 * creates and saves them in), word[n] >= 0 marks a leaf.
 */
#include "ssvio_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>

struct orc_vocab {
    int k, L, nnodes, scoring, weighting;
    int32_t* parent; uint8_t* desc; double* weight; int32_t* word;
    int32_t* childBegin; int32_t* child;          /* CSR: children of n = child[childBegin[n] .. childBegin[n+1]) */
};

orc_vocab* orc_vocab_create(int k, int L, int nnodes, const int32_t* parent, const uint8_t* desc,
                            const double* weight, const int32_t* word, int scoring, int weighting)
{
    if (nnodes < 1 || k < 1 || L < 0) return NULL;
    for (int n = 1; n < nnodes; ++n) if (parent[n] < 0 || parent[n] >= n) return NULL;
    orc_vocab* v = (orc_vocab*)calloc(1, sizeof *v);
    v->k = k; v->L = L; v->nnodes = nnodes; v->scoring = scoring; v->weighting = weighting;
    v->parent = (int32_t*)malloc(sizeof(int32_t) * nnodes); memcpy(v->parent, parent, sizeof(int32_t) * nnodes);
    v->desc = (uint8_t*)malloc((size_t)32 * nnodes); memcpy(v->desc, desc, (size_t)32 * nnodes);
    v->weight = (double*)malloc(sizeof(double) * nnodes); memcpy(v->weight, weight, sizeof(double) * nnodes);
    v->word = (int32_t*)malloc(sizeof(int32_t) * nnodes); memcpy(v->word, word, sizeof(int32_t) * nnodes);
    v->childBegin = (int32_t*)calloc((size_t)nnodes + 1, sizeof(int32_t));
    v->child = (int32_t*)malloc(sizeof(int32_t) * (nnodes > 1 ? nnodes - 1 : 1));
    for (int n = 1; n < nnodes; ++n) v->childBegin[parent[n] + 1]++;
    for (int n = 0; n < nnodes; ++n) v->childBegin[n + 1] += v->childBegin[n];
    int32_t* fill = (int32_t*)calloc(nnodes, sizeof(int32_t));
    for (int n = 1; n < nnodes; ++n) { const int p = parent[n]; v->child[v->childBegin[p] + fill[p]++] = n; }  /* ascending id */
    free(fill);
    return v;
}

void orc_vocab_destroy(orc_vocab* v)
{
    if (!v) return;
    free(v->parent); free(v->desc); free(v->weight); free(v->word); free(v->childBegin); free(v->child); free(v);
}

/* Vocabulary::transform(feature, word_id, weight, nid, levelsup) for every row.  A feature that reaches a leaf
 * above level L - levelsup never has its node id assigned in DBoW3 (the variable is left uninitialised there);
 * canonical here: node 0. */
void orc_bow_transform(const orc_vocab* v, const uint8_t* desc, int n, int stride, int levelsup,
                       int32_t* word, int32_t* node, double* weight)
{
    const int nid_level = v->L - levelsup;
    for (int i = 0; i < n; ++i) {
        const uint8_t* f = desc + (size_t)i * stride;
        int final_id = 0, current_level = 0, nid = 0;
        while (v->childBegin[final_id + 1] > v->childBegin[final_id]) {          /* do { } while (!isLeaf) */
            ++current_level;
            int best_d = 1 << 30, best = -1;
            for (int c = v->childBegin[final_id]; c < v->childBegin[final_id + 1]; ++c) {
                const int id = v->child[c];
                const int d = orc_hamming256(f, v->desc + (size_t)32 * id);
                if (d < best_d) { best_d = d; best = id; }
            }
            final_id = best;
            if (current_level == nid_level) nid = final_id;
        }
        word[i] = v->word[final_id]; node[i] = nid; weight[i] = v->weight[final_id];
    }
}

/* BowVector of Vocabulary::transform(features, v, fv, levelsup) from the per-feature words and weights:
 * out_word / out_value in ascending word id; returns the number of entries.
 * weighting: 0 TF_IDF, 1 TF, 2 IDF, 3 BINARY; scoring (DBoW3's ScoringType): 0 L1_NORM, 1 L2_NORM, 2 CHI_SQUARE, 3 KL,
 * 4 BHATTACHARYYA, 5 DOT_PRODUCT -- mustNormalize: L2 for 1, none for 5, L1 for the others. */
int orc_bow_vector(const int32_t* word, const double* weight, int n, int scoring, int weighting,
                   int32_t* out_word, double* out_value)
{
    int m = 0;
    for (int i = 0; i < n; ++i) {
        if (!(weight[i] > 0)) continue;
        int lo = 0;
        while (lo < m && out_word[lo] < word[i]) ++lo;                 /* std::map lower_bound */
        if (lo < m && out_word[lo] == word[i]) {
            if (weighting == 0 || weighting == 1) out_value[lo] += weight[i];    /* addWeight; addIfNotExist keeps the first */
        } else {
            memmove(out_word + lo + 1, out_word + lo, sizeof(int32_t) * (m - lo));
            memmove(out_value + lo + 1, out_value + lo, sizeof(double) * (m - lo));
            out_word[lo] = word[i]; out_value[lo] = weight[i]; ++m;
        }
    }
    const int must = scoring != 5;
    if ((weighting == 0 || weighting == 1) && m > 0 && !must) {
        const double nd = (double)m;
        for (int j = 0; j < m; ++j) out_value[j] /= nd;
    }
    if (must) {                                                         /* BowVector::normalize */
        double norm = 0.0;
        if (scoring != 1) for (int j = 0; j < m; ++j) norm += fabs(out_value[j]);
        else { for (int j = 0; j < m; ++j) norm += out_value[j] * out_value[j]; norm = sqrt(norm); }
        if (norm > 0.0) for (int j = 0; j < m; ++j) out_value[j] /= norm;
    }
    return m;
}

static int cmp_i64(const void* a, const void* b)
{
    const int64_t x = *(const int64_t*)a, y = *(const int64_t*)b;
    return x < y ? -1 : x > y;
}

/* MatchByDBoW, src/FeatureMatching.cpp:58-125, given the two transforms (:68-69).  The FeatureVector iteration
 * (:72-123) visits the node ids present in both frames in ascending order; inside a node the query features in
 * ascending index, each against the node's train features in ascending index (:82-102). */
int orc_match_dbow(const uint8_t* q, int nq, int qstride, const int32_t* qnode, const double* qweight,
                   const uint8_t* t, int nt, int tstride, const int32_t* tnode, const double* tweight,
                   int threshold, float nn_ratio, orc_dmatch* out)
{
    if (!(nq > 0 && nt > 0)) return 0;                                  /* :61-63 */
    int64_t* order = (int64_t*)malloc(sizeof(int64_t) * nq);          /* query features by (node id, index): std::map order */
    int m1 = 0;
    for (int i = 0; i < nq; ++i) if (qweight[i] > 0) order[m1++] = ((int64_t)qnode[i] << 32) | (int64_t)i;
    qsort(order, m1, sizeof(int64_t), cmp_i64);
    int n = 0;
    {
        for (int o = 0; o < m1; ++o) {
            const int i = (int)(order[o] & 0xFFFFFFFF), node = qnode[i];
            int bestDist1 = 256, bestDist2 = 256, bestIdx = -1, any = 0;  /* :85-87 */
            for (int j = 0; j < nt; ++j) {
                if (!(tweight[j] > 0) || tnode[j] != node) continue;
                any = 1;
                const int dist = orc_hamming256(q + (size_t)i * qstride, t + (size_t)j * tstride);
                if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx = j; }
                else if (dist < bestDist2) bestDist2 = dist;
            }
            if (!any) continue;                                         /* node absent from frame 2: :117-122 skip it */
            if (bestDist1 <= threshold) {                               /* :104 */
                if ((float)bestDist1 < nn_ratio * (float)bestDist2) {   /* :105 */
                    orc_dmatch m; m.queryIdx = i; m.trainIdx = bestIdx; m.imgIdx = -1; m.distance = (float)bestDist1;
                    out[n++] = m;
                }
            }
        }
    }
    free(order);
    return n;
}
