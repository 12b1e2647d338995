/* ssvio_oracle.h -- CPU restatement of SSVIO's ORB-extract + Hamming-match path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under ssvio_amd/ or include/ may include,
 * link or call this; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg do, and only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED: the reference (StarRealMan/SSVIO) ships no tests, fixtures or
 * golden vectors, cannot be built here (OpenCV/DBoW3 absent, F3)
 * and the per-pixel arithmetic lives in un-vendored OpenCV 3.x (pinned only as
 * find_package(OpenCV 3.1), CMakeLists.txt:18).  Every function below cites the
 * reference file:line it follows; the OpenCV primitives follow the published
 * OpenCV 3.x generic C++ algorithms as laid down in SURVEY.md Appendix A.  The
 * restatement is pinned only by hand-derived known-answer tests (tests/golden).
 */
#ifndef SSVIO_ORACLE_H
#define SSVIO_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y, size, angle, response; int32_t octave, class_id; } orc_keypoint; /* cv::KeyPoint, 28 B */
typedef struct { int32_t queryIdx, trainIdx, imgIdx; float distance; } orc_dmatch;            /* cv::DMatch, 16 B */

/* ---- primitives ---- */
int  orc_cvround(double v);                                                           /* A0 */
void orc_rgb2gray(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride); /* A1, src/Frame.cpp:7 */
void orc_resize_linear(const uint8_t* src, int sw, int sh, int sstride,
                       uint8_t* dst, int dw, int dh, int dstride);                    /* A2, src/Feature.cpp:813 */
void orc_border_reflect101(uint8_t* buf, int w, int h, int stride, int pad);          /* A7, src/Feature.cpp:815-821: fills the
                                                                                         pad-wide rim around the w x h ROI at (pad,pad) */
int  orc_fast9(const uint8_t* img, int w, int h, int stride, int threshold,
               int* xs, int* ys, int* scores, int cap);                               /* A3, src/Feature.cpp:872-878; returns count */
int  orc_fast_score_ref(const uint8_t* p, int stride, int threshold);                 /* cornerScore<16> at one pixel */
void orc_gaussian7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride); /* A4, src/Feature.cpp:548 */
float orc_fast_atan2(float y, float x);                                               /* A5, src/Feature.cpp:112 */
void orc_sincosf(float x, float* s, float* c);    /* canonical sinf/cosf: double-precision evaluation rounded once to float */
void orc_gauss_kernel_fixed(int k[7]);            /* the 8-bit fixed-point kernel derived from getGaussianKernel(7,2) */

/* ---- extractor (src/Feature.cpp:379-916) ---- */
typedef struct orc_extractor orc_extractor;
orc_extractor* orc_create(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th); /* :379-444 */
void orc_destroy(orc_extractor*);
/* tables built by the constructor */
const float* orc_scale_factors(orc_extractor*);     /* mvScaleFactor  */
const float* orc_inv_scale_factors(orc_extractor*); /* mvInvScaleFactor */
const float* orc_level_sigma2(orc_extractor*);
const float* orc_inv_level_sigma2(orc_extractor*);
const int*   orc_features_per_level(orc_extractor*);
const int*   orc_umax(orc_extractor*);              /* 16 entries */
/* Extract (:504-568).  Returns #keypoints (<= cap), -1 on bad geometry, 0 for an empty image. */
int  orc_extract(orc_extractor*, const uint8_t* gray, int w, int h, int stride,
                 orc_keypoint* kps, uint8_t* desc, int cap);
/* Frame::Frame (src/Frame.cpp:3-9): 3-channel interleaved -> gray -> Extract */
int  orc_extract_rgb(orc_extractor*, const uint8_t* rgb, int w, int h, int stride,
                     orc_keypoint* kps, uint8_t* desc, int cap);
/* intermediates of the LAST orc_extract call, for stage-level parity */
int  orc_level_size(orc_extractor*, int level, int* w, int* h);
const uint8_t* orc_level_roi(orc_extractor*, int level, int* stride);      /* un-blurred level, ROI origin; rim of 19 px exists around it */
const uint8_t* orc_level_blur(orc_extractor*, int level, int* stride);     /* blurred clone (NULL if level had no keypoints) */
int  orc_level_candidates(orc_extractor*, int level, const float** xs, const float** ys, const float** resp); /* pre-quadtree, coords rel. to minBorder */
int  orc_level_keypoints(orc_extractor*, int level, const orc_keypoint** kps); /* post-quadtree+orientation, level coordinates */
int  orc_level_tie_at_cut(orc_extractor*, int level);   /* 1 if the size-sorted split loop met an equal-size tie */
/* cell grid (:835-869) */
typedef struct { int nCols, nRows, wCell, hCell, minBX, minBY, maxBX, maxBY; } orc_grid;
int  orc_cell_grid(int level_w, int level_h, orc_grid* g);                 /* -1 if degenerate */
/* standalone quadtree (:570-798, :17-81); coords relative to (minX,minY); out = indices into the input */
int  orc_distribute_octtree(const float* xs, const float* ys, const float* resp, int n,
                            int minX, int maxX, int minY, int maxY, int N, int* out_idx, int cap, int* tie_flag);
/* IC_Angle (:84-113) + rBRIEF (:446-482) at one keypoint */
float orc_ic_angle(const uint8_t* img, int stride, int x, int y, int* m01, int* m10);
void  orc_orb_descriptor(const uint8_t* blurred, int stride, int x, int y, float angle_deg, uint8_t desc[32]);
void  orc_orb_descriptor_libm(const uint8_t* blurred, int stride, int x, int y, float angle_deg, uint8_t desc[32]); /* host libm cosf/sinf */

/* ---- matching (src/FeatureMatching.cpp) ---- */
int  orc_hamming256(const uint8_t* a, const uint8_t* b);                               /* :211-228 */
/* raw 2-NN per query (:85-102): idx (-1 if none), best, second; all trains, ascending */
void orc_knn2_raw(const uint8_t* q, int nq, int qstride, const uint8_t* t, int nt, int tstride,
                  int32_t* idx, int32_t* d1, int32_t* d2);
/* BFMatcher(NORM_HAMMING).match (App. A6): one (idx, dist) per query, lowest index on ties */
void orc_bf_nn(const uint8_t* q, int nq, int qstride, const uint8_t* t, int nt, int tstride,
               int32_t* idx, int32_t* dist);
int  orc_match_bruteforce(const uint8_t* q, int nq, int qstride, const uint8_t* t, int nt, int tstride,
                          orc_dmatch* out);                                            /* :20-55; returns #matches */
int  orc_match_knn2_ratio(const uint8_t* q, int nq, int qstride, const uint8_t* t, int nt, int tstride,
                          int threshold, float nn_ratio, orc_dmatch* out);             /* :85-112 over all rows */
int  orc_match_crosscheck_cv(const uint8_t* q, int nq, int qstride, const uint8_t* t, int nt, int tstride, orc_dmatch* out);
int  orc_match_crosscheck(const uint8_t* q, int nq, int qstride, const uint8_t* t, int nt, int tstride,
                          orc_dmatch* out);                                            /* App. A6 mutual-NN */
/* MatchByProject (:128-189): project p3d with R, t, K; 2-NN among the train keypoints inside the +-radian window */
int  orc_match_project(const double* p3d, const uint8_t* q, int n1, int qstride,
                       const double* p2d, const uint8_t* t, int n2, int tstride,
                       double radian, const double* K, const double* R, const double* tr,
                       int threshold, float nn_ratio, orc_dmatch* out);

/* ---- MatchByDBoW (src/FeatureMatching.cpp:58-125, :192-208; DBoW3's transform restated, bow_oracle.c) ---- */
typedef struct orc_vocab orc_vocab;
/* node table: node 0 = root, parent[n] < n, children in ascending id, desc[nnodes][32], word[n] >= 0 marks a leaf.
 * scoring: DBoW3's ScoringType 0..5 (L1 L2 CHI_SQUARE KL BHATTACHARYYA DOT_PRODUCT); weighting: 0 TF_IDF, 1 TF, 2 IDF, 3 BINARY (DBoW3's enums) */
orc_vocab* orc_vocab_create(int k, int L, int nnodes, const int32_t* parent, const uint8_t* desc,
                            const double* weight, const int32_t* word, int scoring, int weighting);
void orc_vocab_destroy(orc_vocab*);
void orc_bow_transform(const orc_vocab* v, const uint8_t* desc, int n, int stride, int levelsup,
                       int32_t* word, int32_t* node, double* weight);
int  orc_bow_vector(const int32_t* word, const double* weight, int n, int scoring, int weighting,
                    int32_t* out_word, double* out_value);
int  orc_match_dbow(const uint8_t* q, int nq, int qstride, const int32_t* qnode, const double* qweight,
                    const uint8_t* t, int nt, int tstride, const int32_t* tnode, const double* tweight,
                    int threshold, float nn_ratio, orc_dmatch* out);

/* ---- behind the matcher (frame_oracle.c) ---- */
/* cam = {InnerCx, InnerCy, InnerFx, InnerFy, DepthScale/65535} (include/Init.h:14-20) */
void orc_get_3d_point(const orc_keypoint* kps, int index, const uint16_t* depth, int depth_stride_px,
                      const float cam[5], float out[3]);                                  /* src/Frame.cpp:93-108 */
void orc_assemble_correspondences(const orc_dmatch* matches, int nm,
                                  const orc_keypoint* last_kps, const uint16_t* last_depth,
                                  const orc_keypoint* cur_kps, const uint16_t* cur_depth, int depth_stride_px,
                                  const float cam[5],
                                  float* last3d, float* cur2d, float* cur3d, orc_dmatch* pre_good, int* n_pre,
                                  float* edge_last3d, float* edge_cur3d, orc_dmatch* final_good, int* n_final); /* src/Odometry.cpp:44-64,94-108 */
int  orc_track_map_points(const orc_dmatch* last, int nlast, const orc_dmatch* cur, int ncur, orc_dmatch* out); /* src/Map.cpp:100-123 */
void orc_in_match_vec(const orc_dmatch* vec, int n, int nkp, int32_t* out);              /* src/Map.cpp:151-162 */
void orc_manage_map_points(const orc_keypoint* kps, int nkp, const uint16_t* depth, int depth_stride_px, const float cam[5],
                           const float world_trans[16], const orc_dmatch* last_match_vec, int nm,
                           const int32_t* obs, int nobs, int32_t* match_num, int32_t* map_point, float* world); /* src/Map.cpp:125-149, src/Frame.cpp:138-147 */
void orc_rgb_cloud(const uint16_t* depth, int depth_stride_px, const uint8_t* bgr, int bgr_stride, int w, int h,
                   float cx, float cy, float fx, float fy, double depth_scale, uint8_t* out);  /* driver/Xtion_Driver.cpp:138-162, :219-229 */
int orc_voxel_fuse(const uint8_t* final_cloud, int n_final, const uint8_t* frame_cloud, int n_frame,
                   const float* this_trans, float voxel_size, uint8_t* out);                    /* src/Map.cpp:164-197, one key frame */
#ifdef __cplusplus
}
#endif
#endif
