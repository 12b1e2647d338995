/* ssvio_gpu.h -- C ABI of the MI355X (gfx950) ORB-extract + Hamming-match library.
 *
 * This is the drop-in boundary for SSVIO's front-end hot path.  Every entry point
 * names the reference interface it replaces (file:line under StarRealMan/SSVIO).
 * Plain pointers and sizes only; no C++/torch types.  The C++ adapter classes in
 * include/ssvio/ (ORBextractor, FeatureMatching) sit on top of exactly these
 * symbols, see INTEGRATION.md.
 *
 * Conventions
 *   - every function returns SSVIO_OK (0) or a negative SSVIO_ERR_* code and never
 *     throws; ssvio_last_error() gives the thread-local message of the last failure.
 *   - a handle owns one HIP stream and all its device memory; one caller at a time
 *     per handle (the reference objects are not re-entrant either: SURVEY.md §8b).
 *   - "host" entry points take host pointers and are synchronous.  "_dev" entry
 *     points take device pointers, enqueue on the handle's stream and return
 *     without synchronising unless stated.
 *   - there is no CPU fallback: without a usable GPU every create() fails with
 *     SSVIO_ERR_NOGPU.
 */
#ifndef SSVIO_GPU_H
#define SSVIO_GPU_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SSVIO_OK             0
#define SSVIO_ERR_INVALID   (-1)   /* bad argument */
#define SSVIO_ERR_HIP       (-2)   /* HIP runtime error (message has the HIP string) */
#define SSVIO_ERR_CAPACITY  (-3)   /* caller's output buffer too small */
#define SSVIO_ERR_GEOMETRY  (-4)   /* image too small / aspect ratio for which the reference divides by zero */
#define SSVIO_ERR_NOGPU     (-5)   /* no gfx950 device visible */

#define SSVIO_MAX_LEVELS 16

/* cv::KeyPoint memory layout (28 bytes) -- what Extract() fills, include/Feature.h:42 */
typedef struct ssvio_keypoint {
    float x, y;        /* pt, level-0 coordinates (src/Feature.cpp:557-563) */
    float size;        /* (int)(15*scale[level]) (src/Feature.cpp:900,908) */
    float angle;       /* degrees [0,360) (src/Feature.cpp:112) */
    float response;    /* FAST score (src/Feature.cpp:872) */
    int32_t octave;    /* pyramid level */
    int32_t class_id;  /* -1 */
} ssvio_keypoint;

/* cv::DMatch memory layout (16 bytes) -- what MatchBy*() fill, include/FeatureMatching.h:24 */
typedef struct ssvio_dmatch {
    int32_t queryIdx, trainIdx, imgIdx;
    float distance;
} ssvio_dmatch;

/* ORBextractor constructor arguments, include/Feature.h:38 / src/Odometry.cpp:8-13 */
typedef struct ssvio_orb_params {
    int32_t nfeatures;     /* FeatureNum  (config/default_conf.yaml:16) */
    float   scale_factor;  /* ScaleFactor (:17) */
    int32_t nlevels;       /* LevelNum    (:18), 1..SSVIO_MAX_LEVELS */
    int32_t ini_th_fast;   /* IniThFAST   (:19) */
    int32_t min_th_fast;   /* MinThFAST   (:20) */
} ssvio_orb_params;

typedef struct ssvio_orb ssvio_orb;
typedef struct ssvio_matcher ssvio_matcher;

/* ---- library ------------------------------------------------------------ */
const char* ssvio_last_error(void);
int  ssvio_version(void);
int  ssvio_device_count(int* count);

/* ---- extractor: replaces class ORBextractor (include/Feature.h:32-103) ---- */
/* ORBextractor::ORBextractor, src/Feature.cpp:379-444 */
int  ssvio_orb_create(const ssvio_orb_params* params, int device, ssvio_orb** out);
void ssvio_orb_destroy(ssvio_orb* h);
/* GetScaleFactors / GetInvScaleFactor / GetLevelSigmaSquares / GetInvLevelSigmaSquares
 * (include/Feature.h:57-75) and mnFeaturesPerLevel; each array has nlevels entries, NULL to skip */
int  ssvio_orb_get_tables(ssvio_orb* h, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                          int32_t* features_per_level);
/* upper bound on keypoints per frame (sum of quotas + quadtree overshoot); size kps/desc with it */
int  ssvio_orb_max_keypoints(ssvio_orb* h);
/* level geometry for a w x h input: sizes and FAST cell grid (src/Feature.cpp:805, :835-849) */
int  ssvio_orb_level_geometry(ssvio_orb* h, int width, int height, int level,
                              int32_t* lw, int32_t* lh, int32_t* ncols, int32_t* nrows, int32_t* wcell, int32_t* hcell);

/* ORBextractor::Extract(image, mask, keypoints, descriptors), src/Feature.cpp:504-568.
 * gray: 8UC1 host image.  kps[cap], desc[cap*32] caller-owned; *n = number written.
 * Empty image (NULL / w<=0 / h<=0) -> *n = 0, SSVIO_OK (silent return, :507-508). */
int  ssvio_orb_extract(ssvio_orb* h, const uint8_t* gray, int width, int height, int stride,
                       ssvio_keypoint* kps, uint8_t* desc, int cap, int* n);
/* Frame::Frame front half, src/Frame.cpp:3-9: cvtColor(CV_RGB2GRAY) then Extract.  rgb: 8UC3 interleaved */
int  ssvio_orb_extract_rgb(ssvio_orb* h, const uint8_t* rgb, int width, int height, int stride,
                           ssvio_keypoint* kps, uint8_t* desc, int cap, int* n);
/* Batched Extract over `batch` equally sized frames (frame f at imgs + f*frame_stride).
 * channels = 1 (gray) or 3 (RGB2GRAY first).  kps[batch*cap], desc[batch*cap*32], n[batch].
 * Results are identical to `batch` separate ssvio_orb_extract calls. */
int  ssvio_orb_extract_batch(ssvio_orb* h, const uint8_t* imgs, int batch, int channels,
                             int width, int height, int stride, size_t frame_stride,
                             ssvio_keypoint* kps, uint8_t* desc, int cap, int* n);
/* Same with device-resident frames; results stay in HBM inside the handle:
 * d_kps[batch][cap'], d_desc[batch][cap'][32], d_n[batch] with cap' = ssvio_orb_max_keypoints().
 * The call returns after the work is complete; results valid until the next call on the handle. */
int  ssvio_orb_extract_batch_dev(ssvio_orb* h, const void* d_imgs, int batch, int channels,
                                 int width, int height, int stride, size_t frame_stride);
/* Pipelined form ("sustained fps (pipelined)"): only enqueues the work on the
 * handle's stream and returns.  The device results may be consumed by later work on the same stream
 * (bind the matcher to it: ssvio_orb_get_stream -> ssvio_matcher_set_stream); the host-visible counts
 * (ssvio_orb_last_counts / ssvio_orb_fetch_results) wait for completion themselves, ssvio_orb_sync
 * does so explicitly.  Not available with the host quadtree (SSVIO_QUADTREE=host), which waits anyway. */
int  ssvio_orb_extract_batch_dev_async(ssvio_orb* h, const void* d_imgs, int batch, int channels,
                                       int width, int height, int stride, size_t frame_stride);
int  ssvio_orb_sync(ssvio_orb* h);
/* the hipStream_t the handle enqueues on (its own unless ssvio_orb_set_stream changed it) */
int  ssvio_orb_get_stream(ssvio_orb* h, void** hip_stream);
int  ssvio_orb_device_results(ssvio_orb* h, const void** d_kps, const void** d_desc, const int32_t** d_n,
                              int* cap_per_frame);
/* copy the device-resident results of the last *_dev call to the host */
int  ssvio_orb_fetch_results(ssvio_orb* h, int batch, ssvio_keypoint* kps, uint8_t* desc, int cap, int* n);
/* host copy of the counts of the last call (no device access) */
int  ssvio_orb_last_counts(ssvio_orb* h, int batch, int* n);

/* public member mvImagePyramid (include/Feature.h:45): level image of frame `frame` of the last
 * call, WITH the 19-px BORDER_REFLECT_101 rim (src/Feature.cpp:806-821).  dst has (lh+38) rows of
 * dst_stride >= lw+38 bytes.  The rim is never read by Extract, so the GPU path does not
 * keep it; it is synthesised here on request. */
int  ssvio_orb_pyramid_level(ssvio_orb* h, int frame, int level, uint8_t* dst, int dst_stride, int* lw, int* lh);

/* stage-level taps for parity tests (last call, frame `frame`) */
int  ssvio_orb_debug_level_image(ssvio_orb* h, int frame, int level, int blurred, uint8_t* dst, int dst_stride);
/* candidates before the quadtree: x,y relative to (minBorderX,minBorderY), response; returns count via *n */
int  ssvio_orb_debug_candidates(ssvio_orb* h, int frame, int level, int32_t* xs, int32_t* ys, int32_t* resp, int cap, int* n);

/* DistributeOctTree on its own (src/Feature.cpp:570-798): host-side stage of this version.
 * xs/ys relative to (minX,minY); out_idx receives indices into the input, in the reference's list order. */
int  ssvio_distribute_octtree(const float* xs, const float* ys, const float* resp, int n,
                              int minX, int maxX, int minY, int maxY, int N, int32_t* out_idx, int cap, int* nout);

/* per-kernel GPU timing (hipEvent pairs on the handle's stream). enable!=0 starts/clears collection.
 * ssvio_orb_kernel_times: fills up to cap entries; names are static strings. */
int  ssvio_orb_set_profiling(ssvio_orb* h, int enable);
int  ssvio_orb_kernel_times(ssvio_orb* h, const char** names, double* total_ms, int32_t* launches, int cap, int* count);
/* bracket only the kernel of this name (NULL or "" = every kernel): two events per call instead of two per
 * kernel, for timing one kernel inside a throughput run without slowing the stream down */
int  ssvio_orb_profile_only(ssvio_orb* h, const char* kernel);
/* bind the handle to a caller-owned hipStream_t (NULL restores the handle's own stream) */
int  ssvio_orb_set_stream(ssvio_orb* h, void* hip_stream);

/* ---- matcher: replaces class FeatureMatching (include/FeatureMatching.h:13-54) ---- */
/* FeatureMatching::FeatureMatching reads Threshold / NNRatio (src/FeatureMatching.cpp:10-11).
 * The DBoW3 vocabulary it also loads belongs to MatchByDBoW, which is out of scope. */
int  ssvio_matcher_create(int threshold, float nn_ratio, int device, ssvio_matcher** out);
void ssvio_matcher_destroy(ssvio_matcher* m);
int  ssvio_matcher_set_stream(ssvio_matcher* m, void* hip_stream);
int  ssvio_matcher_set_profiling(ssvio_matcher* m, int enable);
int  ssvio_matcher_kernel_times(ssvio_matcher* m, const char** names, double* total_ms, int32_t* launches, int cap, int* count);
int  ssvio_matcher_profile_only(ssvio_matcher* m, const char* kernel);

/* FeatureMatching::MatchByBruteForce(desp1, desp2, matches), src/FeatureMatching.cpp:20-55:
 * BFMatcher(NORM_HAMMING).match + keep dist < max(2*min_dist, 30).  Rows are 32 bytes, `stride`
 * bytes apart (non-continuous Mats).  out[cap >= nq]; *n = 0 when either set is empty (:25-27). */
int  ssvio_match_bruteforce(ssvio_matcher* m, const uint8_t* q, int nq, int qstride,
                            const uint8_t* t, int nt, int tstride, ssvio_dmatch* out, int cap, int* n);
/* 2-NN + Threshold/NNRatio acceptance of src/FeatureMatching.cpp:85-112 over all rows of both sets
 * (the brute-force form of the MatchByDBoW inner loop; BASELINE config 3). imgIdx = -1 like DMatch(). */
int  ssvio_match_knn2_ratio(ssvio_matcher* m, const uint8_t* q, int nq, int qstride,
                            const uint8_t* t, int nt, int tstride, ssvio_dmatch* out, int cap, int* n);
/* BFMatcher(NORM_HAMMING, crossCheck=true) as mutual nearest neighbours (no reference code) */
int  ssvio_match_crosscheck(ssvio_matcher* m, const uint8_t* q, int nq, int qstride,
                            const uint8_t* t, int nt, int tstride, ssvio_dmatch* out, int cap, int* n);
/* raw 2-NN of src/FeatureMatching.cpp:85-102 / ComputeMatchingScore :211-228:
 * idx[i] (-1 if best distance is 256), d1[i], d2[i] */
int  ssvio_match_knn2_raw(ssvio_matcher* m, const uint8_t* q, int nq, int qstride,
                          const uint8_t* t, int nt, int tstride, int32_t* idx, int32_t* d1, int32_t* d2);

/* FeatureMatching::MatchByProject(vp3d1, desp1, vp2d2, desp2, radian, K, R, t, matches), include/FeatureMatching.h:30-32,
 * src/FeatureMatching.cpp:128-189 (the DBoW3-free half): reference point i (p3d[i][3], descriptor row i
 * of q) is projected with R, t, K (row-major 3x3 / 3 doubles, the reference's double arithmetic incl. its bounds test
 * against the principal point, :141); among the train keypoints p2d[j][2] strictly inside the +-radian window the best /
 * second-best Hamming distances decide: kept iff best < Threshold and best < NNRatio * second (:177-178).
 * out[cap >= nq] receives DMatch{queryIdx = i, trainIdx, imgIdx = -1, distance} in ascending i. */
int  ssvio_match_project(ssvio_matcher* m, const double* p3d, const uint8_t* q, int nq, int qstride,
                         const double* p2d, const uint8_t* t, int nt, int tstride, double radian,
                         const double* K, const double* R, const double* tr, ssvio_dmatch* out, int cap, int* n);

/* Batched matching over descriptor sets resident in HBM (BASELINE configs 2, 4, 5).
 * d_desc: set s occupies rows [s*set_stride_rows, +counts[s]) of a contiguous [rows][32] byte array.
 * d_counts: int32 per set (device).  pairs: host array of npairs (query_set, train_set) int32 pairs.
 * mode: 0 = bruteforce filter, 1 = knn2+ratio, 2 = crosscheck.
 * Output (device, owned by the handle until the next call): d_matches[npairs][max_rows] ssvio_dmatch
 * and d_nmatches[npairs].  Enqueued on the stream; no synchronisation. */
#define SSVIO_MODE_BRUTEFORCE 0
#define SSVIO_MODE_KNN2_RATIO 1
#define SSVIO_MODE_CROSSCHECK 2
int  ssvio_match_batch_dev(ssvio_matcher* m, const void* d_desc, int set_stride_rows, int max_rows,
                           const int32_t* d_counts, const int32_t* pairs, int npairs, int mode,
                           const void** d_matches, const int32_t** d_nmatches);
/* blocks until the matcher's stream is idle */
int  ssvio_matcher_sync(ssvio_matcher* m);
/* copy batched results to the host: out[npairs][max_rows], n[npairs] */
int  ssvio_match_batch_fetch(ssvio_matcher* m, int npairs, int max_rows, ssvio_dmatch* out, int32_t* n);

#ifdef __cplusplus
}
#endif
#endif
