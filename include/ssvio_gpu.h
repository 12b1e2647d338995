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
/* test hook: the next n device allocations by which a matcher handle grows its buffers fail as if HIP were out of memory
 * (the call returns SSVIO_ERR_HIP; the handle must stay usable).  n = 0 switches it off. */
void ssvio_test_fail_next_allocs(int n);

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

/* The same, produced WITH the call instead of fetched after it (what the C++ adapter uses to fill mvImagePyramid on every Extract,
 * as the reference does, src/Feature.cpp:514, :800-824): once enabled, every single-frame host call (ssvio_orb_extract[_rgb]) also
 * stores the rimmed levels of its frame into pinned host memory owned by the handle -- a kernel writes them across the bus beside
 * the FAST / quadtree stages, no copy commands -- and ssvio_orb_exported_pyramid_level hands out level `level`: *base = the rimmed
 * image ((lh + 38) rows of *stride bytes; the level itself starts at base + 19 * stride + 19).  Valid until the handle's next call. */
int  ssvio_orb_set_export_pyramid(ssvio_orb* h, int enable);
int  ssvio_orb_exported_pyramid_level(ssvio_orb* h, int level, const uint8_t** base, int* stride, int* lw, int* lh);

/* Replay the extractor's launch sequence (pyramid, FAST, blur beside the quadtree, describe: ~12 launches, one fork / join) as a hipGraph,
 * captured once per (batch size, buffer generation) -- what SSVIO_GRAPH=1 does for every handle of a process, here per handle.  Same
 * results.  No counterpart in the reference (src/Feature.cpp:503-568 is one synchronous call); for callers that want one submission per
 * call.  Off by default: on this stack the plain launches are 2-3 % ahead with device-resident frames and level with the replay when the
 * same host thread streams the frames up from pinned memory (163-164 k against 162 k frames/s, DESIGN.md sections 4 and 6). */
int  ssvio_orb_set_graph(ssvio_orb* h, int enable);

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
 * The DBoW3 vocabulary it also loads (:8-9) is a separate object here: ssvio_vocab_create below. */
int  ssvio_matcher_create(int threshold, float nn_ratio, int device, ssvio_matcher** out);
void ssvio_matcher_destroy(ssvio_matcher* m);
/* Runs the matcher's work on a caller's stream (NULL: back to its own).  The stream is borrowed: it has to stay alive while
 * calls are made on the handle; ssvio_matcher_destroy itself does not touch a borrowed stream (it waits for the device),
 * so the stream's owner -- often the extractor, ssvio_orb_get_stream -- may be destroyed before the matcher. */
int  ssvio_matcher_set_stream(ssvio_matcher* m, void* hip_stream);
/* Which kernel computes ComputeMatchingScore + the best / second-best search (src/FeatureMatching.cpp:211-228, :85-102) for the
 * shapes beyond one small pair (the batched entry points; single pairs above 2048 x 4096 rows).  Both give identical results:
 *   SSVIO_KERNEL_VALU (default): k_hamming_knn2, v_xor + v_bcnt per dword, the kernel north_star describes;
 *   SSVIO_KERNEL_MFMA: k_hamming_knn2_mfma, the same distance as an exact int8 contraction on the matrix cores
 *     (v_mfma_i32_32x32x32_i8; hamming = |a| - sum a_k t_k with the bits as +-1 / 0-1 integers), the accumulator doubling as the
 *     (distance, train row) sort key.
 *   SSVIO_KERNEL_MFMA_ALL: the matrix-core kernel for every shape, small single pairs included (those otherwise keep their own
 *     one-launch kernel, k_hamming_knn2_small, which is the faster one there); for tests and measurements.
 * The environment variable SSVIO_MATCH_KERNEL=mfma (mfma_all) makes that the default of new handles. */
#define SSVIO_KERNEL_VALU 0
#define SSVIO_KERNEL_MFMA 1
#define SSVIO_KERNEL_MFMA_ALL 2
int  ssvio_matcher_set_kernel(ssvio_matcher* m, int kernel);
int  ssvio_matcher_get_kernel(ssvio_matcher* m, int* kernel);
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
/* BFMatcher(NORM_HAMMING, crossCheck=true) the way OpenCV 3.x itself computes it ("OpenCV-compat"; recalled, not
 * verifiable here): only the reverse search is run -- every train row votes for its nearest query row (lowest query index on
 * ties), a query keeps the closest train among its voters (lowest train index on ties) and is emitted iff it got a vote.  This
 * differs from the mutual nearest neighbour above when a query's own nearest train prefers another query.  imgIdx = 0. */
int  ssvio_match_crosscheck_cv(ssvio_matcher* m, const uint8_t* q, int nq, int qstride,
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
 * mode: 0 = bruteforce filter, 1 = knn2+ratio, 2 = crosscheck (mutual NN), 4 = crosscheck as OpenCV computes it.
 * Output (device, owned by the handle until the next call): d_matches[npairs][max_rows] ssvio_dmatch
 * and d_nmatches[npairs].  Enqueued on the stream; no synchronisation. */
#define SSVIO_MODE_BRUTEFORCE 0
#define SSVIO_MODE_KNN2_RATIO 1
#define SSVIO_MODE_CROSSCHECK 2
#define SSVIO_MODE_CROSSCHECK_CV 4  /* the OpenCV-compat cross-check of ssvio_match_crosscheck_cv (3 is internal) */
int  ssvio_match_batch_dev(ssvio_matcher* m, const void* d_desc, int set_stride_rows, int max_rows,
                           const int32_t* d_counts, const int32_t* pairs, int npairs, int mode,
                           const void** d_matches, const int32_t** d_nmatches);
/* Loop-candidate matching (BASELINE config 5) keeps only a fixed-size summary per keyframe pair: the reference has
 * no consumer for it (src/Loop.cpp:1 is empty), each record is what MatchByBruteForce's `matches` (src/FeatureMatching.cpp:47-54)
 * reduces to.  rec[k] for pair k of the LAST ssvio_match_batch_dev / ssvio_match_dbow_batch_dev call on this handle, reduced on the
 * device into d_records[npairs] (caller's device memory, 16-byte records); enqueued on the stream, no synchronisation.
 * id_mode SSVIO_PAIR_ID_SEQUENTIAL: pair = id_base + k; SSVIO_PAIR_ID_TRIANGULAR: pair = hi*(hi-1)/2 + lo of the pair's two set
 * indices (the numbering of unordered pairs). */
typedef struct ssvio_pair_summary { int32_t pair; int32_t n_matches; int64_t sum_dist; } ssvio_pair_summary;
#define SSVIO_PAIR_ID_SEQUENTIAL 0
#define SSVIO_PAIR_ID_TRIANGULAR 1
int  ssvio_match_batch_summary_dev(ssvio_matcher* m, int id_mode, int id_base, void* d_records);
/* The rank's share of the loop-candidate walk (BASELINE config 5; what a C++ `Loop` -- empty in the reference, include/Loop.h:1-4,
 * src/Loop.cpp:1 -- would call): all unordered pairs (i, j), j < i, of nsets key frames, every pair-block matched as
 * MatchByBruteForce(desc_i, desc_j) does (src/FeatureMatching.cpp:20-55; `mode` may select one of the other filters) and reduced on the
 * device to one ssvio_pair_summary with the pair's triangular id i*(i-1)/2 + j.  The triangular block matrix is cut into super-blocks
 * of sb x sb key frames which are dealt to `world` ranks, largest first onto the least loaded rank (ties: lowest rank) -- a pure
 * function of (nsets, sb, world), so every rank computes the same deal; ssvio_loop_candidates_plan returns this rank's number of
 * records, the largest number any rank gets (the fixed record size of an all_gather) and this rank's number of super-blocks.
 * ssvio_loop_candidates_dev walks this rank's super-blocks -- pair list built on the device, one batched matcher pass and one
 * k_pair_summary launch per super-block, matcher scratch bounded by sb*sb pair-blocks -- and writes its records, in walk order, to
 * d_records[0 .. *n_records) (device memory, records_cap entries).  Everything is enqueued on the handle's stream; nothing waits and
 * nothing returns to the host.  Descriptor layout as ssvio_match_batch_dev.  The collective (one all_gather of the records: RCCL
 * over xGMI) stays with the caller -- ssvio_amd/dist.py:loop_candidates is that caller. */
int  ssvio_loop_candidates_plan(int nsets, int sb, int rank, int world, int64_t* n_mine, int64_t* n_max, int* n_tiles);
int  ssvio_loop_candidates_dev(ssvio_matcher* m, const void* d_desc, int set_stride_rows, int max_rows, const int32_t* d_counts,
                               int nsets, int sb, int rank, int world, int mode, void* d_records, int64_t records_cap, int64_t* n_records);
/* The same batched matcher for callers that cannot allocate.  No counterpart in the reference: it exists for callers under HIP
 * graph capture, tight serving loops and harnesses that forbid hipMalloc.  The call allocates nothing (no hipMalloc / hipFree /
 * pinned memory / events / copy commands): pair list, inputs and outputs are the caller's DEVICE memory -- d_pairs[npairs][2] int32,
 * d_matches[npairs][max_rows] ssvio_dmatch, d_nmatches[npairs] -- and the kernels' partial 2-NN lists live in a fixed device array of
 * SSVIO_NOALLOC_SCRATCH_BYTES per GPU (resolved once per device in ssvio_matcher_create).  Needs npairs * max_rows * 8 bytes of it
 * (twice that for the cross-check); beyond that the call returns SSVIO_ERR_CAPACITY and launches nothing.  All calls on one GPU
 * share the array: calls that overlap on one GPU (two handles, two streams) race -- serialise them.  Results are identical to
 * ssvio_match_batch_dev's.  Enqueued on the handle's stream; no synchronisation; kernel profiling does not cover it. */
#define SSVIO_NOALLOC_SCRATCH_BYTES (32u << 20)
int  ssvio_match_batch_dev_noalloc(ssvio_matcher* m, const void* d_desc, int set_stride_rows, int max_rows,
                                   const int32_t* d_counts, const int32_t* d_pairs, int npairs, int mode,
                                   void* d_matches, int32_t* d_nmatches);
/* blocks until the matcher's stream is idle */
int  ssvio_matcher_sync(ssvio_matcher* m);
/* copy batched results to the host: out[npairs][max_rows], n[npairs] */
int  ssvio_match_batch_fetch(ssvio_matcher* m, int npairs, int max_rows, ssvio_dmatch* out, int32_t* n);


/* ---- MatchByDBoW ------------------------------------------------------------------------
 * FeatureMatching::FeatureMatching loads a DBoW3::Vocabulary (src/FeatureMatching.cpp:8-9); DBoW3 is an un-vendored
 * third-party library and the vocabulary blob is missing from the reference tree (.MISSING_LARGE_BLOBS), so the
 * vocabulary crosses this boundary as a plain node table: node 0 is the root, parent[n] < n, the children of a node
 * are its nodes in ascending id (the order DBoW3 creates, saves and loads them in), desc[nnodes][32] (row 0 unused),
 * weight[nnodes], word[nnodes] = word id of a leaf (ignored for inner nodes).  k, L: branching factor and depth
 * (DBoW3's m_k, m_L).  scoring: DBoW3's ScoringType 0..5 (L1_NORM, L2_NORM, CHI_SQUARE, KL, BHATTACHARYYA, DOT_PRODUCT);
 * weighting: 0 TF_IDF, 1 TF, 2 IDF, 3 BINARY (DBoW3's enums; they only shape the BowVector the host assembles, not the matching). */
typedef struct ssvio_vocab ssvio_vocab;
int  ssvio_vocab_create(int device, int k, int L, int nnodes, const int32_t* parent, const uint8_t* desc, const double* weight,
                        const int32_t* word, int scoring, int weighting, ssvio_vocab** out);
void ssvio_vocab_destroy(ssvio_vocab* v);
int  ssvio_vocab_info(const ssvio_vocab* v, int* k, int* L, int* nnodes, int* scoring, int* weighting);
/* FeatureMatching::ComputeBoWVector (src/FeatureMatching.cpp:192-208) = DBoW3 Vocabulary::transform(features, bowVec,
 * featVec, levelsup = 4), per feature: word[i], weight[i] of the leaf reached and node[i], the node passed at level
 * L - levelsup (0 when that level is <= 0 or is never reached).  The BowVector (word -> summed weight, normalised) and
 * the FeatureVector (node -> ascending feature indices) follow from these on the host; features with weight <= 0 enter
 * neither. */
int  ssvio_bow_transform(ssvio_matcher* m, const ssvio_vocab* v, const uint8_t* desc, int n, int stride, int levelsup,
                         int32_t* word, int32_t* node, double* weight);
/* FeatureMatching::MatchByDBoW(desp1, desp2, matches), include/FeatureMatching.h:27, src/FeatureMatching.cpp:58-125: both
 * sets are transformed (levelsup = 4 in the reference, :206); for every vocabulary node present in both feature vectors,
 * in ascending node id, every query feature of the node (ascending index) takes its best / second-best Hamming distance
 * over the node's train features and is kept iff best <= Threshold and best < NNRatio * second (:104-105).
 * out[cap >= nq]: DMatch{queryIdx, trainIdx, imgIdx = -1, distance} in that order.  *n = 0 when either set is empty
 * (:61-63; the reference then leaves `matches` untouched -- the adapter's business). */
int  ssvio_match_dbow(ssvio_matcher* m, const ssvio_vocab* v, const uint8_t* q, int nq, int qstride,
                      const uint8_t* t, int nt, int tstride, int levelsup, ssvio_dmatch* out, int cap, int* n);
/* per-feature transform results of the last ssvio_match_dbow / ssvio_bow_transform call (mBowVec1_/mFeatVec1_ for side 0,
 * mBowVec2_/mFeatVec2_ for side 1, include/FeatureMatching.h:46-49); *n = rows of that side */
int  ssvio_bow_last_transform(ssvio_matcher* m, int side, int32_t* word, int32_t* node, double* weight, int cap, int* n);
/* batched, device-resident form (same layout as ssvio_match_batch_dev; nsets descriptor sets, each transformed once) */
int  ssvio_match_dbow_batch_dev(ssvio_matcher* m, const ssvio_vocab* v, const void* d_desc, int set_stride_rows, int max_rows,
                                const int32_t* d_counts, int nsets, const int32_t* pairs, int npairs, int levelsup,
                                const void** d_matches, const int32_t** d_nmatches);

/* ---- behind the matcher --------------------------------------------------------------- */
/* Frame's static camera parameters, include/Frame.h:46-52, include/Init.h:14-20.  depth_scale is the already divided
 * value (yaml DepthScale / 65535.0 narrowed to float, Init.h:18); 1/fx and 1/fy are formed in float like Init.h:19-20. */
typedef struct ssvio_camera { float cx, cy, fx, fy, depth_scale; } ssvio_camera;
/* Frame::Get3DPoint(index), src/Frame.cpp:93-108, for keypoints 0..n-1: out[n][3] floats.  depth: 16U image, w x h,
 * stride_px pixels per row; the pixel is (int)pt.y, (int)pt.x like Mat::at<ushort>(float, float). */
int  ssvio_get_3d_points(ssvio_matcher* m, const ssvio_keypoint* kps, int n, const uint16_t* depth, int w, int h, int stride_px,
                         const ssvio_camera* cam, float* out);
/* The two filter loops of Odometry::OptimizeTransform, src/Odometry.cpp:44-64 and :94-108, over matches[nm]
 * (trainIdx -> last frame, queryIdx -> current frame).  First lists (matches whose last-frame point has
 * z >= depth_scale, in match order): last3d[.][3], cur2d[.][2], cur3d[.][3], pre_good[.] -- the solvePnPRansac inputs.
 * Second lists (of those, the ones whose current-frame point has z >= depth_scale): edge_last3d, edge_cur3d (the
 * AddCVMeasure pairs) and final_good (_final_good_match).  Every output array holds nm entries. */
int  ssvio_assemble_correspondences(ssvio_matcher* m, const ssvio_dmatch* matches, int nm,
                                    const ssvio_keypoint* last_kps, int nlast, const uint16_t* last_depth,
                                    const ssvio_keypoint* cur_kps, int ncur, const uint16_t* cur_depth,
                                    int w, int h, int stride_px, const ssvio_camera* cam,
                                    float* last3d, float* cur2d, float* cur3d, ssvio_dmatch* pre_good, int* n_pre,
                                    float* edge_last3d, float* edge_cur3d, ssvio_dmatch* final_good, int* n_final);
/* device-resident, batched: chains behind ssvio_match_*_batch_dev and ssvio_orb_extract_batch_dev without a host
 * round trip.  pairs[npairs][2] = (current set, last set) as given to the matcher; d_kps / d_kpcounts as returned by
 * ssvio_orb_device_results (kp_stride entries per set); d_depth: one w x h 16U image per set, depth_frame_stride pixels
 * apart.  Outputs stay in HBM inside the handle, [pair][max_rows] each; d_counts[pair][2] = {n_pre, n_final}. */
typedef struct ssvio_correspondences_dev {
    const float* last3d; const float* cur2d; const float* cur3d; const ssvio_dmatch* pre_good;
    const float* edge_last3d; const float* edge_cur3d; const ssvio_dmatch* final_good;
    const int32_t* counts;
} ssvio_correspondences_dev;
int  ssvio_assemble_batch_dev(ssvio_matcher* m, const void* d_matches, const int32_t* d_nmatches, int max_rows,
                              const int32_t* pairs, int npairs, const void* d_kps, int kp_stride, const int32_t* d_kpcounts,
                              const void* d_depth, size_t depth_frame_stride, int w, int h, int stride_px,
                              const ssvio_camera* cam, ssvio_correspondences_dev* out);
/* Map::TrackMapPoints(last_match_vec, this_match_vec), src/Map.cpp:100-123: for every last match, in order, the first
 * this-match whose trainIdx equals its queryIdx gives DMatch{queryIdx = this.queryIdx, trainIdx = last.trainIdx,
 * imgIdx = -1, distance = FLT_MAX (cv::DMatch() defaults)}.  out[cap >= nlast].  The O(N^2) search is an index map here. */
int  ssvio_track_map_points(ssvio_matcher* m, const ssvio_dmatch* last, int nlast, const ssvio_dmatch* cur, int ncur,
                            ssvio_dmatch* out, int cap, int* n);
/* Map::InMatchVec(i, vec), src/Map.cpp:151-162, for i = 0..nkp-1 (the loop of ManageMapPoints, :128-131):
 * out[i] = index of the first match with queryIdx == i, or -1 */
int  ssvio_in_match_vec(ssvio_matcher* m, const ssvio_dmatch* vec, int n, int nkp, int32_t* out);
/* Map::ManageMapPoints(key_frame, last_match_vec), src/Map.cpp:125-149, with Frame::GetMapPointID, src/Frame.cpp:138-147:
 * the decisions and values of its loop for keypoints i = 0..nkp-1 (MapPoint objects and the SetObserve bookkeeping stay
 * with the caller, who walks the three arrays in order of i):
 *   match_num[i] = InMatchVec(i, last_match_vec) (:130);
 *   match_num[i] == -1, a new map point (:131-140): world[i][3] = R * Get3DPoint(i) + t, R and t from world_trans =
 *     key_frame->GetAbsPose().inverse() (:127), 16 floats as Eigen::Matrix4f::data() holds them (column-major);
 *     map_point[i] = -1;
 *   otherwise (:141-147): map_point[i] = the id GetKeyFrames(-1)->GetMapPointID(last_match_vec[match_num[i]].queryIdx)
 *     returns, i.e. .first of the first pair of obs whose .second is that point id; -1 where the reference finds none
 *     (it then runs off the end of a non-void function); world[i] = 0.
 * obs[nobs][2] = that key frame's _observed_mappid_pid_vec as {map_point_id, point_id} (std::pair<int, int> layout).
 * Both linear searches are index maps here.  Float order: (R(r,0) x + (R(r,1) y + R(r,2) z)) + t_r, no contraction. */
int  ssvio_manage_map_points(ssvio_matcher* m, const ssvio_keypoint* kps, int nkp, const uint16_t* depth, int w, int h, int stride_px,
                             const ssvio_camera* cam, const float* world_trans, const ssvio_dmatch* last_match_vec, int nm,
                             const int32_t* obs, int nobs, int32_t* match_num, int32_t* map_point, float* world);

/* ---- per-pixel cloud (first half) ---------------------------------------------------------- */
/* The cloud loop of XtionCamera::GrabImage, driver/Xtion_Driver.cpp:219-229, over GetRGB3DPoint (:138-162): one
 * pcl::PointXYZRGB (32 bytes: x, y, z, 1.0f, b, g, r, a = 255, 12 bytes of padding written as 0) per pixel, point index
 * u * h + v for pixel (x = u, y = v).  depth: 16U w x h; bgr: 8UC3 with bytes B, G, R (:215).  depth_scale is the driver's
 * DOUBLE _DepthScale (driver/Xtion_Driver.h:47: (float)yaml DepthScale / 65535.0), cam->depth_scale is not used.
 * out: w*h records. */
int  ssvio_rgb_cloud(ssvio_matcher* m, const uint16_t* depth, int depth_stride_px, const uint8_t* bgr, int bgr_stride,
                     int w, int h, const ssvio_camera* cam, double depth_scale, void* out);

/* ---- point-cloud map (second half) ------------------------------------------------------- */
/* Map::MapPointCloudFusion, src/Map.cpp:164-197.  The map (_final_cloud) lives in HBM inside the matcher handle; every
 * ssvio_map_fuse* call is ONE iteration of the reference's key-frame loop (:166-192):
 *   the frame's cloud is transformed with Eigen::Affine3f::Identity().rotate(R).translate(t), R / t the rotation / translation
 *   blocks of this_trans = GetAbsPose().inverse() (:174-181) -- as written that is p' = R*p + R*t --, appended to the map (:183),
 *   and the whole is filtered by pcl::VoxelGrid with leaf voxel_size (= _VoxelSize, config/default_conf.yaml:23) on all axes
 *   (:185-190); the filtered cloud is the new map (:192).
 * PCL (transformPointCloud, VoxelGrid::applyFilter with CentroidPoint, version 1.8 / 1.9) is restated from PCL's published source
 * (DESIGN.md); parity unpinned (PCL is absent from this image).  One canonical choice: PCL sorts a voxel's points with the unstable std::sort
 * before it sums them; here they are summed in their order in the concatenated cloud (map points first, then the frame's).
 * Output order: ascending voxel index, as PCL.  Points are 32-byte pcl::PointXYZRGB records (ssvio_rgb_cloud's layout); outputs
 * carry data[3] = 1 and zeroed padding.  this_trans: 16 floats in Eigen::Matrix4f::data() order (column-major).  Non-finite
 * coordinates are refused (the reference's clouds are dense).  When the voxel grid would overflow 32-bit indices PCL returns its
 * input unchanged; so does this.  Each call waits for its result (the new map size sizes the next call). */
int  ssvio_map_set(ssvio_matcher* m, const void* points, int n);                 /* load a map from the host; n = 0 clears it */
int  ssvio_map_fuse(ssvio_matcher* m, const void* cloud, int n, const float* this_trans, float voxel_size);       /* host cloud */
int  ssvio_map_fuse_dev(ssvio_matcher* m, const void* d_cloud, int n, const float* this_trans, float voxel_size); /* device cloud */
/* the frame's cloud is built on the device from its depth and BGR images exactly as ssvio_rgb_cloud does, then fused */
int  ssvio_map_fuse_frame(ssvio_matcher* m, const uint16_t* depth, int depth_stride_px, const uint8_t* bgr, int bgr_stride, int w, int h,
                          const ssvio_camera* cam, double depth_scale, const float* this_trans, float voxel_size);
/* *n = points in the map; out (may be NULL) receives them when cap >= *n */
int  ssvio_map_fetch(ssvio_matcher* m, void* out, int cap, int* n);
int  ssvio_map_device(ssvio_matcher* m, const void** d_points, int* n);

#ifdef __cplusplus
}
#endif
#endif
