/* orbx — MI355X-native ORB front-end: the C-ABI drop-in boundary.
 *
 * The reference has no FFI layer: its boundary for this path is three C++ classes compiled into
 * libORB_SLAM3.so.  Everything those classes compute is exported here as plain C
 * (pointers + sizes, no C++/torch types); include/ORBextractor.h, include/ORBmatcher.h and
 * include/ORBVocabulary.h are header-only adapters that re-create the reference signatures on top of
 * these entry points, so Frame.cc / Tracking.cc / LocalMapping.cc compile unchanged.
 *
 * Conventions: functions return ORBX_OK (0) or a negative error code and never throw; a context is not
 * thread-safe, distinct contexts are (the reference runs two extractor instances concurrently in stereo,
 * src/Frame.cc:122-125).  "d_" pointers are device (HBM) pointers on the context's GPU, all others are
 * host pointers.  `stream` is a hipStream_t passed as void* (NULL = the context's own stream).
 * There is no CPU fallback: without a gfx950 device every compute entry point fails with ORBX_E_DEVICE.
 */
#ifndef ORBX_H
#define ORBX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORBX_OK 0
#define ORBX_E_INVALID (-1)   /* bad argument */
#define ORBX_E_EMPTY (-2)     /* empty image: the reference returns -1 (src/ORBextractor.cc:1090-1091) */
#define ORBX_E_DEVICE (-3)    /* no usable GPU / HIP runtime error (see orbx_last_error) */
#define ORBX_E_CAPACITY (-4)  /* an internal or caller buffer is too small */
#define ORBX_E_FORMAT (-5)    /* vocabulary file malformed */
#define ORBX_E_TIMEOUT (-6)   /* a bounded wait ran out (orbx_replay_wait_gathered_host) */

typedef struct orbx_ctx orbx_ctx;
typedef struct orbx_voc orbx_voc;

/* Layout-identical to cv::KeyPoint (pt.x, pt.y, size, angle, response, octave, class_id): 28 bytes. */
typedef struct orbx_keypoint {
  float x, y, size, angle, response;
  int32_t octave, class_id;
} orbx_keypoint;

/* ---- extractor: replaces ORB_SLAM3::ORBextractor (include/ORBextractor.h:49-83) ------------------- */

/* ORBextractor::ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)
 * src/ORBextractor.cc:409-469.  device_id < 0 selects the current HIP device. */
int orbx_create(orbx_ctx** out, int nfeatures, float scale_factor, int nlevels, int ini_th_fast, int min_th_fast,
                int device_id);
void orbx_destroy(orbx_ctx* ctx);
const char* orbx_last_error(const orbx_ctx* ctx);

/* Rows a caller must provide per frame for keypoints / descriptors: nfeatures + 3*nlevels. */
int orbx_keypoint_capacity(const orbx_ctx* ctx);

/* GetLevels / GetScaleFactor(s) / GetInverseScaleFactors / GetScaleSigmaSquares / GetInverseScaleSigmaSquares
 * (include/ORBextractor.h:63-82) and the per-level feature quotas (src/ORBextractor.cc:434-445).
 * Any output pointer may be NULL; arrays hold nlevels entries. */
int orbx_levels(const orbx_ctx* ctx);
int orbx_scale_tables(const orbx_ctx* ctx, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                      int32_t* features_per_level);

/* ORBextractor::operator()(image, mask [ignored], keypoints, descriptors, vLappingArea)
 * src/ORBextractor.cc:1086-1168, one frame, host buffers (H2D image, kernels, D2H results).
 * img: CV_8UC1 rows x cols with `stride` bytes per row.  kps / desc hold orbx_keypoint_capacity() rows
 * (desc: 32 bytes per row).  *n_out = number of keypoints, *mono_index_out = the value operator() returns.
 * Returns ORBX_E_EMPTY for an empty image (reference: -1). */
int orbx_extract(orbx_ctx* ctx, const uint8_t* img, int rows, int cols, size_t stride, int lap0, int lap1,
                 orbx_keypoint* kps, uint8_t* desc, int* n_out, int* mono_index_out);

/* The same computation over `nframes` frames of one shape that are already resident in HBM, results left in
 * HBM (the batch-replay / benchmark path; frames are independent).
 *   d_imgs   : frame f, row r at d_imgs + f*frame_stride + r*row_stride
 *   d_kps    : [nframes][capacity] orbx_keypoint      d_desc : [nframes][capacity][32] bytes
 *   d_counts : [nframes][2] int32 = {n keypoints, monoIndex}; n = -1 marks a frame whose quadtree overflowed a level's
 *              capacity (never expected; the host entry points turn it into ORBX_E_CAPACITY)
 * Asynchronous on `stream`. */
int orbx_extract_batch_device(orbx_ctx* ctx, const uint8_t* d_imgs, int nframes, int rows, int cols, size_t row_stride,
                              size_t frame_stride, int lap0, int lap1, orbx_keypoint* d_kps, uint8_t* d_desc,
                              int32_t* d_counts, void* stream);

/* Image ingestion fused behind the upload: what Tracking::GrabImageMonocular does before the extractor
 * sees the frame (src/Tracking.cc:1572-1585, cv::cvtColor COLOR_RGB2GRAY / BGR2GRAY / RGBA2GRAY / BGRA2GRAY) runs on the
 * device; img is interleaved, channels = 3 or 4, rgb_order != 0 when R comes first (mbRGB).  OpenCV's 8-bit formula
 * (R*9798 + G*19235 + B*3735 + 2^14) >> 15.  The grey plane is level 0 of the pyramid afterwards.  Otherwise as orbx_extract. */
int orbx_extract_color(orbx_ctx* ctx, const uint8_t* img, int rows, int cols, size_t stride, int channels, int rgb_order, int lap0,
                       int lap1, orbx_keypoint* kps, uint8_t* desc, int* n_out, int* mono_index_out);

/* The other ingestion step: System::TrackMonocular / TrackStereo / TrackRGBD resize the incoming image to the configured size
 * before tracking (cv::resize(im, resizedIm, settings_->newImSize()), src/System.cc:441-446, INTER_LINEAR; an exact 2 x 2
 * downscale is OpenCV's INTER_AREA shortcut).  img: CV_8UC1 rows x cols; new_rows x new_cols = newImSize; the resize runs on
 * the device behind the upload (the pyramid's fixed-point bilinear kernel) and its output is level 0.  Otherwise as orbx_extract. */
int orbx_extract_resized(orbx_ctx* ctx, const uint8_t* img, int rows, int cols, size_t stride, int new_rows, int new_cols, int lap0, int lap1,
                         orbx_keypoint* kps, uint8_t* desc, int* n_out, int* mono_index_out);

/* Host-buffer convenience over the batch path (H2D all frames, extract, D2H). counts: [nframes][2]. */
int orbx_extract_batch(orbx_ctx* ctx, const uint8_t* imgs, int nframes, int rows, int cols, size_t row_stride,
                       size_t frame_stride, int lap0, int lap1, orbx_keypoint* kps, uint8_t* desc, int32_t* counts);

/* mvImagePyramid[level] of frame `frame` of the last extraction (include/ORBextractor.h:83), copied to host
 * (the reference's stereo matcher reads it, src/Frame.cc:818,908-925).  dst may be NULL to query w/h. */
int orbx_pyramid_level(orbx_ctx* ctx, int frame, int level, uint8_t* dst, size_t dst_stride, int* w, int* h);

/* "Whose rows are these": after orbx_extract / orbx_extract_color / orbx_extract_resized an adapter that has copied the frame's descriptor
 * rows into the caller's own buffer (Frame::mDescriptors, src/Frame.cc:311) names that buffer here, so that a later
 * orbx_bow_transform_published on the very same buffer finds the records the extraction graph already computed (below).  Contract (the
 * reference's own): nobody writes into the buffer after the extraction.  n = the extraction's keypoint count; a mismatch makes the call a
 * no-op.  (Rounds 3-4 also handed these rows to the first search target made from the buffer, device to device; that path measured
 * slower than the host rows — 58.0 vs 51.2 us for the first search of a frame — and was removed: targets always take the host bytes.) */
int orbx_publish_descriptors(orbx_ctx* ctx, const void* host_desc, int n);

/* Host mirror of mvImagePyramid for the single-frame path: with orbx_set_host_pyramid(ctx, 1) every orbx_extract also
 * copies the pyramid levels >= 1 of its frame into pinned host memory (one asynchronous copy inside the call);
 * orbx_host_pyramid_level then returns a pointer into that mirror, valid until the next extraction of the context
 * (the reference re-points mvImagePyramid on every call too).  Level 0 is the caller's own image:
 * *data = NULL. */
int orbx_set_host_pyramid(orbx_ctx* ctx, int on);
/* Allocate the context's device buffers for batches of up to `nframes` frames of rows x cols now instead of inside the
 * first extraction call (the buffers are persistent and grow-only; a later call with another shape re-sizes them). */
int orbx_reserve(orbx_ctx* ctx, int rows, int cols, int nframes);
/* Options of one context.  Scheduling / launch-shape knobs (results never depend on them; the ORBX_* environment variables set
 * the defaults at orbx_create): "fork_blur" | "fork_fast0" | "fork_qt" (0|1: run that kernel on a second stream
 * beside its neighbour), "graph" (0|1), "fast_threads" (64|128|256), "fast_pk" (0|1), "desc_k" (1|2|4|8|16),
 * "desc_lds" (0|1), "desc_fused_blur" (-1|0|1: the Gaussian inside the descriptor kernel — -1 = where it pays), "streams" (1|2), "fast_stage_dma" (0|1: the FAST tile by LDS-DMA loads), "qt_fused" (0|1), ... — the full table with
 * defaults is INTEGRATION.md section 7.
 * The FIVE options that DO change results select which build of "the reference CPU path" the output equals, bit for bit (INTEGRATION.md
 * section 6 has the release table; include/orbx_cv_calibrate.h finds the values for the OpenCV at hand, tools/validate_opencv.cpp checks
 * them; also ORBX_GAUSS_KERNEL / ORBX_GAUSS_ROUND / ORBX_GAUSS_TAIL / ORBX_ATAN_FMA / ORBX_BRIEF_FMA in the environment at orbx_create):
 *   "gauss_kernel"  cv::GaussianBlur(7x7, sigma 2) of CV_8UC1 (src/ORBextractor.cc:1133), the 8.8 fixed-point weights:
 *                   0 (default) = {18,34,48,56,48,34,18}, rounding error diffused, sum 256 — OpenCV >= 4.5.1;
 *                   1 = {18,34,49,55,49,34,18}, every coefficient rounded on its own, sum 257 — OpenCV 3.x .. 4.5.0
 *   "gauss_round"   its column pass: 0 (default) = (acc + 2^15) >> 16; 1 = exact ties to even; 2 = floor.  Every variant saturates to 255.
 *   "gauss_tail"    V in {0, 4, 8, 16, 32, 64}: the last (width mod V) columns of every row round as "gauss_round" 0 — the scalar tail of a
 *                   SIMD column loop of vector length V (0 = no tail)
 *   "atan_fma"      cv::fastAtan2 (src/ORBextractor.cc:102): 0 (default) = separate multiply / add; 1 = the Horner steps and 90 - p*c fused
 *                   (OpenCV's AVX2 dispatch copy)
 *   "brief_fma"     the reference's own pattern rotation (src/ORBextractor.cc:118-120): 0 (default) = separate operations;
 *                   1 = fma(x, b, y*a), fma(x, a, -(y*b)) — what -march=native makes of it on an FMA machine
 * Unknown names and out-of-range values return ORBX_E_INVALID. */
int orbx_set_option(orbx_ctx* ctx, const char* name, int value);
int orbx_get_option(const orbx_ctx* ctx, const char* name);   /* the current value (all options are >= 0); ORBX_E_INVALID for an unknown name */
int orbx_host_pyramid_level(orbx_ctx* ctx, int level, const uint8_t** data, size_t* stride, int* w, int* h);

/* The five result-changing options by NAME: which build of "the reference CPU path" (src/ORBextractor.cc over some OpenCV, compiled with
 * some flags: CMakeLists.txt:10-13,33, README.md:560) the output equals, bit for bit.  Profiles (INTEGRATION.md section 6 maps each to its
 * option values and says what is recalled and what is pinned):
 *   "opencv>=4.5.1" (= "default": what orbx_create gives)      "opencv-4.4" (= "opencv-4.4-avx2": OpenCV 3.4.2 .. 4.5.0, AVX2 dispatch)
 *   "opencv-4.4-sse" | "opencv-4.4-avx512" | "opencv-4.4-scalar"  (the same releases, other vector lengths / no SIMD)
 *   "opencv-3.2" (= "opencv<=3.4.1")
 * fma_build: bit 0 = src/ORBextractor.cc itself was built with -march=native on an FMA machine ("brief_fma"); bit 1 = OpenCV runs its AVX2
 * (FMA-contracted) dispatch copy of cv::fastAtan2 ("atan_fma"; refused for profiles whose OpenCV has none).  3 = a native build on an AVX2 host.
 * orbx_cpu_profile_values is the table alone (no context, no device): values = {gauss_kernel, gauss_round, gauss_tail, atan_fma, brief_fma}.
 * orbx_get_cpu_profile reports the ACTIVE set ("name +flags (option=value ...)"; "custom" when the Gaussian triple matches no profile).
 * include/ORBextractor.h does not need these: it calibrates itself against the OpenCV it is built with (include/orbx_cv_calibrate.h). */
const char* orbx_cpu_profile_name(int i);   /* i = 0, 1, ...: NULL ends the table */
const char* orbx_cpu_profile_description(const char* name);
int orbx_cpu_profile_values(const char* name, int fma_build, int values[5]);
int orbx_set_cpu_profile(orbx_ctx* ctx, const char* name, int fma_build);
int orbx_get_cpu_profile(const orbx_ctx* ctx, char* buf, size_t buf_bytes, int values[5]);

/* (Stage dumps and numeric test hooks — orbx_debug_* — are not part of this ABI: include/orbx_debug.h, liborbx_debug.so.) */

/* Per-kernel device time of the extractor, measured with HIP events on the launch stream.
 * orbx_profile_enable(ctx,1) makes every following extraction record events around each kernel;
 * orbx_profile_read returns, for kernel slot i < ORBX_NUM_KERNELS, accumulated milliseconds and launches. */
#define ORBX_NUM_KERNELS 6
int orbx_profile_enable(orbx_ctx* ctx, int on);
int orbx_profile_read(orbx_ctx* ctx, double ms[ORBX_NUM_KERNELS], int64_t launches[ORBX_NUM_KERNELS]);
const char* orbx_kernel_name(int slot);
/* Device time of the single-frame path: with "graph_timing" set (orbx_set_option) every orbx_extract records an event before and after
 * its replayed graph; this returns the last call's elapsed device time in microseconds (< 0: no timed call yet).  The host part of a call
 * (image into the pinned buffer, graph launch, wake-up) is the call's wall time minus this. */
double orbx_last_graph_device_us(orbx_ctx* ctx);
/* The same for a window pass over a resident target (orbx_target_search / _search_view / _nearest): with "window_timing" set, HIP events are
 * recorded around the k_window launch on the stream it runs on; this returns the last pass's device time in microseconds (< 0: none yet). */
double orbx_last_window_device_us(orbx_ctx* ctx);

/* ---- matcher primitives: replace the inner loops of ORB_SLAM3::ORBmatcher --------------------------- */

/* ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:2058-2074; dup FORB::distance, FORB.cpp:77-96):
 * 256-bit Hamming distance of two 32-byte descriptors.  Pure host helper (one pair). */
int orbx_hamming(const uint8_t a[32], const uint8_t b[32]);

/* Guided nearest-neighbour search over CSR candidate lists — the shared inner loop of the 12 Search... / Fuse
 * routines.  Query q (0..nq-1) is compared with train rows cand[row_ptr[q] .. row_ptr[q+1]).
 *   best_idx[q]   : train index of the minimum distance; ties -> FIRST candidate in list order
 *                   (strict `<` update, e.g. src/ORBmatcher.cc:103-111), or LAST when last_wins != 0
 *                   (the `<=` update of SearchForTriangulation, src/ORBmatcher.cc:1017); -1 if no candidate
 *   best_dist[q]  : that distance (256 if none)
 *   second_idx/second_dist[q]: the runner-up under the same (distance, list position) order, i.e. exactly the
 *                   bestDist2 / bestLevel2 bookkeeping of src/ORBmatcher.cc:103-118 (-1 / 256 if none)
 *   dist_out      : optional [nnz] all candidate distances (for the host-side greedy replays), may be NULL
 * Host pointers; q_desc [nq][32], t_desc [nt][32]. */
int orbx_nn_csr(orbx_ctx* ctx, const uint8_t* q_desc, int nq, const uint8_t* t_desc, int nt, const int32_t* row_ptr,
                const int32_t* cand, int last_wins, int32_t* best_idx, int32_t* best_dist, int32_t* second_idx,
                int32_t* second_dist, int32_t* dist_out);

struct orbx_candidate;   /* {int32_t idx, dist}, defined with the window searches below */
/* Guided matching inside vocabulary nodes — SearchByBoW (src/ORBmatcher.cc:223-425, :765-899) and SearchForTriangulation (:901-1146) compare
 * every feature of a node on one side with every feature of the same node on the other.  Queries are GROUPED: all queries of group g
 * (q_group[q] = g) share the candidate list group_cand[group_ptr[g] .. group_ptr[g+1]) (rows of t_desc), so the host never builds the
 * per-query repetition of those lists; and only candidates at Hamming distance <= max_dist are reported — the host replays of those
 * routines never look at the others (TH_LOW for the triangulation search; for the ratio test the bound beyond which it passes anyway).
 *   q_off[q], q_cnt[q] : query q's near candidates are entries[q_off[q] .. q_off[q] + q_cnt[q]), in the order of its group's list
 *   entries[i]         : {idx = row of t_desc, dist}
 *   *n_entries         : entries produced; when that exceeds pool_cap the call returns ORBX_E_CAPACITY and nothing else is valid (the
 *                        caller repeats with a larger pool or uses orbx_nn_csr) */
int orbx_nn_groups(orbx_ctx* ctx, const uint8_t* q_desc, const int32_t* q_group, int nq, const uint8_t* t_desc, int nt,
                   const int32_t* group_ptr, const int32_t* group_cand, int ngroups, int max_dist, int32_t* q_off, int32_t* q_cnt,
                   struct orbx_candidate* entries, int pool_cap, int* n_entries);

/* cv::BFMatcher(NORM_HAMMING).knnMatch(k=2) as used at src/Frame.cc:43,1144: for each query the two nearest
 * train descriptors (ties -> lower train index first).  idx/dist: [nq][2]; missing entries = -1 / 256. */
int orbx_knn2_allpairs(orbx_ctx* ctx, const uint8_t* q_desc, int nq, const uint8_t* t_desc, int nt, int32_t* idx,
                       int32_t* dist);

/* Device-resident variants of the two searches above (all pointers in HBM, async on `stream`). */
int orbx_nn_csr_device(orbx_ctx* ctx, const uint8_t* d_q, int nq, const uint8_t* d_t, int nt, const int32_t* d_row_ptr,
                       const int32_t* d_cand, int last_wins, int32_t* d_best_idx, int32_t* d_best_dist,
                       int32_t* d_second_idx, int32_t* d_second_dist, int32_t* d_dist_out, void* stream);
int orbx_knn2_allpairs_device(orbx_ctx* ctx, const uint8_t* d_q, int nq, const uint8_t* d_t, int nt, int32_t* d_idx,
                              int32_t* d_dist, void* stream);

/* ---- frame grid + guided search (the callers' side of the matcher) --------------------- */

/* Frame::AssignFeaturesToGrid + Frame::GetFeaturesInArea (src/Frame.cc:385-416, :657-735) for many queries at once,
 * on the GPU: the 64 x 48 grid of the frame's keypoints `kps` (image bounds mnMinX..mnMaxY as in src/Frame.cc:153-160)
 * is built once, then query q = (x, y, r, minLevel, maxLevel) returns the keypoint indices the reference returns, in
 * the reference's order (cells x-major, then y, then insertion order — that order decides ties in the searches).
 * CSR output: row_ptr [nq + 1], cand [cand_cap].  Returns the number of candidates (>= 0) or a negative error
 * (ORBX_E_CAPACITY if cand_cap is too small; at most 32 768 keypoints).  Host pointers. */
int orbx_features_in_area(orbx_ctx* ctx, const orbx_keypoint* kps, int n, float min_x, float min_y, float max_x, float max_y,
                          const float* qx, const float* qy, const float* qr, const int32_t* qmin_level, const int32_t* qmax_level,
                          int nq, int32_t* row_ptr, int32_t* cand, int cand_cap);

/* ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) (src/ORBmatcher.cc:648-763):
 * level-0 keypoints of F1 against F2's grid window around vbPrevMatched.  Candidate lists and all Hamming distances
 * are computed on the GPU; the greedy, order-dependent assignment (a later query may steal an earlier match), the
 * TH_LOW / nn_ratio tests and the rotation-histogram filter are replayed on the host in the reference's query order.
 * prev_xy: [n1][2] in/out (vbPrevMatched); matches12: [n1] out (vnMatches12); *nmatches = return value of the
 * reference routine.  kps are the frames' undistorted keypoints, desc their [n][32] descriptors.  Host pointers. */
int orbx_search_for_initialization(orbx_ctx* ctx, const orbx_keypoint* kps1, const uint8_t* desc1, int n1, const orbx_keypoint* kps2,
                                   const uint8_t* desc2, int n2, float min_x, float min_y, float max_x, float max_y, float* prev_xy,
                                   int window_size, float nn_ratio, int check_orientation, int32_t* matches12, int* nmatches);

/* The shared core of the guided searches in ONE call (no candidate lists travelling host -> device): window query
 * over the frame grid (as orbx_features_in_area), the searches' own candidate gates, the Hamming distance of every
 * surviving candidate and the per-query best / second.  kp_skip [n] (optional): 1 = keypoint is never a candidate
 * (e.g. already bound to an observed map point, src/ORBmatcher.cc:81-83).  kp_uright [n] + q_xr [nq] (optional, both or
 * neither): rectified-stereo gate, candidate dropped when uRight > 0 and |q_xr - uRight| > r (:85-90).
 * Out: CSR row_ptr [nq + 1], cand / dist [cand_cap] (either may be NULL), best / second idx / dist [nq] (any may be
 * NULL; -1 / 256 when missing; ties -> first candidate).  Returns the number of candidates or a negative error. */
int orbx_window_search(orbx_ctx* ctx, const orbx_keypoint* kps, const uint8_t* desc, int n, float min_x, float min_y, float max_x,
                       float max_y, const uint8_t* kp_skip, const float* kp_uright, const float* qx, const float* qy, const float* qr,
                       const int32_t* qmin_level, const int32_t* qmax_level, const uint8_t* q_desc, const float* q_xr, int nq,
                       int32_t* row_ptr, int32_t* cand, int32_t* dist, int cand_cap, int32_t* best_idx, int32_t* best_dist,
                       int32_t* second_idx, int32_t* second_dist);

/* A Frame's or KeyFrame's feature grid as the caller holds it (include/Frame.h:250-252, include/KeyFrame.h:318-322, :469):
 *   min_x / min_y = mnMinX / mnMinY (a KeyFrame keeps them truncated to int, include/KeyFrame.h:403-406 — pass what the object
 *   holds), inv_w / inv_h = mfGridElementWidthInv / mfGridElementHeightInv;
 *   cell_start [64*48 + 1] + cell_idx: mGrid[ix][iy] flattened cell by cell, ix major (cell id = ix * 48 + iy), i.e. the
 *   keypoint lists exactly as Frame::AssignFeaturesToGrid left them; cell_start == NULL: the keypoints are assigned on the
 *   device with Frame::PosInGrid's arithmetic (src/Frame.cc:725-735) from min_x / min_y / inv_w / inv_h. */
typedef struct orbx_grid {
  float min_x, min_y, inv_w, inv_h;
  const int32_t* cell_start;
  const int32_t* cell_idx;
} orbx_grid;

/* orbx_window_search over a caller-held grid: Frame::GetFeaturesInArea (src/Frame.cc:657-723) and
 * KeyFrame::GetFeaturesInArea (src/KeyFrame.cc:704-748; it has no level filter: pass -1 / -1, or the level range of the
 * caller's candidate loop) for `nq` queries, every candidate's Hamming distance, best / second per query.  This is the
 * device part of the seven KeyFrame-side routines and of the two-camera blocks of include/ORBmatcher.h
 * (orb_slam3_modified_amd/csrc/ref_adapter/ORBmatcher.cc).  If cand_cap is too small the call fails with ORBX_E_CAPACITY and
 * row_ptr[nq] holds the capacity needed.  Otherwise as orbx_window_search. */
int orbx_window_search_grid(orbx_ctx* ctx, const orbx_keypoint* kps, const uint8_t* desc, int n, const orbx_grid* grid,
                            const uint8_t* kp_skip, const float* kp_uright, const float* qx, const float* qy, const float* qr,
                            const int32_t* qmin_level, const int32_t* qmax_level, const uint8_t* q_desc, const float* q_xr, int nq,
                            int32_t* row_ptr, int32_t* cand, int32_t* dist, int cand_cap, int32_t* best_idx, int32_t* best_dist,
                            int32_t* second_idx, int32_t* second_dist);

/* A search target resident in HBM.  A Frame / KeyFrame is searched many times (Tracking: two or three calls per frame;
 * LocalMapping / LoopClosing: every keyframe against many others) while its keypoints, descriptors and grid never change after
 * construction: orbx_target_create uploads them ONCE (arguments as orbx_window_search_grid / orbx_window_nearest; kp_uright and
 * inv_level_sigma2 may be NULL), orbx_target_search / orbx_target_nearest then move only the queries — no re-upload of the frame,
 * no device-to-host copy, no stream synchronisation: one kernel that reads the queries from, and writes the results to, mapped
 * pinned memory, the host polls a done word.  Results are those of the host-buffer entry points.  A target belongs to the context
 * that created it (one context per thread); destroy it before the context. */
typedef struct orbx_target orbx_target;
int orbx_target_create(orbx_ctx* ctx, const orbx_keypoint* kps, const uint8_t* desc, int n, const orbx_grid* grid, const float* kp_uright,
                       const float* inv_level_sigma2, int nlevels, orbx_target** target);
/* replaces the contents of an existing target (its device block is kept when large enough: a cache of frames recycles targets
 * without allocating).  If the call fails the target is INVALID until a later assign succeeds: orbx_target_search / _nearest /
 * _size on it return ORBX_E_INVALID (never the results of an empty or of the previous target). */
int orbx_target_assign(orbx_ctx* ctx, orbx_target* target, const orbx_keypoint* kps, const uint8_t* desc, int n, const orbx_grid* grid,
                       const float* kp_uright, const float* inv_level_sigma2, int nlevels);
void orbx_target_destroy(orbx_target* target);
int orbx_target_size(const orbx_target* target);   /* number of keypoints; ORBX_E_INVALID for an invalid target */
/* = orbx_window_search_grid on the target; q_xr needs a target created with kp_uright */
int orbx_target_search(orbx_ctx* ctx, const orbx_target* target, const uint8_t* kp_skip, const float* qx, const float* qy, const float* qr,
                       const int32_t* qmin_level, const int32_t* qmax_level, const uint8_t* q_desc, const float* q_xr, int nq, int32_t* row_ptr,
                       int32_t* cand, int32_t* dist, int cand_cap, int32_t* best_idx, int32_t* best_dist, int32_t* second_idx,
                       int32_t* second_dist);
/* orbx_target_search without the copy-out: the candidate lists are read where the kernel wrote them — the call's pinned, mapped blob.
 * Query q's candidates are pool[spans[q].start .. spans[q].start + spans[q].count), in GetFeaturesInArea's order, each with its
 * Hamming distance.  Both pointers stay valid until the SECOND next view call on this context and across its other calls (two blobs are
 * used alternately: a two-camera rig views the left and the right frame back to back).  Returns the total number of candidates. */
/* One record per query of a list view.  start / count: the query's segment of the candidate pool, in the reference's list order.
 * best_* / second_*: the two smallest (distance, list position) among that segment — what a loop over the segment with
 * `if (d < best) { second = best; best = d; } else if (d < second) second = d;` ends with — idx = -1 / dist = 256 when there is none.
 * A caller whose own gates pass on the best (and second) entry can skip the loop: an entry that is minimal in the whole segment is
 * minimal in every subset that contains it. */
typedef struct orbx_list_span { int32_t start, count, best_idx, best_dist, second_idx, second_dist, reserved0, reserved1; } orbx_list_span;
typedef struct orbx_candidate { int32_t idx, dist; } orbx_candidate;
int orbx_target_search_view(orbx_ctx* ctx, const orbx_target* target, const uint8_t* kp_skip, const float* qx, const float* qy, const float* qr,
                            const int32_t* qmin_level, const int32_t* qmax_level, const uint8_t* q_desc, const float* q_xr, int nq,
                            const orbx_list_span** spans, const orbx_candidate** pool);
/* The same call split into ISSUE and WAIT, for a caller with two batches of queries and host work of its own on either side (the drop-in
 * ORBmatcher's per-frame routines: the reference's per-point pre-pass of the second half runs while the device works on the first, the replay
 * of the first half while it works on the second).  _begin packs the queries into the next of the context's two view blobs and queues the window
 * kernel — nothing waits — and returns the slot (0 / 1) to hand to _end; at most one call per blob can be pending, so at most two in flight.
 * _end waits for that call and hands out the view exactly as orbx_target_search_view does (same lifetime rule: valid until the second next view
 * call); a pool that was too small, or a build without the mapped-blob path, makes _end run the ordinary synchronous call.  The query arrays
 * (and kp_skip) must stay valid and unchanged until _end; other calls on the context are allowed in between. */
int orbx_target_search_view_begin(orbx_ctx* ctx, const orbx_target* target, const uint8_t* kp_skip, const float* qx, const float* qy, const float* qr,
                                  const int32_t* qmin_level, const int32_t* qmax_level, const uint8_t* q_desc, const float* q_xr, int nq);
int orbx_target_search_view_end(orbx_ctx* ctx, int slot, const orbx_list_span** spans, const orbx_candidate** pool);
/* Gives back the slot of a _begin whose _end will not be called (an error between the two halves, a dropped ticket); waits for the kernel that
 * may still be writing the blob.  A slot without a pending call: ORBX_OK.  The synchronous orbx_target_search_view never takes a blob with a
 * pending call: with one pending it uses the other, with both pending it returns ORBX_E_INVALID. */
int orbx_target_search_view_cancel(orbx_ctx* ctx, int slot);
/* = orbx_window_nearest on the target; reprojection_gate != 0 needs a target created with kp_uright + inv_level_sigma2, and q_ur */
int orbx_target_nearest(orbx_ctx* ctx, const orbx_target* target, int reprojection_gate, const float* qx, const float* qy, const float* qr,
                        const int32_t* qmin_level, const int32_t* qmax_level, const float* q_ur, const uint8_t* q_desc, int nq, int32_t* best_idx,
                        int32_t* best_dist);

/* Arg-min only — the routines whose queries are independent: Fuse x2 (src/ORBmatcher.cc:1148, :1340) and
 * SearchBySim3 (:1457) take best_idx / best_dist straight from the device, nothing is replayed.  Window + level range as
 * above; first minimum in candidate order wins; -1 / 256 without a candidate.  inv_level_sigma2 != NULL (nlevels entries,
 * then kp_uright [n] and q_ur [nq] are required) adds the reprojection gate of Fuse (:1269-1296): with e = (qx - kp.x,
 * qy - kp.y [, q_ur - kp_uright]) a candidate is dropped when |e|^2 * inv_level_sigma2[kp.octave] > 7.8 (kp_uright >= 0) or
 * > 5.99 (monocular keypoint).  Host pointers. */
int orbx_window_nearest(orbx_ctx* ctx, const orbx_keypoint* kps, const uint8_t* desc, int n, const orbx_grid* grid,
                        const float* kp_uright, const float* inv_level_sigma2, int nlevels, const float* qx, const float* qy,
                        const float* qr, const int32_t* qmin_level, const int32_t* qmax_level, const float* q_ur,
                        const uint8_t* q_desc, int nq, int32_t* best_idx, int32_t* best_dist);

/* Frame::UndistortKeyPoints (src/Frame.cc:747-780): kps_un = kps when dist_coef[0] == 0, otherwise the coordinates go through
 * cv::undistortPoints(pts, pts, K, distCoef, Mat(), K) — OpenCV 4.x's 5-iteration fixed point of the radial (k1 k2 [k3]) +
 * tangential (p1 p2) model in double, as recalled (no OpenCV to pin it against: unpinned like the extractor's five OpenCV
 * primitives); every other keypoint field is copied.  dist_coef = mDistCoef (k1 k2 p1 p2 [k3]), n_coef = 4 or 5; fx fy cx cy =
 * the entries of mK.  The _device form works on resident arrays laid out like the outputs of orbx_extract_batch_device
 * ([nframes][capacity] keypoints, [nframes][2] counts), asynchronously on `stream`: the keypoints never leave HBM between the
 * extraction and the frame grid / guided searches. */
int orbx_undistort_keypoints(orbx_ctx* ctx, const orbx_keypoint* kps, int n, float fx, float fy, float cx, float cy, const float* dist_coef,
                             int n_coef, orbx_keypoint* kps_un);
int orbx_undistort_keypoints_device(orbx_ctx* ctx, const orbx_keypoint* d_kps, const int32_t* d_counts, int nframes, int capacity, float fx,
                                    float fy, float cx, float cy, const float* dist_coef, int n_coef, orbx_keypoint* d_kps_un, void* stream);

/* ORBmatcher::SearchByProjection(Frame& F, const vector<MapPoint*>&, th, bFarPoints, thFarPoints)
 * (src/ORBmatcher.cc:43-141; Tracking::SearchLocalPoints' per-frame call) for single-camera / rectified-stereo frames
 * (F.Nleft == -1), with the MapPoint / Frame state flattened into arrays:
 *   frame:      kps_un = F.mvKeysUn, desc = F.mDescriptors, u_right = F.mvuRight (NULL for monocular),
 *               kp_obs [n] in/out = F.mvpMapPoints[i] ? Observations() : -1, scale_factors = F.mvScaleFactors;
 *   map points: mp_in_view = mbTrackInView && !isBad() && !(bFarPoints && mTrackDepth > thFarPoints), mp_proj_x/y/xr =
 *               mTrackProjX/Y/XR, mp_view_cos = mTrackViewCos, mp_level = mnTrackScaleLevel, mp_desc = GetDescriptor(),
 *               mp_obs = Observations().
 * Windows, gates and all Hamming distances run on the GPU; the order-dependent part (a keypoint bound by an earlier map
 * point of this call is no candidate for later ones) is replayed on the host in the reference's order.
 * kp_match [n] out: index of the map point bound to keypoint i by this call (F.mvpMapPoints[i] = pMP), else -1;
 * *nmatches = the reference's return value. */
int orbx_search_by_projection(orbx_ctx* ctx, const orbx_keypoint* kps_un, const uint8_t* desc, const float* u_right, int32_t* kp_obs, int n,
                              float min_x, float min_y, float max_x, float max_y, const float* scale_factors, int nlevels,
                              const uint8_t* mp_in_view, const float* mp_proj_x, const float* mp_proj_y, const float* mp_proj_xr,
                              const float* mp_view_cos, const int32_t* mp_level, const uint8_t* mp_desc, const int32_t* mp_obs, int nmp,
                              float th, float nn_ratio, int32_t* kp_match, int* nmatches);

/* ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono) (src/ORBmatcher.cc:1676-1885;
 * Tracking::TrackWithMotionModel's per-frame call) for CurrentFrame.Nleft == -1, from the point where the last frame's
 * map points have been projected (:1705-1718 are the caller's pose / camera-model arithmetic):
 *   lp_valid [nlast] = pMP && !mvbOutlier[i] && invzc >= 0 && uv inside the image bounds; lp_u / lp_v = uv; lp_invz = invzc
 *   (read only when u_right is given); lp_octave = nLastOctave; lp_angle = the last frame's keypoint angle; lp_desc =
 *   pMP->GetDescriptor(); lp_obs = pMP->Observations().  direction: 0 neither, 1 bForward, 2 bBackward (:1691-1692).
 * Frame arrays as in orbx_search_by_projection; mbf = CurrentFrame.mbf.
 * kp_match [n] out: -1 untouched, >= 0 index i of the last-frame point now bound to the keypoint, -2 set to NULL by the
 * rotation-consistency filter; kp_obs updated accordingly; *nmatches = the reference's return value. */
int orbx_search_by_projection_last(orbx_ctx* ctx, const orbx_keypoint* kps_un, const uint8_t* desc, const float* u_right, int32_t* kp_obs,
                                   int n, float min_x, float min_y, float max_x, float max_y, const float* scale_factors, int nlevels,
                                   float mbf, const uint8_t* lp_valid, const float* lp_u, const float* lp_v, const float* lp_invz,
                                   const int32_t* lp_octave, const float* lp_angle, const uint8_t* lp_desc, const int32_t* lp_obs, int nlast,
                                   float th, int direction, int check_orientation, int32_t* kp_match, int* nmatches);

/* ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, vector<MapPoint*>& vpMapPointMatches) (src/ORBmatcher.cc:223-425;
 * TrackReferenceKeyFrame, Relocalization) for single-camera frames / keyframes (F.Nleft == -1, pKF->mpCamera2 == NULL).
 * Keyframe side: kf_desc [nkf][32], kf_angle = mvKeysUn[i].angle, kf_valid [nkf] = map point non-NULL && !isBad(), the
 * FeatureVector as CSR (kf_fv_node ascending [n_kf_nodes], kf_fv_ptr [n_kf_nodes + 1], kf_fv_idx = the vectors' contents in
 * order).  Frame side likewise (f_angle = F.mvKeys[i].angle).  All Hamming distances between features of the same
 * vocabulary node run on the GPU (one launch); the greedy part (a frame feature taken by an earlier keyframe feature is
 * no candidate any more), TH_LOW / ratio tests and the rotation filter are replayed in the reference's order.
 * match_kf [nf] out: keyframe feature whose map point lands in vpMapPointMatches[i], -1 for NULL; *nmatches = return value. */
int orbx_search_by_bow(orbx_ctx* ctx, const uint8_t* kf_desc, const float* kf_angle, const uint8_t* kf_valid, int nkf,
                       const uint32_t* kf_fv_node, const int32_t* kf_fv_ptr, const uint32_t* kf_fv_idx, int n_kf_nodes,
                       const uint8_t* f_desc, const float* f_angle, int nf, const uint32_t* f_fv_node, const int32_t* f_fv_ptr,
                       const uint32_t* f_fv_idx, int n_f_nodes, float nn_ratio, int check_orientation, int32_t* match_kf,
                       int* nmatches);

/* Frame::ComputeStereoMatches (src/Frame.cc:811-981) on the DEVICE pyramids of the left and right extractor (the
 * last frame each context extracted; both on one GPU, same image shape) — this is the reader of the reference's public
 * ORBextractor::mvImagePyramid, so with it no pyramid has to be copied to the host.  kps / desc: the keypoints and
 * descriptors the two extractors returned (host pointers); mb = baseline, mbf = baseline * fx (include/Frame.h).
 * u_right / depth: mvuRight / mvDepth, -1 where no match; *nmatches = matches kept after the median filter. */
int orbx_stereo_matches(orbx_ctx* left, orbx_ctx* right, const orbx_keypoint* kpsL, const uint8_t* descL, int nL,
                        const orbx_keypoint* kpsR, const uint8_t* descR, int nR, float mb, float mbf, float* u_right, float* depth,
                        int* nmatches);

/* ---- bag of words: replaces ORBVocabulary = DBoW2::TemplatedVocabulary<cv::Mat, FORB> ----------------- */

/* TemplatedVocabulary::loadFromTextFile (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1338-1424).
 * Blank lines are skipped (the reference manufactures a phantom node from a trailing newline).
 * The tree is uploaded to the GPU of `ctx`. */
int orbx_voc_load_text(orbx_ctx* ctx, const char* path, orbx_voc** out);
/* Build from memory: node i (1-based ids, node 0 = root) has parent[i], is_leaf[i], 32-byte descriptor, weight. */
int orbx_voc_create(orbx_ctx* ctx, int k, int L, int scoring, int weighting, int nnodes_excl_root, const int32_t* parent,
                    const uint8_t* is_leaf, const uint8_t* desc, const double* weight, orbx_voc** out);
void orbx_voc_destroy(orbx_voc* voc);
/* TemplatedVocabulary::saveToTextFile (TemplatedVocabulary.h:1428-1449), byte-identical output (6-digit weights). */
int orbx_voc_save_text(const orbx_voc* voc, const char* path);
/* Exact binary cache of a vocabulary (SURVEY.md §8(f).4 "binary cache"): "ORBXVOC1" + k, L, scoring, weighting + arrays. */
int orbx_voc_save_binary(const orbx_voc* voc, const char* path);
int orbx_voc_load_binary(orbx_ctx* ctx, const char* path, orbx_voc** out);
int orbx_voc_info(const orbx_voc* voc, int* k, int* L, int* nnodes, int* nwords);

/* The per-feature part of TemplatedVocabulary::transform(features, BowVector&, FeatureVector&, levelsup)
 * (TemplatedVocabulary.h:1127-1259): tree descent of every descriptor.
 *   word[i]   : word id of the leaf reached      weight[i] : that word's weight (0 => feature is dropped)
 *   node[i]   : node id at level L - levelsup (the FeatureVector key)
 * The ordered-map accumulation + L1 normalisation (BowVector.cpp:34-85, FeatureVector.cpp:30-45) is
 * orbx_bow_finalize below (host, ascending-id order, double). */
int orbx_bow_transform(orbx_voc* voc, const uint8_t* desc, int n, int levelsup, uint32_t* word, double* weight,
                       uint32_t* node);
int orbx_bow_transform_device(orbx_voc* voc, const uint8_t* d_desc, int n, int levelsup, uint32_t* d_word,
                              double* d_weight, uint32_t* d_node, void* stream);
/* The same for descriptor rows that an extractor context has published (orbx_publish_descriptors: `host_desc` is the caller's buffer that
 * holds the rows of that context's last single-frame extraction, contiguous, n rows) — Frame::ComputeBoW after Frame::ExtractORB
 * (src/Frame.cc:738-745 after :311).  Returns ORBX_OK when the extraction's own graph already ran the descent with this vocabulary and
 * levelsup (the records are copied from the context's pinned result block: no device round trip), 1 when it did not — the caller then
 * uses orbx_bow_transform; the first such call attaches the vocabulary to the publishing context, so that from its next extraction on
 * the single-frame graph ends with the descent.  The buffer must still hold the published bytes (a digest of them is compared). */
int orbx_bow_transform_published(orbx_voc* voc, const void* host_desc, int n, int levelsup, uint32_t* word, double* weight, uint32_t* node);
/* BowVector accumulate (addWeight) + normalize(L1): out ids ascending, values double; returns nnz in *n_out.
 * ids/vals hold n entries. */
int orbx_bow_finalize(const orbx_voc* voc, const uint32_t* word, const double* weight, int n, uint32_t* ids,
                      double* vals, int* n_out);
/* L1Scoring::score (Thirdparty/DBoW2/DBoW2/ScoringObject.cpp:23-68) of two sorted sparse vectors. */
double orbx_bow_score_l1(const uint32_t* ida, const double* va, int na, const uint32_t* idb, const double* vb, int nb);
/* One query vector against many database vectors (CSR), on the GPU: the KeyFrameDatabase scoring loop
 * (src/KeyFrameDatabase.cc:162,303,391,531,662,783).  scores: [ndb]. */
int orbx_bow_score_l1_batch(orbx_ctx* ctx, const uint32_t* q_ids, const double* q_vals, int nq, const int32_t* db_ptr,
                            const uint32_t* db_ids, const double* db_vals, int ndb, double* scores);

/* ---- keyframe database ------------------------------------------------------------------
 * KeyFrameDatabase (src/KeyFrameDatabase.cc) with the keyframes' BowVectors resident in HBM as CSR.  add / erase / clear
 * mirror :39-45 / :47-66 / :68-72.  orbx_kfdb_query is the first two phases shared by DetectLoopCandidates (:100-165),
 * DetectCandidates (:228-310, :355-398), DetectBestCandidates (:468-535), DetectNBestCandidates (:604-665) and
 * DetectRelocalizationCandidates (:733-790): every keyframe sharing at least one word with the query, in the order of the
 * reference's lKFsSharingWords list (query words ascending; inside one word's inverted list, the order of add()), its
 * number of common words, maxCommonWords, minCommonWords = (int)(maxCommonWords * 0.8f) (raised to min_words_floor =
 * DetectBestCandidates' nMinWords, 0 elsewhere) and, for the keyframes with MORE than minCommonWords common words,
 * mpVoc->score(query, keyframe) as a double (callers narrow to float); -1.0 for the others.
 * exclude: keyframe ids that must not enter the list (the query's connected keyframes, keyframes of another map, ...).
 * The covisibility accumulation that follows in each Detect* routine is graph logic of the caller.
 * kf ids are the caller's (KeyFrame::mnId).  Not thread-safe; one database belongs to one context. */
typedef struct orbx_kfdb orbx_kfdb;
int orbx_kfdb_create(orbx_ctx* ctx, orbx_kfdb** out);
void orbx_kfdb_destroy(orbx_kfdb* db);
int orbx_kfdb_add(orbx_kfdb* db, int64_t kf_id, const uint32_t* ids, const double* vals, int n);
int orbx_kfdb_erase(orbx_kfdb* db, int64_t kf_id);
int orbx_kfdb_clear(orbx_kfdb* db);
int orbx_kfdb_size(const orbx_kfdb* db);
int orbx_kfdb_query(orbx_kfdb* db, const uint32_t* q_ids, const double* q_vals, int nq, const int64_t* exclude, int n_exclude,
                    int min_words_floor, int64_t* kf_ids, int32_t* common_words, double* scores, int cap, int* n_sharing,
                    int* max_common_words, int* min_common_words);
/* The two phases on their own, as the drop-in KeyFrameDatabase class uses them (include/KeyFrameDatabase.h,
 * orb_slam3_modified_amd/csrc/ref_adapter/KeyFrameDatabase.cc): orbx_kfdb_sharing = EVERY keyframe sharing a word with the query,
 * in the reference's list order, with its number of common words (no exclusion: which of them enter a routine's list — same map,
 * other map, not connected — and what happens to the others is decided over the caller's own objects); *n_sharing is set even when
 * cap is too small (ORBX_E_CAPACITY).  orbx_kfdb_score = mpVoc->score(query, keyframe) (L1Scoring::score) for exactly the listed
 * keyframes, doubles bit-identical to the reference's. */
int orbx_kfdb_sharing(orbx_kfdb* db, const uint32_t* q_ids, int nq, int64_t* kf_ids, int32_t* common_words, int cap, int* n_sharing);
int orbx_kfdb_score(orbx_kfdb* db, const uint32_t* q_ids, const double* q_vals, int nq, const int64_t* kf_ids, int n, double* scores);

/* ---- batch replay: frames sharded one camera stream per GPU + ONE exchange per step (BASELINE.json config 5) -----------------
 * The reference has no such mode (its System owns one live camera, src/System.cc:197-264); north_star adds it: every GPU (one process each)
 * replays its own stream(s) through the extractor — no collective on the extraction path — and after every step all ranks all-gather their
 * fixed-size feature blocks over RCCL / xGMI, asynchronously on a stream of its own and double-buffered, so that step k's collective runs
 * under step k + 1's kernels.  A C / C++ host drives it with these entry points alone (INTEGRATION.md section 8 has the 20-line loop);
 * liborbx binds RCCL at run time (dlopen: the instance already in the process if there is one, else librccl.so.1; ORBX_RCCL_LIB overrides).
 *
 * Feature block of one rank and step (device memory, one contiguous buffer; every part 256-byte aligned):
 *     [frames][capacity] orbx_keypoint | [frames][capacity][32] descriptor bytes | [frames][2] int32 {n, monoIndex}
 * gather_what: ORBX_GATHER_DESCRIPTORS moves the tail of the block (descriptor rows + counts: north_star's "all-gather of descriptors"),
 * ORBX_GATHER_BLOCKS the whole block (SURVEY §8(e)'s), ORBX_GATHER_NONE nothing (the sharded extraction alone).
 *
 * lanes: 1 .. 16 extractor contexts of ONE device with identical parameters, owned by the caller, each on a stream of its own (two lanes: +7.6 %
 * on 256 x 640x480 — they drift out of phase and fill each other's idle issue slots).  Schedule: by default the lanes take WHOLE steps in turn
 * (step k runs on lane k mod L over all `frames` frames while the other lanes are still busy with the steps before it: +3.5 % over the split
 * form, every launch covers the whole batch; a step's results are complete one step later and every lane holds buffers for `frames` frames);
 * orbx_set_option(lanes[0], "replay_alternate", 0) before creation (or ORBX_REPLAY_ALTERNATE=0) = every lane works on its contiguous share of
 * every step.  orbx_replay_lane_range tells which frames a lane covers ([0, frames) for every lane under the default schedule).
 * Transport, by argument: unique_id != NULL -> ncclCommInitRank(world, unique_id, rank), the 128 bytes coming from orbx_replay_unique_id() on
 * one rank and reaching the others by whatever the host has (a file, MPI, a TCP store); host_exchange != NULL -> the caller's own host
 * all-gather, the block staged through pinned memory (tests, hosts without RCCL between their ranks); both NULL -> world must be 1 and a
 * one-rank RCCL group is made (the self-gather: the collective's own cost on this GPU).  Not thread-safe; one engine per thread.
 * When orbx_replay_create fails, the reason is in orbx_last_error(lanes[0]) (there is no engine yet to ask); afterwards in orbx_replay_last_error. */
typedef struct orbx_replay orbx_replay;
#define ORBX_REPLAY_UNIQUE_ID_BYTES 128   /* sizeof(ncclUniqueId) */
#define ORBX_GATHER_NONE 0
#define ORBX_GATHER_DESCRIPTORS 1
#define ORBX_GATHER_BLOCKS 2
/* recv[r * bytes_per_rank ..) := rank r's `send`, for every r, on every rank; returns 0 on success.  Called from orbx_replay_step. */
typedef int (*orbx_host_exchange_fn)(void* user, const void* send, void* recv, size_t bytes_per_rank);
int orbx_replay_unique_id(uint8_t id[ORBX_REPLAY_UNIQUE_ID_BYTES]);   /* ncclGetUniqueId */
const char* orbx_replay_rccl_info(void);                               /* "rccl 2.x.y (library)" or why none could be loaded */
int orbx_replay_create(orbx_replay** out, orbx_ctx* const* lanes, int nlanes, int frames, int rows, int cols, int gather_what, int rank, int world,
                       const uint8_t* unique_id, orbx_host_exchange_fn host_exchange, void* user);
/* orbx_replay_create in two halves, FOR HOSTS WITH MORE THAN ONE RANK: prepare = everything a rank can fail at on its own (argument and
 * lane checks, buffers, streams, resolving the RCCL library; use_rccl != 0: the transport will be RCCL) and touches no other rank;
 * connect = ncclCommInitRank, in which a rank waits for ALL its peers.  A rank that failed locally never arrives, so the host must agree over
 * its own control plane (an all-reduce of "prepare succeeded") that every rank is ready BEFORE any rank calls connect — bench.py and
 * replay.py do.  orbx_replay_create = prepare + connect for one-rank hosts and hosts that accept that risk.  The lanes must agree in everything
 * that decides a result byte (nfeatures, levels, scale factor, thresholds, the five CPU-path options): a mismatch is ORBX_E_INVALID with the
 * reason in orbx_last_error(lanes[0]).  The engine sets the lanes' stream-fork options for its launch shape and puts them back in destroy. */
int orbx_replay_prepare(orbx_replay** out, orbx_ctx* const* lanes, int nlanes, int frames, int rows, int cols, int gather_what, int rank, int world,
                        int use_rccl, orbx_host_exchange_fn host_exchange, void* user);
int orbx_replay_connect(orbx_replay* r, const uint8_t* unique_id);     /* unique_id NULL: world must be 1 (the self-gather) */
void orbx_replay_destroy(orbx_replay* r);                              /* waits for everything in flight; the lanes stay the caller's */
const char* orbx_replay_last_error(const orbx_replay* r);
const char* orbx_replay_transport(const orbx_replay* r);
/* sizes and offsets of the block, and of what the exchange moves (any pointer may be NULL) */
int orbx_replay_layout(const orbx_replay* r, int* frames, int* capacity, size_t* block_bytes, size_t* desc_off, size_t* counts_off, size_t* send_off,
                       size_t* send_bytes, int* nlanes);
int orbx_replay_lane_range(const orbx_replay* r, int lane, int* f0, int* f1);   /* frames [f0, f1) of a step's batch */
/* One step: the hot path over `frames` resident frames (frame f, row y at d_frames + f*frame_stride + y*row_stride) into block (step & 1), then —
 * when the exchange is on — the all-gather of that block into gathered buffer (step & 1), queued behind the step's kernels.  Returns at once
 * (everything is asynchronous; with a host_exchange the call waits for this step's kernels) with the buffer index 0 / 1, or a negative code. */
int orbx_replay_step(orbx_replay* r, const uint8_t* d_frames, size_t row_stride, size_t frame_stride, int lap0, int lap1);
int orbx_replay_drain(orbx_replay* r);                                 /* wait for every lane and for the gather stream */
int orbx_replay_set_gather(orbx_replay* r, int on);                    /* switch the exchange off / on between steps (measurements) */
int orbx_replay_block(orbx_replay* r, int i, uint8_t** d_block);       /* this rank's block i (device pointer; layout above) */
int orbx_replay_gathered(orbx_replay* r, int i, int rank, const uint8_t** d_part);   /* rank's send_bytes inside gathered buffer i (device pointer) */
/* Ordering a consumer of gathered buffer i against ONE step's exchange without draining the engine (the pointer above carries no ordering:
 * the collective runs on the engine's private gather stream).  wait: `consumer` (a hipStream_t, passed as void* so that this header needs no
 * HIP) waits ON THE DEVICE for the last collective queued into buffer i; everything launched on it afterwards sees the gathered data.
 * release: records "the consumer has finished with buffer i" at the current tail of `consumer`; the next collective INTO buffer i (two steps
 * later) waits for it.  A consumer that does neither must call orbx_replay_drain.  wait_host: the same wait on the host, bounded:
 * ORBX_E_TIMEOUT after timeout_ms (< 0: unbounded) — the sign of a peer that left; see orbx_replay_abort. */
int orbx_replay_wait_gathered(orbx_replay* r, int i, void* consumer_stream);
int orbx_replay_release_gathered(orbx_replay* r, int i, void* consumer_stream);
int orbx_replay_wait_gathered_host(orbx_replay* r, int i, int timeout_ms);
/* Failure containment.  A lane error inside orbx_replay_step does NOT keep the rank out of that step's exchange (the other ranks would wait in
 * theirs for ever): the rank takes part, in this and in every later step, with a POISONED block — every {n, monoIndex} of the block is
 * {-1, -1}, which no extraction produces — and orbx_replay_step returns the first error from then on (orbx_replay_failed: 0 or that code).
 * The host tells its peers by its own means and leaves with orbx_replay_destroy (all ranks together) or orbx_replay_abort (ncclCommAbort: no
 * hand-shake with ranks that may be gone; the engine keeps working with the exchange off). */
int orbx_replay_failed(const orbx_replay* r);
int orbx_replay_abort(orbx_replay* r);
/* host copies (drain first): what = 0: block i, 1: gathered buffer i (world * send_bytes); orbx_replay_write_block is the reverse, for block i */
int orbx_replay_read(orbx_replay* r, int what, int i, void* host_dst, size_t offset, size_t nbytes);
int orbx_replay_write_block(orbx_replay* r, int i, const void* host_src, size_t offset, size_t nbytes);
/* average device time of one step's collective (HIP events on the gather stream) since the last reset; *avg_ms = -1 when none was timed */
int orbx_replay_gather_ms(orbx_replay* r, double* avg_ms, long long* n, int reset);

#ifdef __cplusplus
}
#endif
#endif /* ORBX_H */
