/* orbx_debug.h — the DIAGNOSTIC ABI of orbx: stage dumps of an extraction and numeric test hooks.  NOT part of the drop-in boundary
 * (include/orbx.h) and not in liborbx.so: these entry points live in orb_slam3_modified_amd/liborbx_debug.so (csrc/orbx_debug.hip), which the
 * parity tests and the profiling tools load NEXT TO the product library.  They take the orbx_ctx a product call created (the context is plain
 * memory: both libraries are built from the same sources) and read what its last extraction left on the device, or run small kernels of their
 * own around the product's device functions.  No stability promise. */
#ifndef ORBX_DEBUG_H
#define ORBX_DEBUG_H

#include "orbx.h"

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#define ORBX_DEBUG_EXPORT __attribute__((visibility("default")))
#else
#define ORBX_DEBUG_EXPORT
#endif

/* The 7x7 Gaussian-blurred copy of mvImagePyramid[level] the descriptors were sampled from
 * (cv::GaussianBlur, src/ORBextractor.cc:1132-1133), for stage-level parity tests.  dst: h rows of w bytes. */
ORBX_DEBUG_EXPORT int orbx_debug_blur_level(orbx_ctx* ctx, int frame, int level, uint8_t* dst, size_t dst_stride);

/* Stage dumps of the last extraction, for parity tests.
 *  stage 0: FAST candidates handed to the quadtree, in the reference's order (vToDistributeKeys,
 *           src/ORBextractor.cc:863-868): packed x | y<<12 | score<<24, border-relative coordinates.
 *  stage 1: keypoints kept by the quadtree, list order (src/ORBextractor.cc:758-776): same packing,
 *           level coordinates.
 * Returns the number of entries (or a negative error); dst may be NULL. */
ORBX_DEBUG_EXPORT int orbx_debug_level_points(orbx_ctx* ctx, int frame, int level, int stage, uint32_t* dst, int cap);

/* Test hook for the two float paths of the descriptor kernel: angle[i] = cv::fastAtan2(y[i], x[i]) (or y[i] itself
 * when angle_is_input), a[i] / b[i] = cosf / sinf(angle * pi/180) as the reference computes them
 * (src/ORBextractor.cc:102,111-112).  Host pointers. */
ORBX_DEBUG_EXPORT int orbx_debug_trig(orbx_ctx* ctx, const float* y, const float* x, int n, int angle_is_input, float* angle, float* a,
                    float* b);

/* Exhaustive test hook for the device cos/sin path: *hash = order-independent 64-bit digest of
 * (cosf, sinf)(angle * pi/180) over the `count` float bit patterns starting at `first_bits`; the oracle computes the same
 * digest with the host glibc, so one call covers every angle in [0, 360] (1.13e9 floats). */
ORBX_DEBUG_EXPORT int orbx_debug_trig_hash(orbx_ctx* ctx, uint32_t first_bits, uint32_t count, uint64_t* hash);

/* The same for cv::fastAtan2: digest over `count` pseudo-random integer moment pairs (|m| <= 3e6, the range IC_Angle
 * produces; every 16th pair has m10 = 0) generated from `seed` by a fixed integer mix on both sides. */
ORBX_DEBUG_EXPORT int orbx_debug_atan_hash(orbx_ctx* ctx, uint32_t seed, uint32_t count, uint64_t* hash);
/* The same for the rotated test pattern of the steered BRIEF (src/ORBextractor.cc:118-120): digest of (ry, rx) of all 512 pattern points
 * over `count` consecutive float bit patterns of the keypoint angle, first_bits + i; honours the "brief_fma" option. */
ORBX_DEBUG_EXPORT int orbx_debug_brief_hash(orbx_ctx* ctx, uint32_t first_bits, uint32_t count, uint64_t* hash);

/* Test hook for the quadtree's exact std::sort: sorts elems[0..n) (n <= 2048; key = high 32 bits, payload = low 32 bits) with
 * the workgroup-parallel restatement of libstdc++'s introsort the kernel uses (src/ORBextractor.cc:697-701 sorts with
 * std::sort and a comparator that leaves ties to the library's internals), one workgroup of `threads` (64..512) threads. */
ORBX_DEBUG_EXPORT int orbx_debug_gnu_sort(orbx_ctx* ctx, uint64_t* elems, int n, int threads);

/* Counter-calibration hook: copies nbytes (multiple of 16) from d_src to d_dst on the device with `width` (1, 4 or
 * 16) bytes per lane per access — a kernel with exactly known HBM traffic, used by tools/pmc_traffic.py to calibrate
 * rocprofv3's FETCH_SIZE / WRITE_SIZE for the access widths the extractor kernels use.  Asynchronous on `stream`. */
ORBX_DEBUG_EXPORT int orbx_debug_calib_copy(orbx_ctx* ctx, const void* d_src, void* d_dst, size_t nbytes, int width, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ORBX_DEBUG_H */
