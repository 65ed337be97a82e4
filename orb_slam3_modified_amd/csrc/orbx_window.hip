// orbx window pass — the device part shared by the guided searches of ORBmatcher (src/ORBmatcher.cc): for a batch of
// queries (centre, radius, level range, descriptor) over ONE feature grid
//   * the window of Frame::GetFeaturesInArea (src/Frame.cc:657-723) / KeyFrame::GetFeaturesInArea (src/KeyFrame.cc:704-748),
//     candidates in the reference's order (grid cells x-major, then y, then insertion order — that order decides ties),
//   * the routines' static candidate gates (keypoint excluded, rectified-stereo consistency :85-90, Fuse's reprojection
//     test :1269-1296),
//   * ORBmatcher::DescriptorDistance of every surviving candidate,
//   * the running best / second of the candidate loops (:96-118; first minimum wins),
// in ONE kernel, one wave per query.  Host-buffer entry points move their inputs in one pinned blob (one H2D copy), launch,
// and read the results back in one copy: no per-call allocation, no mid-call synchronisation.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "orbx_internal.h"

namespace orbx {

constexpr int kWinCols = 64, kWinRows = 48, kWinCells = kWinCols * kWinRows;   // include/Frame.h:44-45

struct WinQueryOut { int32_t start, count, best_idx, best_dist, second_idx, second_dist, pad0, pad1; };
struct WinQueryIn { float x, y, r, aux; int32_t lo, hi, pad0, pad1; uint32_t desc[8]; };   // 64 bytes
static_assert(sizeof(WinQueryIn) == 64, "one cache line per query");

struct WinArgs {
  const orbx_keypoint* kps;
  const uint8_t* desc;
  const int32_t* cell_start;   // [kWinCells + 1]
  const int32_t* cell_idx;
  float minX, minY, invW, invH;
  const WinQueryIn* qin;   // one 64-byte record per query: a wave fetches its query in ONE coalesced read (the records may sit in mapped host
                           // memory, where nine small reads per query cost nine PCIe transactions each)
  int has_aux;             // the records carry q_xr / q_ur
  int nq;
  const uint8_t* kp_skip;
  const float* kp_uright;
  const float* inv_sigma2;
  WinQueryOut* out;
  int2* pool;          // {candidate index, distance}, one contiguous segment per query
  int pool_cap;
  int32_t* total;      // [0] candidates of all queries (also when the pool is too small: the host then reports the capacity needed),
                       // [1] queries finished; both zeroed by the input upload
  int32_t* out_hdr;    // the wave that finishes last copies the total here (and sets word 1), so that header + records + pool leave in ONE copy
  int host_out;        // out / pool / out_hdr are mapped host memory
  const uint8_t* skip_map;   // kp_skip in mapped host memory: staged into LDS by every workgroup (n_skip bytes, n_skip % 4 == 0)
  int n_skip;
  int self_reset;      // the last wave zeroes `total` again (resident-target calls keep the two counters in the context)
  int compact;         // 1: only {start, count} per query are written (the caller did not ask for best / second)
};

struct WinQueryShort { int32_t start, count; };

struct WDesc { unsigned long long w[4]; };
__device__ __forceinline__ WDesc wload(const uint8_t* p) {
  const uint4* q = (const uint4*)p;
  const uint4 a = q[0], b = q[1];
  WDesc d;
  d.w[0] = (unsigned long long)a.x | ((unsigned long long)a.y << 32);
  d.w[1] = (unsigned long long)a.z | ((unsigned long long)a.w << 32);
  d.w[2] = (unsigned long long)b.x | ((unsigned long long)b.y << 32);
  d.w[3] = (unsigned long long)b.z | ((unsigned long long)b.w << 32);
  return d;
}
__device__ __forceinline__ int wham(const WDesc& a, const WDesc& b) {
  return __popcll(a.w[0] ^ b.w[0]) + __popcll(a.w[1] ^ b.w[1]) + __popcll(a.w[2] ^ b.w[2]) + __popcll(a.w[3] ^ b.w[3]);
}

// key = distance << 48 | position in the candidate list << 24 | keypoint index: ascending keys = the reference's
// "strict <, first candidate wins" order; position and index below 2^24
constexpr unsigned long long kNoWinKey = ~0ull;

#ifndef ORBX_WIN_WPG
#define ORBX_WIN_WPG 16   // measured 4 / 8 / 16: 13.7 / 11.8 / 10.6 us per 800-query pass (best and second only)
#endif
constexpr int kWinWPG = ORBX_WIN_WPG;   // queries (waves) per workgroup: the wide fence and the two global atomics are paid once per workgroup

template <bool LISTS, bool CHI2>
__global__ __launch_bounds__(64 * kWinWPG) void k_window(const WinArgs a) {
  extern __shared__ __align__(16) uint8_t lskip[];
  // One global atomic per WORKGROUP, not per query, for the pool reservation and for the completion count: 800 waves hitting one
  // address serialise in the L2 (measured: 20 of this kernel's 23 us); the waves of a workgroup meet in LDS first.
  __shared__ int wg_cnt, wg_base, wg_done;
  if (threadIdx.x == 0) { wg_cnt = 0; wg_done = 0; }
  if (a.skip_map) {   // block-uniform; the flags change with every call and are gathered at random: over PCIe once per workgroup, then LDS
    for (int i = threadIdx.x * 4; i < a.n_skip; i += 64 * kWinWPG * 4) *(uint32_t*)(lskip + i) = *(const uint32_t*)(a.skip_map + i);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int q0 = blockIdx.x * kWinWPG + (threadIdx.x >> 6);
  const bool active = q0 < a.nq;               // the spare waves of the last workgroup shadow its last query (they take part in the
  const int q = active ? q0 : a.nq - 1;        // workgroup barriers) and write nothing
  const int nactive = min(kWinWPG, a.nq - (int)blockIdx.x * kWinWPG);
  const uint32_t rec = ((const uint32_t*)(a.qin + q))[lane & 15];   // 64 bytes, one request
  const float x = __uint_as_float(__builtin_amdgcn_readlane(rec, 0)), y = __uint_as_float(__builtin_amdgcn_readlane(rec, 1));
  const float r = __uint_as_float(__builtin_amdgcn_readlane(rec, 2));
  const float aux = a.has_aux ? __uint_as_float(__builtin_amdgcn_readlane(rec, 3)) : 0.f;
  const int minLevel = (int)__builtin_amdgcn_readlane(rec, 4), maxLevel = (int)__builtin_amdgcn_readlane(rec, 5);
  const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
  // src/Frame.cc:665-687 with separately rounded float operations
  const int nMinCellX = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, a.minX), r), a.invW)));
  const int nMaxCellX = min(kWinCols - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, a.minX), r), a.invW)));
  const int nMinCellY = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, a.minY), r), a.invH)));
  const int nMaxCellY = min(kWinRows - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, a.minY), r), a.invH)));
  const bool any = nMinCellX < kWinCols && nMaxCellX >= 0 && nMinCellY < kWinRows && nMaxCellY >= 0;

  auto passes = [&](int idx) -> bool {
    const orbx_keypoint* kp = a.kps + idx;
    const int octave = kp->octave;
    const float kx = kp->x, ky = kp->y;
    bool ok = true;
    if (bCheckLevels) {
      if (octave < minLevel) ok = false;
      if (maxLevel >= 0 && octave > maxLevel) ok = false;
    }
    ok = ok && fabsf(__fsub_rn(kx, x)) < r && fabsf(__fsub_rn(ky, y)) < r;
    if (a.skip_map ? lskip[idx] != 0 : (a.kp_skip && a.kp_skip[idx])) ok = false;
    if (CHI2) {
      if (ok) {
        const float ur = a.kp_uright[idx];
        const float ex = __fsub_rn(x, kx), ey = __fsub_rn(y, ky);
        float e2 = __fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey));
        double lim = 5.99;
        if (ur >= 0.f) { const float er = __fsub_rn(aux, ur); e2 = __fadd_rn(e2, __fmul_rn(er, er)); lim = 7.8; }
        if ((double)__fmul_rn(e2, a.inv_sigma2[octave]) > lim) ok = false;
      }
    } else if (a.kp_uright && a.has_aux) {
      const float ur = a.kp_uright[idx];
      if (ur > 0.f && fabsf(__fsub_rn(aux, ur)) > r) ok = false;
    }
    return ok;
  };

  // The window is a short list of CSR segments — one per grid column, the rows of a column are contiguous — and everything about a
  // candidate hangs on a chain of dependent loads (segment bounds -> keypoint index -> keypoint -> descriptor).  Walking the columns
  // one after the other pays that chain per column (a latency-bound kernel: ~1 us per link on a GPU that idles between calls), so
  // the segments are laid end to end instead: lane c fetches the bounds of column c (one round trip for all columns), a wave scan
  // gives every segment its offset in the concatenated list, and the lanes then stride that list — one chain per 64 candidates.
  __shared__ int seg_off[kWinWPG][kWinCols + 1], seg_beg[kWinWPG][kWinCols];
  int* soff = seg_off[threadIdx.x >> 6];
  int* sbeg = seg_beg[threadIdx.x >> 6];
  const int ncol = any ? max(nMaxCellX - nMinCellX + 1, 0) : 0;   // <= 64
  int len = 0, beg = 0;
  if (lane < ncol) {
    const int ix = nMinCellX + lane;
    beg = a.cell_start[ix * kWinRows + nMinCellY];
    len = max(a.cell_start[ix * kWinRows + nMaxCellY + 1] - beg, 0);   // (an inverted row range, like the reference's empty loop)
  }
  int inc = len;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o); if (lane >= o) inc += u; }
  const int ntot = __shfl(inc, 63);
  if (lane < ncol) { soff[lane] = inc - len; sbeg[lane] = beg; }
  if (lane == 0) soff[ncol] = ntot;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  auto cell_slot = [&](int p) -> int {   // position p of the concatenated list -> index into cell_idx
    int lo = 0, hi = ncol - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (soff[mid] <= p) lo = mid; else hi = mid - 1;
    }
    return sbeg[lo] + (p - soff[lo]);
  };

  // keypoint index first; the candidate's descriptor is then requested TOGETHER with its keypoint (both hang on the index only), before
  // the gates are known: one link less in the chain of dependent loads, at the price of the descriptors of the gated-out candidates
  int count = 0, base = 0;
  const bool one_round = ntot <= 64;   // wave-uniform; the usual case: everything about the candidates stays in registers
  bool ok1 = false;
  int idx1 = 0;
  WDesc dt1;
  dt1.w[0] = dt1.w[1] = dt1.w[2] = dt1.w[3] = 0;
  if (LISTS) {   // how many, so that the query's candidates get one contiguous segment of the pool
    if (one_round) {
      if (lane < ntot) {
        idx1 = a.cell_idx[cell_slot(lane)];
        dt1 = wload(a.desc + (size_t)idx1 * 32);
        ok1 = passes(idx1);
      }
      count = __popcll(__ballot(ok1));
    } else {
      for (int p0 = 0; p0 < ntot; p0 += 64) {
        const int p = p0 + lane;
        const bool ok = p < ntot && passes(a.cell_idx[cell_slot(p)]);
        count += __popcll(__ballot(ok));
      }
    }
    int off = 0;
    if (lane == 0 && active && count) off = atomicAdd(&wg_cnt, count);   // LDS: this query's offset inside the workgroup's segment
    __syncthreads();
    if (threadIdx.x == 0) wg_base = wg_cnt ? atomicAdd(a.total, wg_cnt) : 0;
    __syncthreads();
    base = wg_base + __shfl(off, 0);
  }
  const bool write = LISTS && active && count && base + count <= a.pool_cap;
  WDesc dq;
#pragma unroll
  for (int w = 0; w < 4; w++)
    dq.w[w] = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane(rec, 8 + 2 * w) |   // (the builtin returns int: no sign extension)
              (unsigned long long)(uint32_t)__builtin_amdgcn_readlane(rec, 9 + 2 * w) << 32;
  unsigned long long k1 = kNoWinKey, k2 = kNoWinKey;
  int pos = 0;
  for (int p0 = 0; p0 < ntot; p0 += 64) {
    const int p = p0 + lane;
    int idx = idx1;
    bool ok = ok1;
    WDesc dt = dt1;
    if (!(LISTS && one_round)) {
      ok = false;
      if (p < ntot) {
        idx = a.cell_idx[cell_slot(p)];
        dt = wload(a.desc + (size_t)idx * 32);
        ok = passes(idx);
      }
    }
    const unsigned long long bal = __ballot(ok);
    if (ok) {
      const int my = pos + __popcll(bal & ((1ull << lane) - 1ull));
      const int d = wham(dq, dt);
      const unsigned long long key = ((unsigned long long)d << 48) | ((unsigned long long)my << 24) | (unsigned long long)idx;
      if (key < k1) { k2 = k1; k1 = key; }
      else if (key < k2) k2 = key;
      if (write) a.pool[base + my] = make_int2(idx, d);
    }
    pos += __popcll(bal);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long p1 = __shfl_xor(k1, o), p2 = __shfl_xor(k2, o);
    const unsigned long long lo = k1 < p1 ? k1 : p1, hi = k1 < p1 ? p1 : k1;
    const unsigned long long s2 = k2 < p2 ? k2 : p2;
    k1 = lo;
    k2 = hi < s2 ? hi : s2;
  }
  if (lane == 0 && active) {
    if (a.compact) {
      WinQueryShort o;
      o.start = base; o.count = LISTS ? count : pos;
      ((WinQueryShort*)a.out)[q] = o;
    } else {
      WinQueryOut o;
      o.start = base; o.count = LISTS ? count : pos;
      o.best_idx = k1 == kNoWinKey ? -1 : (int)(k1 & 0xffffffu);
      o.best_dist = k1 == kNoWinKey ? 256 : (int)(k1 >> 48);
      o.second_idx = k2 == kNoWinKey ? -1 : (int)(k2 & 0xffffffu);
      o.second_dist = k2 == kNoWinKey ? 256 : (int)(k2 >> 48);
      o.pad0 = o.pad1 = 0;
      a.out[q] = o;
    }
    // the query that finishes last publishes the total (its own atomicAdd on the total is ordered before this counter).  The
    // records and the pool may live in mapped host memory (a.host_out): every wave makes its stores visible to the host before
    // it counts itself, and the last one sets the done flag in the same 8-byte store as the total — the host polls that word
    // instead of waiting for a copy and a stream synchronisation.
    // (a system- or agent-scope release writes the L2 back and is far too expensive to do per query: the waves of a workgroup
    // order their stores at workgroup scope around the LDS counter, and only the last of them — its L2 is the one they all
    // wrote to — pays for the wide fence)
    // a workgroup-scope release emits no vmcnt wait on gfx9 outside tgsplit mode: drain this wave's own stores (records, pool entries)
    // before it counts itself — the last wave's wide fence below only waits for ITS outstanding stores
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    const bool wg_last = atomicAdd(&wg_done, 1) == nactive - 1;
    if (wg_last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      if (a.host_out) __threadfence_system(); else __threadfence();
    }
    if (wg_last && atomicAdd(a.total + 1, 1) == (int)gridDim.x - 1) {
      __threadfence();
      const unsigned long long hdr = (unsigned long long)(uint32_t)atomicAdd(a.total, 0) | 1ull << 32;
      if (a.self_reset) { a.total[0] = 0; a.total[1] = 0; __threadfence(); }   // every other wave has left the counters
      if (a.host_out) { __atomic_store_n((unsigned long long*)a.out_hdr, hdr, __ATOMIC_RELEASE); __threadfence_system(); }
      else *(unsigned long long*)a.out_hdr = hdr;
    }
  }
}

// Frame::AssignFeaturesToGrid on the device (src/Frame.cc:385-416, PosInGrid :725-735) for grids the caller does not hold:
// cell of every keypoint, then an LDS bitonic sort of (cell << 16 | index); ascending keys list the cells in mGrid[ix][iy]
// order and, inside a cell, the keypoints in insertion order.  One workgroup, at most 32 768 keypoints.
__global__ __launch_bounds__(1024) void k_window_grid(const orbx_keypoint* __restrict__ kps, int n, float minX, float minY, float invW,
                                                      float invH, int npad, int32_t* __restrict__ cell_idx, int32_t* __restrict__ cell_start) {
  extern __shared__ uint32_t wkeys[];
  const int t = threadIdx.x, T = blockDim.x;
  for (int i = t; i < npad; i += T) {
    uint32_t key = 0xffffffffu;
    if (i < n) {
      const int posX = (int)roundf(__fmul_rn(__fsub_rn(kps[i].x, minX), invW));
      const int posY = (int)roundf(__fmul_rn(__fsub_rn(kps[i].y, minY), invH));
      const bool in = posX >= 0 && posX < kWinCols && posY >= 0 && posY < kWinRows;
      key = ((in ? (uint32_t)(posX * kWinRows + posY) : 0xfffeu) << 16) | (uint32_t)i;
    }
    wkeys[i] = key;
  }
  __syncthreads();
  for (int k = 2; k <= npad; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = t; i < npad; i += T) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const uint32_t a = wkeys[i], b = wkeys[ixj];
          const bool up = (i & k) == 0;
          if ((a > b) == up) { wkeys[i] = b; wkeys[ixj] = a; }
        }
      }
      __syncthreads();
    }
  }
  for (int i = t; i < n; i += T) cell_idx[i] = (int32_t)(wkeys[i] & 0xffffu);
  for (int c = t; c <= kWinCells; c += T) {
    const uint32_t want = (uint32_t)c << 16;
    int lo = 0, hi = n;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (wkeys[mid] < want) lo = mid + 1; else hi = mid;
    }
    cell_start[c] = lo;
  }
}

// Frame::UndistortKeyPoints (src/Frame.cc:747-780): cv::undistortPoints(pts, pts, K, distCoef, Mat(), K) on the keypoint
// coordinates, OpenCV 4.x's iteration in double with separately rounded operations (5 iterations, radial k1 k2 k3 +
// tangential p1 p2; no tilt, R = I, P = K).  One thread per keypoint; everything but pt is copied.
struct UndistortParams { double fx, fy, cx, cy, ifx, ify, k[14]; };
__global__ __launch_bounds__(256) void k_undistort(const orbx_keypoint* __restrict__ in, const int32_t* __restrict__ counts, int cap, int n_fixed,
                                                   UndistortParams P, orbx_keypoint* __restrict__ out) {
  const int f = blockIdx.y;
  const int n = counts ? counts[2 * f] : n_fixed;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  orbx_keypoint kp = in[(size_t)f * cap + i];
  double x = (double)kp.x, y = (double)kp.y;
  const double u = x, v = y;
  x = __dmul_rn(__dsub_rn(x, P.cx), P.ifx);
  y = __dmul_rn(__dsub_rn(y, P.cy), P.ify);
  const double x0 = x, y0 = y;
  const double* k = P.k;
  for (int j = 0; j < 5; j++) {
    const double r2 = __dadd_rn(__dmul_rn(x, x), __dmul_rn(y, y));
    const double num = __dadd_rn(1.0, __dmul_rn(__dadd_rn(__dmul_rn(__dadd_rn(__dmul_rn(k[7], r2), k[6]), r2), k[5]), r2));
    const double den = __dadd_rn(1.0, __dmul_rn(__dadd_rn(__dmul_rn(__dadd_rn(__dmul_rn(k[4], r2), k[1]), r2), k[0]), r2));
    const double icdist = __ddiv_rn(num, den);
    if (icdist < 0) { x = __dmul_rn(__dsub_rn(u, P.cx), P.ifx); y = __dmul_rn(__dsub_rn(v, P.cy), P.ify); break; }
    // 2*k[2]*x*y + k[3]*(r2 + 2*x*x) + k[8]*r2 + k[9]*r2*r2, left to right
    const double dX = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(__dmul_rn(__dmul_rn(2.0, k[2]), x), y),
                                                    __dmul_rn(k[3], __dadd_rn(r2, __dmul_rn(__dmul_rn(2.0, x), x)))),
                                          __dmul_rn(k[8], r2)),
                                __dmul_rn(__dmul_rn(k[9], r2), r2));
    // k[2]*(r2 + 2*y*y) + 2*k[3]*x*y + k[10]*r2 + k[11]*r2*r2
    const double dY = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(k[2], __dadd_rn(r2, __dmul_rn(__dmul_rn(2.0, y), y))),
                                                    __dmul_rn(__dmul_rn(__dmul_rn(2.0, k[3]), x), y)),
                                          __dmul_rn(k[10], r2)),
                                __dmul_rn(__dmul_rn(k[11], r2), r2));
    x = __dmul_rn(__dsub_rn(x0, dX), icdist);
    y = __dmul_rn(__dsub_rn(y0, dY), icdist);
  }
  // (xx, yy, ww) = K * (x, y, 1): RR[0][0]*x + RR[0][1]*y + RR[0][2] with RR[0][1] = 0 etc.
  const double xx = __dadd_rn(__dadd_rn(__dmul_rn(P.fx, x), __dmul_rn(0.0, y)), P.cx);
  const double yy = __dadd_rn(__dadd_rn(__dmul_rn(0.0, x), __dmul_rn(P.fy, y)), P.cy);
  const double ww = __ddiv_rn(1.0, __dadd_rn(__dadd_rn(__dmul_rn(0.0, x), __dmul_rn(0.0, y)), 1.0));
  kp.x = (float)__dmul_rn(xx, ww);
  kp.y = (float)__dmul_rn(yy, ww);
  out[(size_t)f * cap + i] = kp;
}

static bool undistort_params(float fx, float fy, float cx, float cy, const float* dist, int ndist, UndistortParams& P) {
  std::memset(&P, 0, sizeof(P));
  P.fx = fx; P.fy = fy; P.cx = cx; P.cy = cy;
  P.ifx = 1. / P.fx; P.ify = 1. / P.fy;
  for (int i = 0; i < ndist && i < 14; i++) P.k[i] = (double)dist[i];
  return ndist > 0 && dist[0] != 0.0f;
}

// pinned host staging of the host-buffer entry points, grow-only
hipError_t host_stage(orbx_ctx* ctx, size_t bytes, uint8_t** p) {
  if (bytes > ctx->h_call_bytes) {
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return e;
    if (ctx->h_call) (void)hipHostFree(ctx->h_call);
    ctx->h_call = nullptr; ctx->h_call_bytes = 0;
    const size_t want = std::max<size_t>(bytes + bytes / 2, 1 << 20);
    e = hipHostMalloc((void**)&ctx->h_call, want, hipHostMallocMapped | hipHostMallocCoherent);   // kernels read and write it directly
    if (e != hipSuccess) { (void)hipGetLastError(); e = hipHostMalloc((void**)&ctx->h_call, want, hipHostMallocDefault); }
    if (e != hipSuccess) return e;
    ctx->h_call_bytes = want;
  }
  *p = ctx->h_call;
  return hipSuccess;
}

// The blob of a VIEW call (orbx_target_search_view: the caller reads the lists in place): two of them, used alternately, so that a view stays
// valid while the NEXT view call of the context runs — a two-camera rig searches the left and the right frame back to back and replays
// both lists together.  The blob is chosen ONCE per orbx_target_search_view call (ctx->view_par, flipped there): the capacity retry of that
// call re-uses — and, when it must, re-allocates — the SAME blob, never the one that still backs the previous view.
static hipError_t host_stage_view(orbx_ctx* ctx, size_t bytes, uint8_t** p) {
  const int i = ctx->view_par;
  if (bytes > ctx->h_view_bytes[i]) {
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return e;
    if (ctx->h_view[i]) (void)hipHostFree(ctx->h_view[i]);
    ctx->h_view[i] = nullptr; ctx->h_view_bytes[i] = 0;
    const size_t want = std::max<size_t>(bytes + bytes / 2, 1 << 20);
    e = hipHostMalloc((void**)&ctx->h_view[i], want, hipHostMallocMapped | hipHostMallocCoherent);
    if (e != hipSuccess) { (void)hipGetLastError(); e = hipHostMalloc((void**)&ctx->h_view[i], want, hipHostMallocDefault); }
    if (e != hipSuccess) return e;
    ctx->h_view_bytes[i] = want;
  }
  *p = ctx->h_view[i];
  return hipSuccess;
}

typedef BlobLayout Layout;

static void pack_queries(uint8_t* dst, const float* qx, const float* qy, const float* qr, const float* qaux, const int32_t* qlo, const int32_t* qhi,
                         const uint8_t* q_desc, int nq) {
  WinQueryIn* o = (WinQueryIn*)dst;
  for (int q = 0; q < nq; q++) {
    o[q].x = qx[q]; o[q].y = qy[q]; o[q].r = qr[q]; o[q].aux = qaux ? qaux[q] : 0.f;
    o[q].lo = qlo[q]; o[q].hi = qhi[q]; o[q].pad0 = o[q].pad1 = 0;
    std::memcpy(o[q].desc, q_desc + (size_t)q * 32, 32);
  }
}

// Input blob from mapped pinned host memory into HBM, 16 bytes per lane (the window kernel gathers keypoints and descriptors at
// random: those reads must not cross PCIe one by one).  One launch costs less than a hipMemcpyAsync of the same 100 KB.
__global__ __launch_bounds__(256) void k_stage_in(const uint4* __restrict__ src, uint4* __restrict__ dst, int n16) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) dst[i] = src[i];
}
// The whole call: pack -> one H2D -> [grid assignment] -> k_window -> one D2H (+ one more for a long tail) -> scatter.
int window_call(orbx_ctx* ctx, const char* who, const orbx_keypoint* kps, const uint8_t* desc, int n, const orbx_grid* grid,
                const uint8_t* kp_skip, const float* kp_uright, const float* inv_sigma2, int nlevels, const float* qx, const float* qy,
                const float* qr, const int32_t* qlo, const int32_t* qhi, const float* qaux, const uint8_t* q_desc, int nq, bool lists,
                int32_t* row_ptr, int32_t* cand, int32_t* dist, int cand_cap, int32_t* best_idx, int32_t* best_dist, int32_t* second_idx,
                int32_t* second_dist) {
  if (row_ptr) for (int q = 0; q <= nq; q++) row_ptr[q] = 0;
  for (int q = 0; q < nq; q++) {
    if (best_idx) best_idx[q] = -1;
    if (best_dist) best_dist[q] = 256;
    if (second_idx) second_idx[q] = -1;
    if (second_dist) second_dist[q] = 256;
  }
  if (nq == 0 || n == 0) return 0;
  if (n >= (1 << 24)) return set_err(ctx, ORBX_E_CAPACITY, std::string(who) + ": more than 16 M keypoints");
  const bool have_grid = grid->cell_start != nullptr;
  if (have_grid) {
    if (!grid->cell_idx) return set_err(ctx, ORBX_E_INVALID, std::string(who) + ": grid without cell_idx");
    if (grid->cell_start[0] != 0) return set_err(ctx, ORBX_E_INVALID, std::string(who) + ": grid cell_start[0] != 0");
    for (int c = 0; c < kWinCells; c++)
      if (grid->cell_start[c + 1] < grid->cell_start[c]) return set_err(ctx, ORBX_E_INVALID, std::string(who) + ": grid cell_start not ascending");
    const int m = grid->cell_start[kWinCells];
    for (int i = 0; i < m; i++)
      if (grid->cell_idx[i] < 0 || grid->cell_idx[i] >= n) return set_err(ctx, ORBX_E_INVALID, std::string(who) + ": grid index out of range");
  } else if (n > 32768) {
    return set_err(ctx, ORBX_E_CAPACITY, std::string(who) + ": more than 32768 keypoints need a caller-held grid");
  }
  if (inv_sigma2)
    for (int i = 0; i < n; i++)
      if (kps[i].octave < 0 || kps[i].octave >= nlevels) return set_err(ctx, ORBX_E_INVALID, std::string(who) + ": keypoint octave outside inv_level_sigma2");
  static const bool trace = getenv("ORBX_TRACE_WINDOW") != nullptr;   // phase times of every call on stderr (diagnostics)
  const auto tr0 = std::chrono::steady_clock::now();
  auto since = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count(); };
  ORBX_HIP(ctx, hipSetDevice(ctx->device));
  const int ngrid = have_grid ? grid->cell_start[kWinCells] : n;
  const int pool_cap = lists ? std::max(cand_cap, 0) : 0;
  // ---- layouts: input blob (host -> device) and output blob (device -> host)
  Layout in;
  const size_t o_total = in.add(16);
  const size_t o_kps = in.add(sizeof(orbx_keypoint) * (size_t)n), o_desc = in.add((size_t)n * 32);
  const size_t o_q = in.add(sizeof(WinQueryIn) * (size_t)nq);
  const size_t o_skip = in.add(kp_skip ? (size_t)n : 0), o_ur = in.add(kp_uright ? 4 * (size_t)n : 0);
  const size_t o_sig = in.add(inv_sigma2 ? 4 * (size_t)nlevels : 0);
  const size_t o_cs = in.add(4 * (size_t)(kWinCells + 1)), o_ci = in.add(4 * (size_t)std::max(ngrid, 1));
  const size_t in_upload = have_grid ? in.size : o_cs;   // without a caller grid the two grid arrays are produced on the device
  Layout out;
  const bool compact = !(best_idx || best_dist || second_idx || second_dist);
  const size_t qrec = compact ? sizeof(WinQueryShort) : sizeof(WinQueryOut);
  const size_t p_hdr = out.add(16), p_q = out.add(qrec * (size_t)nq), p_pool = out.add(8 * (size_t)pool_cap);
  uint8_t* h = nullptr;
  ORBX_HIP(ctx, host_stage(ctx, in.size + out.size, &h));
  uint8_t* hin = h;
  uint8_t* hout = h + in.size;
  std::memset(hin + o_total, 0, 16);
  std::memcpy(hin + o_kps, kps, sizeof(orbx_keypoint) * (size_t)n);
  std::memcpy(hin + o_desc, desc, (size_t)n * 32);
  pack_queries(hin + o_q, qx, qy, qr, qaux, qlo, qhi, q_desc, nq);
  if (kp_skip) std::memcpy(hin + o_skip, kp_skip, (size_t)n);
  if (kp_uright) std::memcpy(hin + o_ur, kp_uright, 4 * (size_t)n);
  if (inv_sigma2) std::memcpy(hin + o_sig, inv_sigma2, 4 * (size_t)nlevels);
  if (have_grid) {
    std::memcpy(hin + o_cs, grid->cell_start, 4 * (size_t)(kWinCells + 1));
    if (ngrid) std::memcpy(hin + o_ci, grid->cell_idx, 4 * (size_t)ngrid);
  }
  const double us_pack = since(tr0);
  ctx->arena.rewind();
  hipError_t aerr = hipSuccess;
  uint8_t* din = (uint8_t*)ctx->arena.alloc(in.size, &aerr);
  ORBX_HIP(ctx, aerr);
  uint8_t* dout = (uint8_t*)ctx->arena.alloc(out.size, &aerr);
  ORBX_HIP(ctx, aerr);
  hipStream_t st = ctx->stream;
  // direct mode (default): the kernels read the input blob from, and write the results to, the mapped pinned blob; the host
  // polls the done word.  ORBX_WINDOW_DIRECT=0 selects copies + stream synchronisation (same results; A/B and fallback).
  const bool direct_wanted = ctx->window_direct;
  uint8_t* hdev = nullptr;
  const bool direct = direct_wanted && hipHostGetDevicePointer((void**)&hdev, h, 0) == hipSuccess && hdev != nullptr;
  if (!direct) (void)hipGetLastError();
  std::memset(hout + p_hdr, 0, 16);
  if (!have_grid && n > 16384) {   // npad * 4 > 64 KB: checked BEFORE anything is queued on the stream
    if (ensure_dynamic_lds((const void*)k_window_grid, 32768 * 4) != hipSuccess)
      return set_err(ctx, ORBX_E_CAPACITY, std::string(who) + ": LDS for the grid sort unavailable");
  }
  if (direct) {
    const int n16 = (int)((in_upload + 15) / 16);
    hipLaunchKernelGGL(k_stage_in, dim3(std::min((n16 + 255) / 256, 256)), dim3(256), 0, st, (const uint4*)hdev, (uint4*)din, n16);
  } else {
    ORBX_HIP(ctx, hipMemcpyAsync(din, hin, in_upload, hipMemcpyHostToDevice, st));
  }
  if (!have_grid) {
    int npad = 2;
    while (npad < n) npad <<= 1;
    hipLaunchKernelGGL(k_window_grid, dim3(1), dim3(1024), (size_t)npad * 4, st, (const orbx_keypoint*)(din + o_kps), n, grid->min_x, grid->min_y,
                       grid->inv_w, grid->inv_h, npad, (int32_t*)(din + o_ci), (int32_t*)(din + o_cs));
  }
  WinArgs a;
  std::memset(&a, 0, sizeof(a));
  a.kps = (const orbx_keypoint*)(din + o_kps); a.desc = din + o_desc;
  a.cell_start = (const int32_t*)(din + o_cs); a.cell_idx = (const int32_t*)(din + o_ci);
  a.minX = grid->min_x; a.minY = grid->min_y; a.invW = grid->inv_w; a.invH = grid->inv_h;
  a.qin = (const WinQueryIn*)(din + o_q); a.has_aux = qaux ? 1 : 0; a.nq = nq;
  a.kp_skip = kp_skip ? din + o_skip : nullptr;
  a.kp_uright = kp_uright ? (const float*)(din + o_ur) : nullptr;
  a.inv_sigma2 = inv_sigma2 ? (const float*)(din + o_sig) : nullptr;
  uint8_t* const res = direct ? hdev + in.size : dout;
  a.out = (WinQueryOut*)(res + p_q); a.pool = (int2*)(res + p_pool); a.pool_cap = pool_cap; a.total = (int32_t*)(din + o_total);
  a.out_hdr = (int32_t*)(res + p_hdr); a.compact = compact ? 1 : 0; a.host_out = direct ? 1 : 0;
  const dim3 gridDim((nq + kWinWPG - 1) / kWinWPG), block(64 * kWinWPG);
  if (inv_sigma2) hipLaunchKernelGGL((k_window<false, true>), gridDim, block, 0, st, a);
  else if (lists) hipLaunchKernelGGL((k_window<true, false>), gridDim, block, 0, st, a);
  else hipLaunchKernelGGL((k_window<false, false>), gridDim, block, 0, st, a);
  ORBX_HIP(ctx, hipGetLastError());
  // ---- results: header + per-query records + the head of the pool in one copy; a long tail in a second one
  const int guess = direct ? pool_cap : std::min(pool_cap, ctx->win_guess > 0 ? ctx->win_guess : 2048);
  double us_issue;
  if (direct) {
    us_issue = since(tr0);
    // poll the done word; look at the stream now and then so that a failed launch or a device fault ends the wait
    const volatile unsigned long long* hdr = (const volatile unsigned long long*)(hout + p_hdr);
    for (unsigned spin = 1;; spin++) {
      if (__atomic_load_n(hdr, __ATOMIC_ACQUIRE) >> 32) break;
      if ((spin & 0x3fff) == 0) {
        const hipError_t qe = hipStreamQuery(st);
        if (qe == hipSuccess) {   // everything retired: the word must be there now
          if (__atomic_load_n(hdr, __ATOMIC_ACQUIRE) >> 32) break;
          return set_err(ctx, ORBX_E_DEVICE, std::string(who) + ": window pass finished without publishing its results");
        }
        if (qe != hipErrorNotReady) { ORBX_HIP(ctx, qe); }
      }
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
    }
  } else {
    ORBX_HIP(ctx, hipMemcpyAsync(hout, dout, p_pool + 8 * (size_t)guess, hipMemcpyDeviceToHost, st));
    us_issue = since(tr0);
    ORBX_HIP(ctx, hipStreamSynchronize(st));
  }
  const double us_sync = since(tr0);
  const int total = *(const int32_t*)(hout + p_hdr);
  const WinQueryOut* qo = (const WinQueryOut*)(hout + p_q);
  const WinQueryShort* qs = (const WinQueryShort*)(hout + p_q);
  auto q_start = [&](int q) { return compact ? qs[q].start : qo[q].start; };
  auto q_count = [&](int q) { return compact ? qs[q].count : qo[q].count; };
  if (lists) {
    ctx->win_guess = total + total / 4 + 256;
    int acc = 0;
    for (int q = 0; q < nq; q++) { acc += q_count(q); row_ptr[q + 1] = acc; }
    if (total > pool_cap) return set_err(ctx, ORBX_E_CAPACITY, std::string(who) + ": candidate buffer too small");
    if (total > guess) {
      ORBX_HIP(ctx, hipMemcpyAsync(hout + p_pool + 8 * (size_t)guess, dout + p_pool + 8 * (size_t)guess, 8 * (size_t)(total - guess),
                                   hipMemcpyDeviceToHost, st));
      ORBX_HIP(ctx, hipStreamSynchronize(st));
    }
    const int2* pool = (const int2*)(hout + p_pool);
    for (int q = 0; q < nq; q++) {
      const int2* seg = pool + q_start(q);
      const int o = row_ptr[q];
      const int cnt = q_count(q);
      for (int c = 0; c < cnt; c++) {
        if (cand) cand[o + c] = seg[c].x;
        if (dist) dist[o + c] = seg[c].y;
      }
    }
  } else if (row_ptr) {
    int acc = 0;
    for (int q = 0; q < nq; q++) { acc += q_count(q); row_ptr[q + 1] = acc; }
  }
  if (!compact) for (int q = 0; q < nq; q++) {
    if (best_idx) best_idx[q] = qo[q].best_idx;
    if (best_dist) best_dist[q] = qo[q].best_dist;
    if (second_idx) second_idx[q] = qo[q].second_idx;
    if (second_dist) second_dist[q] = qo[q].second_dist;
  }
  if (trace)
    std::fprintf(stderr, "[orbx window] %s n=%d nq=%d total=%d in=%zu B out=%zu B: pack %.1f us, issue %.1f, wait %.1f, scatter %.1f\n", who, n, nq, total,
                 in_upload, p_pool + 8 * (size_t)guess, us_pack, us_issue - us_pack, us_sync - us_issue, since(tr0) - us_sync);
  return lists ? total : (row_ptr ? row_ptr[nq] : 0);
}

// ---- resident targets ---------------------------------------------------------------------------------------------------------
// A Frame / KeyFrame is searched many times (Tracking: two or three calls per frame; LocalMapping / LoopClosing: every keyframe
// against many others) while its keypoints, descriptors and grid never change after construction.  orbx_target_create uploads
// them once; a search then moves only the queries, and — with everything it gathers at random already in HBM — needs ONE kernel:
// the queries are read straight from the mapped pinned blob, the per-call kp_skip flags are staged into LDS by every workgroup,
// the results are written into the mapped blob and the host polls the done word.
static int validate_grid(orbx_ctx* ctx, const char* who, const orbx_grid* grid, int n) {
  if (!grid->cell_idx) return set_err(ctx, ORBX_E_INVALID, std::string(who) + ": grid without cell_idx");
  if (grid->cell_start[0] != 0) return set_err(ctx, ORBX_E_INVALID, std::string(who) + ": grid cell_start[0] != 0");
  for (int c = 0; c < kWinCells; c++)
    if (grid->cell_start[c + 1] < grid->cell_start[c]) return set_err(ctx, ORBX_E_INVALID, std::string(who) + ": grid cell_start not ascending");
  const int m = grid->cell_start[kWinCells];
  for (int i = 0; i < m; i++)
    if (grid->cell_idx[i] < 0 || grid->cell_idx[i] >= n) return set_err(ctx, ORBX_E_INVALID, std::string(who) + ": grid index out of range");
  return ORBX_OK;
}

// (re)fills a target: `reuse` keeps its device block when that is large enough (the adapters' LRU of frames recycles targets, so that
// the steady state allocates nothing)
int target_assign(orbx_ctx* ctx, orbx_target* reuse, const orbx_keypoint* kps, const uint8_t* desc, int n, const orbx_grid* grid,
                  const float* kp_uright, const float* inv_sigma2, int nlevels, orbx_target** out) {
  const char* who = reuse ? "orbx_target_assign" : "orbx_target_create";
  if (out) *out = nullptr;
  if (reuse) reuse->valid = false;   // until this refill has succeeded: a failed assign must not leave a target that looks searchable
  if (n >= (1 << 24)) return set_err(ctx, ORBX_E_CAPACITY, std::string(who) + ": more than 16 M keypoints");
  const bool have_grid = grid->cell_start != nullptr;
  if (have_grid) { const int rc = validate_grid(ctx, who, grid, n); if (rc != ORBX_OK) return rc; }
  else if (n > 32768) return set_err(ctx, ORBX_E_CAPACITY, std::string(who) + ": more than 32768 keypoints need a caller-held grid");
  if (inv_sigma2)
    for (int i = 0; i < n; i++)
      if (kps[i].octave < 0 || kps[i].octave >= nlevels) return set_err(ctx, ORBX_E_INVALID, std::string(who) + ": keypoint octave outside inv_level_sigma2");
  ORBX_HIP(ctx, hipSetDevice(ctx->device));
  orbx_target* T = reuse ? reuse : new orbx_target();
  T->ctx = ctx; T->n = n; T->nlevels = inv_sigma2 ? nlevels : 0;
  T->has_ur = kp_uright != nullptr; T->has_sig = inv_sigma2 != nullptr;
  T->min_x = grid->min_x; T->min_y = grid->min_y; T->inv_w = grid->inv_w; T->inv_h = grid->inv_h;
  T->ngrid = have_grid ? grid->cell_start[kWinCells] : n;
  Layout L;
  T->o_kps = L.add(sizeof(orbx_keypoint) * (size_t)std::max(n, 1)); T->o_desc = L.add((size_t)std::max(n, 1) * 32);
  T->o_ur = L.add(kp_uright ? 4 * (size_t)n : 0); T->o_sig = L.add(inv_sigma2 ? 4 * (size_t)nlevels : 0);
  T->o_cs = L.add(4 * (size_t)(kWinCells + 1)); T->o_ci = L.add(4 * (size_t)std::max(T->ngrid, 1));
  T->bytes = L.size;
  const size_t upload = have_grid ? L.size : T->o_cs;
  // the upload is staged in its own pinned buffer (not the per-call blob: the search that follows packs its queries there while
  // the copy kernel may still be reading) and is asynchronous: the stream orders every later search behind it
  uint8_t* h = nullptr;
  hipError_t e = hipSuccess;
  if (ctx->ev_tgt) e = hipEventSynchronize(ctx->ev_tgt);   // the previous upload has left the staging buffer (it almost always has)
  if (e == hipSuccess && L.size > ctx->h_tgt_bytes) {
    if (ctx->h_tgt) (void)hipHostFree(ctx->h_tgt);
    ctx->h_tgt = nullptr; ctx->h_tgt_bytes = 0;
    const size_t want = std::max<size_t>(L.size + L.size / 2, 256 << 10);
    e = hipHostMalloc((void**)&ctx->h_tgt, want, hipHostMallocMapped | hipHostMallocCoherent);
    if (e != hipSuccess) { (void)hipGetLastError(); e = hipHostMalloc((void**)&ctx->h_tgt, want, hipHostMallocDefault); }
    if (e == hipSuccess) ctx->h_tgt_bytes = want;
  }
  if (e == hipSuccess && !ctx->ev_tgt) e = hipEventCreateWithFlags(&ctx->ev_tgt, hipEventDisableTiming);
  h = ctx->h_tgt;
  if (e == hipSuccess && L.size > T->cap) {
    if (T->dev) { (void)hipFree(T->dev); T->dev = nullptr; T->cap = 0; }
    const size_t want = (std::max<size_t>(L.size, 192 << 10) + 0xffff) & ~(size_t)0xffff;
    e = hipMalloc((void**)&T->dev, want);
    if (e == hipSuccess) T->cap = want;
  }
  if (e != hipSuccess) { T->n = 0; if (!reuse) { if (T->dev) (void)hipFree(T->dev); delete T; } ORBX_HIP(ctx, e); }
  if (n) { std::memcpy(h + T->o_kps, kps, sizeof(orbx_keypoint) * (size_t)n); std::memcpy(h + T->o_desc, desc, (size_t)n * 32); }
  if (kp_uright) std::memcpy(h + T->o_ur, kp_uright, 4 * (size_t)n);
  if (inv_sigma2) std::memcpy(h + T->o_sig, inv_sigma2, 4 * (size_t)nlevels);
  if (have_grid) {
    std::memcpy(h + T->o_cs, grid->cell_start, 4 * (size_t)(kWinCells + 1));
    if (T->ngrid) std::memcpy(h + T->o_ci, grid->cell_idx, 4 * (size_t)T->ngrid);
  }
  hipStream_t st = ctx->stream;
  uint8_t* hdev = nullptr;
  if (ctx->window_direct && hipHostGetDevicePointer((void**)&hdev, h, 0) == hipSuccess && hdev) {
    const int n16 = (int)((upload + 15) / 16);   // the block is a multiple of 256 bytes
    hipLaunchKernelGGL(k_stage_in, dim3(std::min((n16 + 255) / 256, 256)), dim3(256), 0, st, (const uint4*)hdev, (uint4*)T->dev, n16);
    e = hipGetLastError();
  } else {
    (void)hipGetLastError();
    e = hipMemcpyAsync(T->dev, h, upload, hipMemcpyHostToDevice, st);
  }
  if (e == hipSuccess && !have_grid) {
    if (n == 0) {
      e = hipMemsetAsync(T->dev + T->o_cs, 0, 4 * (size_t)(kWinCells + 1), st);
    } else {
      int npad = 2;
      while (npad < n) npad <<= 1;
      if ((size_t)npad * 4 > 64 * 1024) e = ensure_dynamic_lds((const void*)k_window_grid, 32768 * 4);
      if (e == hipSuccess) {
        hipLaunchKernelGGL(k_window_grid, dim3(1), dim3(1024), (size_t)npad * 4, st, (const orbx_keypoint*)(T->dev + T->o_kps), n, grid->min_x,
                           grid->min_y, grid->inv_w, grid->inv_h, npad, (int32_t*)(T->dev + T->o_ci), (int32_t*)(T->dev + T->o_cs));
        e = hipGetLastError();
      }
    }
  }
  if (e == hipSuccess) e = hipEventRecord(ctx->ev_tgt, st);
  if (e != hipSuccess) { T->n = 0; if (!reuse) { (void)hipFree(T->dev); delete T; } ORBX_HIP(ctx, e); }
  T->valid = true;
  if (out) *out = T;
  return ORBX_OK;
}

void target_destroy(orbx_target* T) {
  if (!T) return;
  if (T->dev) { (void)hipSetDevice(T->ctx->device); (void)hipStreamSynchronize(T->ctx->stream); (void)hipFree(T->dev); }
  delete T;
}

int window_call_target(orbx_ctx* ctx, const char* who, const orbx_target* T, const uint8_t* kp_skip, bool chi2, const float* qx, const float* qy,
                       const float* qr, const int32_t* qlo, const int32_t* qhi, const float* qaux, const uint8_t* q_desc, int nq, bool lists,
                       int32_t* row_ptr, int32_t* cand, int32_t* dist, int cand_cap, int32_t* best_idx, int32_t* best_dist, int32_t* second_idx,
                       int32_t* second_dist, const orbx_list_span** v_spans = nullptr, const orbx_candidate** v_pool = nullptr) {
  if (v_spans) *v_spans = nullptr;
  if (v_pool) *v_pool = nullptr;
  if (row_ptr) for (int q = 0; q <= nq; q++) row_ptr[q] = 0;
  for (int q = 0; q < nq; q++) {
    if (best_idx) best_idx[q] = -1;
    if (best_dist) best_dist[q] = 256;
    if (second_idx) second_idx[q] = -1;
    if (second_dist) second_dist[q] = 256;
  }
  if (!T->valid) return set_err(ctx, ORBX_E_INVALID, std::string(who) + ": the target's last orbx_target_assign failed; assign it again");
  const int n = T->n;
  if (nq == 0 || n == 0) return 0;
  if (chi2 && !(T->has_sig && T->has_ur)) return set_err(ctx, ORBX_E_INVALID, std::string(who) + ": the target holds no kp_uright / inv_level_sigma2");
  static const bool trace = getenv("ORBX_TRACE_WINDOW") != nullptr;
  const auto tr0 = std::chrono::steady_clock::now();
  auto since = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count(); };
  ORBX_HIP(ctx, hipSetDevice(ctx->device));
  const int pool_cap = lists ? std::max(cand_cap, 0) : 0;
  const int nskip = kp_skip ? (n + 3) & ~3 : 0;
  Layout in;
  const size_t o_q = in.add(sizeof(WinQueryIn) * (size_t)nq), o_skip = in.add((size_t)nskip);
  Layout out;
  const bool compact = !(best_idx || best_dist || second_idx || second_dist) && !v_spans;   // a view hands the full records to the caller
  const size_t qrec = compact ? sizeof(WinQueryShort) : sizeof(WinQueryOut);
  const size_t p_hdr = out.add(16), p_q = out.add(qrec * (size_t)nq), p_pool = out.add(8 * (size_t)pool_cap);
  uint8_t* h = nullptr;
  if (v_spans) { ORBX_HIP(ctx, host_stage_view(ctx, in.size + out.size, &h)); }
  else { ORBX_HIP(ctx, host_stage(ctx, in.size + out.size, &h)); }
  uint8_t* hin = h;
  uint8_t* hout = h + in.size;
  pack_queries(hin + o_q, qx, qy, qr, qaux, qlo, qhi, q_desc, nq);
  if (kp_skip) { std::memcpy(hin + o_skip, kp_skip, (size_t)n); std::memset(hin + o_skip + n, 0, (size_t)(nskip - n)); }
  std::memset(hout + p_hdr, 0, 16);
  const double us_pack = since(tr0);
  hipStream_t st = ctx->stream;
  if (!ctx->d_win_ctr) {
    ORBX_HIP(ctx, hipMalloc((void**)&ctx->d_win_ctr, 64));
    ctx->win_ctr_dirty = true;
  }
  if (ctx->win_ctr_dirty) {   // first use, or a call that did not finish: the self-resetting counters start from zero
    ORBX_HIP(ctx, hipMemsetAsync(ctx->d_win_ctr, 0, 64, st));
    ctx->win_ctr_dirty = false;
  }
  const bool direct_wanted = ctx->window_direct;
  uint8_t* hdev = nullptr;
  const bool direct = direct_wanted && nskip <= 48 * 1024 && hipHostGetDevicePointer((void**)&hdev, h, 0) == hipSuccess && hdev != nullptr;
  if (!direct) (void)hipGetLastError();
  uint8_t* qbase = hdev;   // where the kernel reads the per-call inputs
  uint8_t* dout = nullptr;
  if (!direct) {           // copies + stream synchronisation (ORBX_WINDOW_DIRECT=0, or a target too large for the LDS flags)
    ctx->arena.rewind();
    hipError_t aerr = hipSuccess;
    qbase = (uint8_t*)ctx->arena.alloc(in.size, &aerr);
    ORBX_HIP(ctx, aerr);
    dout = (uint8_t*)ctx->arena.alloc(out.size, &aerr);
    ORBX_HIP(ctx, aerr);
    ORBX_HIP(ctx, hipMemcpyAsync(qbase, hin, in.size, hipMemcpyHostToDevice, st));
  }
  ctx->win_ctr_dirty = true;   // cleared again once the results are in
  WinArgs a;
  std::memset(&a, 0, sizeof(a));
  a.kps = (const orbx_keypoint*)(T->dev + T->o_kps); a.desc = T->dev + T->o_desc;
  a.cell_start = (const int32_t*)(T->dev + T->o_cs); a.cell_idx = (const int32_t*)(T->dev + T->o_ci);
  a.minX = T->min_x; a.minY = T->min_y; a.invW = T->inv_w; a.invH = T->inv_h;
  a.qin = (const WinQueryIn*)(qbase + o_q); a.has_aux = qaux ? 1 : 0; a.nq = nq;
  if (kp_skip) { if (direct) { a.skip_map = qbase + o_skip; a.n_skip = nskip; } else a.kp_skip = qbase + o_skip; }
  a.kp_uright = T->has_ur ? (const float*)(T->dev + T->o_ur) : nullptr;
  a.inv_sigma2 = chi2 ? (const float*)(T->dev + T->o_sig) : nullptr;
  uint8_t* const res = direct ? hdev + in.size : dout;
  a.out = (WinQueryOut*)(res + p_q); a.pool = (int2*)(res + p_pool); a.pool_cap = pool_cap; a.total = ctx->d_win_ctr;
  a.out_hdr = (int32_t*)(res + p_hdr); a.compact = compact ? 1 : 0; a.host_out = direct ? 1 : 0; a.self_reset = 1;
  const dim3 gridDim((nq + kWinWPG - 1) / kWinWPG), block(64 * kWinWPG);
  const size_t lds = a.skip_map ? (size_t)nskip : 0;
  const bool timing = ctx->window_timing;   // measurement only (bench.py's matcher_roofline): HIP events around the pass, on the pass's own stream
  if (timing) {
    if (!ctx->ev_w0) { ORBX_HIP(ctx, hipEventCreate(&ctx->ev_w0)); ORBX_HIP(ctx, hipEventCreate(&ctx->ev_w1)); }
    ORBX_HIP(ctx, hipEventRecord(ctx->ev_w0, st));
  }
  if (chi2) hipLaunchKernelGGL((k_window<false, true>), gridDim, block, lds, st, a);
  else if (lists) hipLaunchKernelGGL((k_window<true, false>), gridDim, block, lds, st, a);
  else hipLaunchKernelGGL((k_window<false, false>), gridDim, block, lds, st, a);
  ORBX_HIP(ctx, hipGetLastError());
  if (timing) ORBX_HIP(ctx, hipEventRecord(ctx->ev_w1, st));
  double us_issue;
  if (direct) {
    us_issue = since(tr0);
    const volatile unsigned long long* hdr = (const volatile unsigned long long*)(hout + p_hdr);
    for (unsigned spin = 1;; spin++) {
      if (__atomic_load_n(hdr, __ATOMIC_ACQUIRE) >> 32) break;
      if ((spin & 0x3fff) == 0) {
        const hipError_t qe = hipStreamQuery(st);
        if (qe == hipSuccess) {
          if (__atomic_load_n(hdr, __ATOMIC_ACQUIRE) >> 32) break;
          return set_err(ctx, ORBX_E_DEVICE, std::string(who) + ": window pass finished without publishing its results");
        }
        if (qe != hipErrorNotReady) { ORBX_HIP(ctx, qe); }
      }
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
    }
  } else {
    ORBX_HIP(ctx, hipMemcpyAsync(hout, dout, out.size, hipMemcpyDeviceToHost, st));
    us_issue = since(tr0);
    ORBX_HIP(ctx, hipStreamSynchronize(st));
  }
  ctx->win_ctr_dirty = false;
  const double us_sync = since(tr0);
  if (timing) {
    float ms = 0.f;
    if (hipEventSynchronize(ctx->ev_w1) == hipSuccess && hipEventElapsedTime(&ms, ctx->ev_w0, ctx->ev_w1) == hipSuccess) ctx->last_window_us = 1e3 * ms;
    else (void)hipGetLastError();
  }
  const int total = *(const int32_t*)(hout + p_hdr);
  const WinQueryOut* qo = (const WinQueryOut*)(hout + p_q);
  const WinQueryShort* qs = (const WinQueryShort*)(hout + p_q);
  auto q_start = [&](int q) { return compact ? qs[q].start : qo[q].start; };
  auto q_count = [&](int q) { return compact ? qs[q].count : qo[q].count; };
  if (v_spans) {   // the caller reads the lists where the kernel left them (the call's pinned blob, valid until the context's next call)
    if (total > pool_cap) { if (row_ptr) row_ptr[nq] = total; return set_err(ctx, ORBX_E_CAPACITY, std::string(who) + ": candidate buffer too small"); }
    static_assert(sizeof(orbx_list_span) == sizeof(WinQueryOut) && sizeof(orbx_candidate) == sizeof(int2), "view layout");
    *v_spans = (const orbx_list_span*)(hout + p_q);
    *v_pool = (const orbx_candidate*)(hout + p_pool);
    if (trace)
      std::fprintf(stderr, "[orbx window] %s (resident target, view) n=%d nq=%d total=%d: pack %.1f us, issue %.1f, wait %.1f\n", who, n, nq, total, us_pack,
                   us_issue - us_pack, us_sync - us_issue);
    return total;
  }
  if (row_ptr) {
    int acc = 0;
    for (int q = 0; q < nq; q++) { acc += q_count(q); row_ptr[q + 1] = acc; }
  }
  if (lists) {
    if (total > pool_cap) return set_err(ctx, ORBX_E_CAPACITY, std::string(who) + ": candidate buffer too small");
    const int2* pool = (const int2*)(hout + p_pool);
    for (int q = 0; q < nq; q++) {
      const int2* seg = pool + q_start(q);
      const int o = row_ptr[q];
      const int cnt = q_count(q);
      for (int c = 0; c < cnt; c++) {
        if (cand) cand[o + c] = seg[c].x;
        if (dist) dist[o + c] = seg[c].y;
      }
    }
  }
  if (!compact) for (int q = 0; q < nq; q++) {
    if (best_idx) best_idx[q] = qo[q].best_idx;
    if (best_dist) best_dist[q] = qo[q].best_dist;
    if (second_idx) second_idx[q] = qo[q].second_idx;
    if (second_dist) second_dist[q] = qo[q].second_dist;
  }
  if (trace)
    std::fprintf(stderr, "[orbx window] %s (resident target) n=%d nq=%d total=%d in=%zu B: pack %.1f us, issue %.1f, wait %.1f, scatter %.1f\n", who, n, nq,
                 total, in.size, us_pack, us_issue - us_pack, us_sync - us_issue, since(tr0) - us_sync);
  return lists ? total : (row_ptr ? row_ptr[nq] : 0);
}

}  // namespace orbx

using namespace orbx;

extern "C" {

int orbx_window_search_grid(orbx_ctx* ctx, const orbx_keypoint* kps, const uint8_t* desc, int n, const orbx_grid* grid,
                            const uint8_t* kp_skip, const float* kp_uright, const float* qx, const float* qy, const float* qr,
                            const int32_t* qmin_level, const int32_t* qmax_level, const uint8_t* q_desc, const float* q_xr, int nq,
                            int32_t* row_ptr, int32_t* cand, int32_t* dist, int cand_cap, int32_t* best_idx, int32_t* best_dist,
                            int32_t* second_idx, int32_t* second_dist) {
  if (!ctx || !grid || n < 0 || nq < 0 || !row_ptr || (n > 0 && (!kps || !desc)) ||
      (nq > 0 && (!qx || !qy || !qr || !qmin_level || !qmax_level || !q_desc)) || cand_cap < 0 || ((kp_uright != nullptr) != (q_xr != nullptr)) ||
      !(grid->inv_w > 0.f) || !(grid->inv_h > 0.f))
    return ctx ? set_err(ctx, ORBX_E_INVALID, "orbx_window_search_grid: bad arguments") : ORBX_E_INVALID;
  return window_call(ctx, "orbx_window_search_grid", kps, desc, n, grid, kp_skip, kp_uright, nullptr, 0, qx, qy, qr, qmin_level, qmax_level, q_xr,
                     q_desc, nq, cand || dist, row_ptr, cand, dist, cand_cap, best_idx, best_dist, second_idx, second_dist);
}

int orbx_undistort_keypoints_device(orbx_ctx* ctx, const orbx_keypoint* d_kps, const int32_t* d_counts, int nframes, int capacity, float fx,
                                    float fy, float cx, float cy, const float* dist_coef, int n_coef, orbx_keypoint* d_kps_un, void* stream) {
  if (!ctx || !d_kps || !d_counts || !d_kps_un || nframes <= 0 || capacity <= 0 || !dist_coef || n_coef < 4 || !(fx != 0.f) || !(fy != 0.f))
    return ctx ? set_err(ctx, ORBX_E_INVALID, "orbx_undistort_keypoints_device: bad arguments") : ORBX_E_INVALID;
  ORBX_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  UndistortParams P;
  if (!undistort_params(fx, fy, cx, cy, dist_coef, n_coef, P)) {   // mDistCoef[0] == 0: mvKeysUn = mvKeys (src/Frame.cc:749-753)
    ORBX_HIP(ctx, hipMemcpyAsync(d_kps_un, d_kps, sizeof(orbx_keypoint) * (size_t)nframes * capacity, hipMemcpyDeviceToDevice, st));
    return ORBX_OK;
  }
  hipLaunchKernelGGL(k_undistort, dim3((capacity + 255) / 256, nframes), dim3(256), 0, st, d_kps, d_counts, capacity, 0, P, d_kps_un);
  ORBX_HIP(ctx, hipGetLastError());
  return ORBX_OK;
}

int orbx_undistort_keypoints(orbx_ctx* ctx, const orbx_keypoint* kps, int n, float fx, float fy, float cx, float cy, const float* dist_coef,
                             int n_coef, orbx_keypoint* kps_un) {
  if (!ctx || n < 0 || (n > 0 && (!kps || !kps_un)) || !dist_coef || n_coef < 4 || !(fx != 0.f) || !(fy != 0.f))
    return ctx ? set_err(ctx, ORBX_E_INVALID, "orbx_undistort_keypoints: bad arguments") : ORBX_E_INVALID;
  if (n == 0) return ORBX_OK;
  UndistortParams P;
  if (!undistort_params(fx, fy, cx, cy, dist_coef, n_coef, P)) { std::memcpy(kps_un, kps, sizeof(orbx_keypoint) * (size_t)n); return ORBX_OK; }
  ORBX_HIP(ctx, hipSetDevice(ctx->device));
  const size_t bytes = sizeof(orbx_keypoint) * (size_t)n;
  uint8_t* h = nullptr;
  ORBX_HIP(ctx, host_stage(ctx, 2 * bytes + 512, &h));
  std::memcpy(h, kps, bytes);
  ctx->arena.rewind();
  hipError_t aerr = hipSuccess;
  uint8_t* din = (uint8_t*)ctx->arena.alloc(bytes, &aerr);
  ORBX_HIP(ctx, aerr);
  uint8_t* dout = (uint8_t*)ctx->arena.alloc(bytes, &aerr);
  ORBX_HIP(ctx, aerr);
  hipStream_t st = ctx->stream;
  ORBX_HIP(ctx, hipMemcpyAsync(din, h, bytes, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_undistort, dim3((n + 255) / 256, 1), dim3(256), 0, st, (const orbx_keypoint*)din, (const int32_t*)nullptr, n, n, P,
                     (orbx_keypoint*)dout);
  ORBX_HIP(ctx, hipGetLastError());
  uint8_t* hout = h + ((bytes + 255) & ~(size_t)255);
  ORBX_HIP(ctx, hipMemcpyAsync(hout, dout, bytes, hipMemcpyDeviceToHost, st));
  ORBX_HIP(ctx, hipStreamSynchronize(st));
  std::memcpy(kps_un, hout, bytes);
  return ORBX_OK;
}

int orbx_window_nearest(orbx_ctx* ctx, const orbx_keypoint* kps, const uint8_t* desc, int n, const orbx_grid* grid,
                        const float* kp_uright, const float* inv_level_sigma2, int nlevels, const float* qx, const float* qy,
                        const float* qr, const int32_t* qmin_level, const int32_t* qmax_level, const float* q_ur,
                        const uint8_t* q_desc, int nq, int32_t* best_idx, int32_t* best_dist) {
  if (!ctx || !grid || n < 0 || nq < 0 || (n > 0 && (!kps || !desc)) ||
      (nq > 0 && (!qx || !qy || !qr || !qmin_level || !qmax_level || !q_desc || !best_idx || !best_dist)) ||
      (inv_level_sigma2 && (!kp_uright || !q_ur || nlevels <= 0)) || !(grid->inv_w > 0.f) || !(grid->inv_h > 0.f))
    return ctx ? set_err(ctx, ORBX_E_INVALID, "orbx_window_nearest: bad arguments") : ORBX_E_INVALID;
  const int rc = window_call(ctx, "orbx_window_nearest", kps, desc, n, grid, nullptr, inv_level_sigma2 ? kp_uright : nullptr, inv_level_sigma2,
                             nlevels, qx, qy, qr, qmin_level, qmax_level, inv_level_sigma2 ? q_ur : nullptr, q_desc, nq, false, nullptr, nullptr,
                             nullptr, 0, best_idx, best_dist, nullptr, nullptr);
  return rc < 0 ? rc : ORBX_OK;
}

int orbx_target_create(orbx_ctx* ctx, const orbx_keypoint* kps, const uint8_t* desc, int n, const orbx_grid* grid, const float* kp_uright,
                       const float* inv_level_sigma2, int nlevels, orbx_target** target) {
  if (!ctx || !target || !grid || n < 0 || (n > 0 && (!kps || !desc)) || (inv_level_sigma2 && (!kp_uright || nlevels <= 0)) ||
      !(grid->inv_w > 0.f) || !(grid->inv_h > 0.f))
    return ctx ? set_err(ctx, ORBX_E_INVALID, "orbx_target_create: bad arguments") : ORBX_E_INVALID;
  return target_assign(ctx, nullptr, kps, desc, n, grid, kp_uright, inv_level_sigma2, nlevels, target);
}

int orbx_target_assign(orbx_ctx* ctx, orbx_target* target, const orbx_keypoint* kps, const uint8_t* desc, int n, const orbx_grid* grid,
                       const float* kp_uright, const float* inv_level_sigma2, int nlevels) {
  if (!ctx || !target || target->ctx != ctx || !grid || n < 0 || (n > 0 && (!kps || !desc)) || (inv_level_sigma2 && (!kp_uright || nlevels <= 0)) ||
      !(grid->inv_w > 0.f) || !(grid->inv_h > 0.f))
    return ctx ? set_err(ctx, ORBX_E_INVALID, "orbx_target_assign: bad arguments") : ORBX_E_INVALID;
  return target_assign(ctx, target, kps, desc, n, grid, kp_uright, inv_level_sigma2, nlevels, nullptr);
}

void orbx_target_destroy(orbx_target* target) { target_destroy(target); }

int orbx_target_size(const orbx_target* target) { return target && target->valid ? target->n : ORBX_E_INVALID; }

int orbx_target_search(orbx_ctx* ctx, const orbx_target* target, const uint8_t* kp_skip, const float* qx, const float* qy, const float* qr,
                       const int32_t* qmin_level, const int32_t* qmax_level, const uint8_t* q_desc, const float* q_xr, int nq, int32_t* row_ptr,
                       int32_t* cand, int32_t* dist, int cand_cap, int32_t* best_idx, int32_t* best_dist, int32_t* second_idx,
                       int32_t* second_dist) {
  if (!ctx || !target || target->ctx != ctx || nq < 0 || !row_ptr ||
      (nq > 0 && (!qx || !qy || !qr || !qmin_level || !qmax_level || !q_desc)) || (q_xr && !target->has_ur) || cand_cap < 0)
    return ctx ? set_err(ctx, ORBX_E_INVALID, "orbx_target_search: bad arguments") : ORBX_E_INVALID;
  const bool lists = cand != nullptr || dist != nullptr;
  return window_call_target(ctx, "orbx_target_search", target, kp_skip, false, qx, qy, qr, qmin_level, qmax_level, q_xr, q_desc, nq, lists, row_ptr,
                            cand, dist, cand_cap, best_idx, best_dist, second_idx, second_dist);
}

int orbx_target_search_view(orbx_ctx* ctx, const orbx_target* target, const uint8_t* kp_skip, const float* qx, const float* qy, const float* qr,
                            const int32_t* qmin_level, const int32_t* qmax_level, const uint8_t* q_desc, const float* q_xr, int nq,
                            const orbx_list_span** spans, const orbx_candidate** pool) {
  if (!ctx || !target || target->ctx != ctx || nq < 0 || !spans || !pool ||
      (nq > 0 && (!qx || !qy || !qr || !qmin_level || !qmax_level || !q_desc)) || (q_xr && !target->has_ur))
    return ctx ? set_err(ctx, ORBX_E_INVALID, "orbx_target_search_view: bad arguments") : ORBX_E_INVALID;
  *spans = nullptr; *pool = nullptr;
  if (nq == 0 || target->n == 0) return target->valid ? 0 : set_err(ctx, ORBX_E_INVALID, "orbx_target_search_view: invalid target");
  // the pool is sized from what earlier calls of this context needed; a call that needs more reports the size and is repeated once
  // this call's blob: normally the other one than the last view's (which stays alive: orbx.h, valid until the second next view call) — but never
  // one that backs a call issued by _begin and not yet collected by _end (its kernel may still be writing it, and _end reads its header)
  int sel = ctx->view_par ^ 1;
  if (ctx->view_call[sel].pending) sel ^= 1;
  if (ctx->view_call[sel].pending)
    return set_err(ctx, ORBX_E_INVALID, "orbx_target_search_view: both view blobs hold a pending call (finish one with _end, or _cancel it, first)");
  ctx->view_par = sel;
  for (int attempt = 0; attempt < 2; attempt++) {
    std::vector<int32_t>& rp = ctx->view_row_ptr;
    rp.assign((size_t)nq + 1, 0);
    const int rc = window_call_target(ctx, "orbx_target_search_view", target, kp_skip, false, qx, qy, qr, qmin_level, qmax_level, q_xr, q_desc, nq, true,
                                      rp.data(), nullptr, nullptr, ctx->view_pool_cap, nullptr, nullptr, nullptr, nullptr, spans, pool);
    if (rc >= 0 || rc != ORBX_E_CAPACITY || attempt) return rc;
    ctx->view_pool_cap = std::max(2 * ctx->view_pool_cap, rp[nq] + rp[nq] / 4 + 64);
  }
  return ORBX_E_CAPACITY;
}

// ---- a view call as issue + wait ------------------------------------------------------------------------------------------------------------
// begin: this call's blob is chosen (the other one keeps the previous view / the other pending call), the queries are packed into it and the
// window kernel is queued; nothing waits.  end: polls the blob's done word and hands out the view — or, when the learnt pool capacity was too
// small (or the mapped-blob path is not available), runs the ordinary synchronous call on the same blob.  The query arrays must stay valid and
// unchanged until _end.  Between a _begin and its _end the context may take other calls (another _begin, searches on other targets): the stream
// orders the kernels, the results of every call live in its own blob.
int orbx_target_search_view_begin(orbx_ctx* ctx, const orbx_target* target, const uint8_t* kp_skip, const float* qx, const float* qy, const float* qr,
                                  const int32_t* qmin_level, const int32_t* qmax_level, const uint8_t* q_desc, const float* q_xr, int nq) {
  if (!ctx || !target || target->ctx != ctx || nq < 0 ||
      (nq > 0 && (!qx || !qy || !qr || !qmin_level || !qmax_level || !q_desc)) || (q_xr && !target->has_ur))
    return ctx ? set_err(ctx, ORBX_E_INVALID, "orbx_target_search_view_begin: bad arguments") : ORBX_E_INVALID;
  if (!target->valid) return set_err(ctx, ORBX_E_INVALID, "orbx_target_search_view_begin: the target's last orbx_target_assign failed; assign it again");
  const int slot = ctx->view_par ^ 1;
  orbx_ctx::ViewCall& vc = ctx->view_call[slot];
  if (vc.pending) return set_err(ctx, ORBX_E_INVALID, "orbx_target_search_view_begin: both view blobs hold a pending call (finish one with _end first)");
  ctx->view_par = slot;
  vc = orbx_ctx::ViewCall();
  vc.pending = true; vc.target = target; vc.kp_skip = kp_skip; vc.qx = qx; vc.qy = qy; vc.qr = qr; vc.q_xr = q_xr; vc.qlo = qmin_level; vc.qhi = qmax_level;
  vc.q_desc = q_desc; vc.nq = nq;
  const int n = target->n;
  if (nq == 0 || n == 0) { vc.sync_fallback = true; return slot; }   // nothing to launch: _end answers from the ordinary path
  ORBX_HIP(ctx, hipSetDevice(ctx->device));
  const int pool_cap = std::max(ctx->view_pool_cap, 0);
  const int nskip = kp_skip ? (n + 3) & ~3 : 0;
  Layout in;
  const size_t o_q = in.add(sizeof(WinQueryIn) * (size_t)nq), o_skip = in.add((size_t)nskip);
  Layout out;
  const size_t p_hdr = out.add(16), p_q = out.add(sizeof(WinQueryOut) * (size_t)nq), p_pool = out.add(8 * (size_t)pool_cap);
  uint8_t* h = nullptr;
  ORBX_HIP(ctx, host_stage_view(ctx, in.size + out.size, &h));
  uint8_t* hdev = nullptr;
  const bool direct = ctx->window_direct && nskip <= 48 * 1024 && hipHostGetDevicePointer((void**)&hdev, h, 0) == hipSuccess && hdev != nullptr;
  if (!direct) { (void)hipGetLastError(); vc.sync_fallback = true; return slot; }   // the copy + synchronise path has nothing to overlap
  pack_queries(h + o_q, qx, qy, qr, q_xr, qmin_level, qmax_level, q_desc, nq);
  if (kp_skip) { std::memcpy(h + o_skip, kp_skip, (size_t)n); std::memset(h + o_skip + n, 0, (size_t)(nskip - n)); }
  std::memset(h + in.size + p_hdr, 0, 16);
  hipStream_t st = ctx->stream;
  if (!ctx->d_win_ctr) { ORBX_HIP(ctx, hipMalloc((void**)&ctx->d_win_ctr, 64)); ctx->win_ctr_dirty = true; }
  const bool other_pending = ctx->view_call[slot ^ 1].pending && !ctx->view_call[slot ^ 1].sync_fallback;
  if (ctx->win_ctr_dirty && !other_pending) ORBX_HIP(ctx, hipMemsetAsync(ctx->d_win_ctr, 0, 64, st));   // (dirty because of the other pending call: its kernel resets the counters itself)
  ctx->win_ctr_dirty = true;
  WinArgs a;
  std::memset(&a, 0, sizeof(a));
  const orbx_target* T = target;
  a.kps = (const orbx_keypoint*)(T->dev + T->o_kps); a.desc = T->dev + T->o_desc;
  a.cell_start = (const int32_t*)(T->dev + T->o_cs); a.cell_idx = (const int32_t*)(T->dev + T->o_ci);
  a.minX = T->min_x; a.minY = T->min_y; a.invW = T->inv_w; a.invH = T->inv_h;
  a.qin = (const WinQueryIn*)(hdev + o_q); a.has_aux = q_xr ? 1 : 0; a.nq = nq;
  if (kp_skip) { a.skip_map = hdev + o_skip; a.n_skip = nskip; }
  a.kp_uright = T->has_ur ? (const float*)(T->dev + T->o_ur) : nullptr;
  uint8_t* const res = hdev + in.size;
  a.out = (WinQueryOut*)(res + p_q); a.pool = (int2*)(res + p_pool); a.pool_cap = pool_cap; a.total = ctx->d_win_ctr;
  a.out_hdr = (int32_t*)(res + p_hdr); a.compact = 0; a.host_out = 1; a.self_reset = 1;
  hipLaunchKernelGGL((k_window<true, false>), dim3((nq + kWinWPG - 1) / kWinWPG), dim3(64 * kWinWPG), a.skip_map ? (size_t)nskip : 0, st, a);
  ORBX_HIP(ctx, hipGetLastError());
  vc.pool_cap = pool_cap; vc.in_size = in.size; vc.p_hdr = p_hdr; vc.p_q = p_q; vc.p_pool = p_pool;
  return slot;
}

int orbx_target_search_view_end(orbx_ctx* ctx, int slot, const orbx_list_span** spans, const orbx_candidate** pool) {
  if (!ctx || slot < 0 || slot > 1 || !spans || !pool) return ctx ? set_err(ctx, ORBX_E_INVALID, "orbx_target_search_view_end: bad arguments") : ORBX_E_INVALID;
  *spans = nullptr; *pool = nullptr;
  orbx_ctx::ViewCall vc = ctx->view_call[slot];
  if (!vc.pending) return set_err(ctx, ORBX_E_INVALID, "orbx_target_search_view_end: no pending call in this slot");
  ctx->view_call[slot].pending = false;
  auto run_sync = [&]() {   // the ordinary call, on THIS call's blob (the other blob may back a live view or another pending call)
    ctx->view_par = slot ^ 1;   // orbx_target_search_view flips to `slot`
    return orbx_target_search_view(ctx, vc.target, vc.kp_skip, vc.qx, vc.qy, vc.qr, vc.qlo, vc.qhi, vc.q_desc, vc.q_xr, vc.nq, spans, pool);
  };
  if (vc.sync_fallback) return run_sync();
  uint8_t* const hout = ctx->h_view[slot] + vc.in_size;
  const volatile unsigned long long* hdr = (const volatile unsigned long long*)(hout + vc.p_hdr);
  for (unsigned spin = 1;; spin++) {
    if (__atomic_load_n(hdr, __ATOMIC_ACQUIRE) >> 32) break;
    if ((spin & 0x3fff) == 0) {
      const hipError_t qe = hipStreamQuery(ctx->stream);
      if (qe == hipSuccess) {
        if (__atomic_load_n(hdr, __ATOMIC_ACQUIRE) >> 32) break;
        return set_err(ctx, ORBX_E_DEVICE, "orbx_target_search_view_end: window pass finished without publishing its results");
      }
      if (qe != hipErrorNotReady) { ORBX_HIP(ctx, qe); }
    }
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
  }
  if (!ctx->view_call[slot ^ 1].pending) ctx->win_ctr_dirty = false;
  const int total = *(const int32_t*)(hout + vc.p_hdr);
  if (total > vc.pool_cap) {   // the learnt capacity was too small: once more, synchronously, with room
    ctx->view_pool_cap = std::max(2 * ctx->view_pool_cap, total + total / 4 + 64);
    return run_sync();
  }
  *spans = (const orbx_list_span*)(hout + vc.p_q);
  *pool = (const orbx_candidate*)(hout + vc.p_pool);
  return total;
}

// A call issued by _begin whose _end will never come (an error between the two, a dropped ticket): the slot is given back.  The kernel that
// may still be writing the blob is waited for first.
int orbx_target_search_view_cancel(orbx_ctx* ctx, int slot) {
  if (!ctx || slot < 0 || slot > 1) return ctx ? set_err(ctx, ORBX_E_INVALID, "orbx_target_search_view_cancel: bad arguments") : ORBX_E_INVALID;
  orbx_ctx::ViewCall& vc = ctx->view_call[slot];
  if (!vc.pending) return ORBX_OK;
  const bool launched = !vc.sync_fallback;
  vc = orbx_ctx::ViewCall();
  if (launched) {
    ORBX_HIP(ctx, hipSetDevice(ctx->device));
    ORBX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (!ctx->view_call[slot ^ 1].pending) ctx->win_ctr_dirty = true;   // whatever the kernel left in the counters: the next call clears them
  }
  return ORBX_OK;
}

int orbx_target_nearest(orbx_ctx* ctx, const orbx_target* target, int reprojection_gate, const float* qx, const float* qy, const float* qr,
                        const int32_t* qmin_level, const int32_t* qmax_level, const float* q_ur, const uint8_t* q_desc, int nq, int32_t* best_idx,
                        int32_t* best_dist) {
  if (!ctx || !target || target->ctx != ctx || nq < 0 ||
      (nq > 0 && (!qx || !qy || !qr || !qmin_level || !qmax_level || !q_desc || !best_idx || !best_dist)) || (reprojection_gate && !q_ur))
    return ctx ? set_err(ctx, ORBX_E_INVALID, "orbx_target_nearest: bad arguments") : ORBX_E_INVALID;
  const int rc = window_call_target(ctx, "orbx_target_nearest", target, nullptr, reprojection_gate != 0, qx, qy, qr, qmin_level, qmax_level,
                                    reprojection_gate ? q_ur : nullptr, q_desc, nq, false, nullptr, nullptr, nullptr, 0, best_idx, best_dist, nullptr,
                                    nullptr);
  return rc < 0 ? rc : ORBX_OK;
}

}  // extern "C"
