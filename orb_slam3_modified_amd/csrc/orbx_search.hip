// orbx guided search: the frame grid (Frame::AssignFeaturesToGrid / GetFeaturesInArea, src/Frame.cc:385-416,
// :657-735) built and queried on the GPU, and ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:648-763) on
// top of it: candidate generation + all Hamming distances on the device, the order-dependent greedy bookkeeping
// (queries are NOT independent in this routine) replayed on the host in the original query order.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>
#include <vector>

#include "orbx_internal.h"

namespace orbx {

constexpr int kGridCols = 64, kGridRows = 48, kGridCells = kGridCols * kGridRows;  // include/Frame.h:44-45
constexpr int kGridMaxPts = 32768;   // keys of the LDS bitonic sort: 4 B each, 128 KiB of the CU's 160

// One workgroup: cell of every keypoint (Frame::PosInGrid, src/Frame.cc:725-735), then a bitonic sort of
// (cell << 16 | index) in LDS.  Cell id = posX * 48 + posY, so ascending keys list the cells in the order
// GetFeaturesInArea walks them (ix outer, iy inner) and, inside a cell, the keypoints in insertion order.
__global__ __launch_bounds__(1024) void k_frame_grid(const orbx_keypoint* __restrict__ kps, int n, float minX, float minY,
                                                     float invW, float invH, int npad, uint16_t* __restrict__ sorted_idx,
                                                     int32_t* __restrict__ cell_start) {
  extern __shared__ uint32_t keys[];
  const int t = threadIdx.x, T = blockDim.x;
  for (int i = t; i < npad; i += T) {
    uint32_t key = 0xffffffffu;
    if (i < n) {
      const int posX = (int)roundf(__fmul_rn(__fsub_rn(kps[i].x, minX), invW));
      const int posY = (int)roundf(__fmul_rn(__fsub_rn(kps[i].y, minY), invH));
      const bool in = posX >= 0 && posX < kGridCols && posY >= 0 && posY < kGridRows;
      key = ((in ? (uint32_t)(posX * kGridRows + posY) : 0xfffeu) << 16) | (uint32_t)i;
    }
    keys[i] = key;
  }
  __syncthreads();
  for (int k = 2; k <= npad; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = t; i < npad; i += T) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const uint32_t a = keys[i], b = keys[ixj];
          const bool up = (i & k) == 0;
          if ((a > b) == up) { keys[i] = b; keys[ixj] = a; }
        }
      }
      __syncthreads();
    }
  }
  for (int i = t; i < n; i += T) sorted_idx[i] = (uint16_t)(keys[i] & 0xffffu);
  for (int c = t; c <= kGridCells; c += T) {  // first sorted position whose cell is >= c
    const uint32_t want = (uint32_t)c << 16;
    int lo = 0, hi = n;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (keys[mid] < want) lo = mid + 1; else hi = mid;
    }
    cell_start[c] = lo;
  }
}

// Frame::GetFeaturesInArea (src/Frame.cc:657-723), one wave per query.  For a fixed ix the cells iy = min..max are
// consecutive cell ids, i.e. ONE contiguous run of the sorted array, already in the reference's order.
template <bool FILL>
__global__ __launch_bounds__(256) void k_features_in_area(const orbx_keypoint* __restrict__ kps,
                                                          const uint16_t* __restrict__ sorted_idx,
                                                          const int32_t* __restrict__ cell_start, float minX, float minY,
                                                          float invW, float invH, const float* __restrict__ qx,
                                                          const float* __restrict__ qy, const float* __restrict__ qr,
                                                          const int32_t* __restrict__ qmin, const int32_t* __restrict__ qmax,
                                                          int nq, int32_t* __restrict__ counts, const int32_t* __restrict__ row_ptr,
                                                          int32_t* __restrict__ cand, const uint8_t* __restrict__ kp_skip,
                                                          const float* __restrict__ kp_uright, const float* __restrict__ qxr) {
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= nq) return;
  const float x = qx[q], y = qy[q], r = qr[q];
  const int minLevel = qmin[q], maxLevel = qmax[q];
  int total = 0;
  const int nMinCellX = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, minX), r), invW)));
  const int nMaxCellX = min(kGridCols - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, minX), r), invW)));
  const int nMinCellY = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, minY), r), invH)));
  const int nMaxCellY = min(kGridRows - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, minY), r), invH)));
  if (nMinCellX < kGridCols && nMaxCellX >= 0 && nMinCellY < kGridRows && nMaxCellY >= 0) {
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    const int base_out = FILL ? row_ptr[q] : 0;
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++) {
      const int s = cell_start[ix * kGridRows + nMinCellY], e = cell_start[ix * kGridRows + nMaxCellY + 1];
      for (int b0 = s; b0 < e; b0 += 64) {
        const int j = b0 + lane;
        bool ok = false;
        int idx = 0;
        if (j < e) {
          idx = sorted_idx[j];
          const orbx_keypoint kp = kps[idx];
          ok = true;
          if (bCheckLevels) {
            if (kp.octave < minLevel) ok = false;
            if (maxLevel >= 0 && kp.octave > maxLevel) ok = false;
          }
          const float distx = __fsub_rn(kp.x, x), disty = __fsub_rn(kp.y, y);
          ok = ok && fabsf(distx) < r && fabsf(disty) < r;
          // the searches' own candidate gates (src/ORBmatcher.cc:84-93): keypoint already bound to an observed map
          // point; rectified-stereo consistency |projXR - uRight| <= r
          if (kp_skip && kp_skip[idx]) ok = false;
          if (kp_uright && qxr) {
            const float ur = kp_uright[idx];
            if (ur > 0.f && fabsf(__fsub_rn(qxr[q], ur)) > r) ok = false;
          }
        }
        const unsigned long long bal = __ballot(ok);
        if (FILL && ok) cand[base_out + total + __popcll(bal & ((1ull << lane) - 1ull))] = idx;
        total += __popcll(bal);
      }
    }
  }
  if (!FILL && lane == 0) counts[q] = total;
}

// exclusive scan of n int32 (single workgroup), out[n] = total
__global__ __launch_bounds__(1024) void k_scan_excl(const int32_t* __restrict__ in, int n, int32_t* __restrict__ out) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  if (t == 0) carry_s = 0;
  __syncthreads();
  for (int b0 = 0; b0 < n; b0 += 1024) {
    const int i = b0 + t;
    const int v = i < n ? in[i] : 0;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o); if (lane >= o) inc += u; }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int off = carry_s;
    for (int k = 0; k < w; k++) off += wsum[k];
    if (i < n) out[i] = off + inc - v;
    __syncthreads();
    if (t == 1023) carry_s = off + inc;
    __syncthreads();
  }
  if (t == 0) out[n] = carry_s;
}


// ---------------------------------------------------------------------------------------------------------
// Frame::ComputeStereoMatches (src/Frame.cc:811-981) on the device pyramids of the two extractors.
// ---------------------------------------------------------------------------------------------------------
struct StereoGeom {
  const uint8_t* left[kMaxLevels];
  const uint8_t* right[kMaxLevels];
  int w[kMaxLevels], h[kMaxLevels], pitchL[kMaxLevels], pitchR[kMaxLevels];
  float scale[kMaxLevels], inv_scale[kMaxLevels];
  int nlevels;
};

// One wave per left keypoint: (1) best right keypoint among those whose row band contains the left row
// (vRowIndices, :820-840), octave within +-1 and u in [uL - maxD, uL]; first minimum in right-index order wins (:873-889);
// (2) 11 x 11 SAD over 11 horizontal shifts on the pyramid level of the left keypoint (:905-935), 121 (shift, row) items
// spread over the lanes; (3) parabola refinement, disparity / depth (:937-966) by lane 0 with separate IEEE operations.
// sad[iL] = best SAD of an accepted match, -1 otherwise (input of the median filter).
__global__ __launch_bounds__(256) void k_stereo_match(StereoGeom sg, const orbx_keypoint* __restrict__ kpsL,
                                                      const uint8_t* __restrict__ descL, int N,
                                                      const orbx_keypoint* __restrict__ kpsR, const uint8_t* __restrict__ descR,
                                                      int Nr, float mb, float mbf, float* __restrict__ uRight,
                                                      float* __restrict__ depth, int32_t* __restrict__ sad) {
  __shared__ int s_part[4][128];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int iL = blockIdx.x * 4 + wv;
  if (iL >= N) return;
  const orbx_keypoint kpL = kpsL[iL];
  float out_u = -1.0f, out_d = -1.0f;
  int out_sad = -1;
  const int levelL = kpL.octave;
  const float vL = kpL.y, uL = kpL.x;
  const float minD = 0.0f, maxD = __fdiv_rn(mbf, mb);
  const float minU = __fsub_rn(uL, maxD), maxU = __fsub_rn(uL, minD);
  const int row = (int)vL;
  unsigned long long best = ~0ull;
  if (row >= 0 && row < sg.h[0] && !(maxU < 0)) {
    const uint4* qp = (const uint4*)(descL + (size_t)iL * 32);
    const uint4 qa = qp[0], qb = qp[1];
    for (int iR = lane; iR < Nr; iR += 64) {
      const orbx_keypoint kpR = kpsR[iR];
      const float r = __fmul_rn(2.0f, sg.scale[kpR.octave]);
      const int maxr = (int)ceilf(__fadd_rn(kpR.y, r)), minr = (int)floorf(__fsub_rn(kpR.y, r));
      if (row < minr || row > maxr) continue;
      if (kpR.octave < levelL - 1 || kpR.octave > levelL + 1) continue;
      if (!(kpR.x >= minU && kpR.x <= maxU)) continue;
      const uint4* tp = (const uint4*)(descR + (size_t)iR * 32);
      const uint4 ta = tp[0], tb = tp[1];
      const int d = __popc(qa.x ^ ta.x) + __popc(qa.y ^ ta.y) + __popc(qa.z ^ ta.z) + __popc(qa.w ^ ta.w) +
                    __popc(qb.x ^ tb.x) + __popc(qb.y ^ tb.y) + __popc(qb.z ^ tb.z) + __popc(qb.w ^ tb.w);
      const unsigned long long key = ((unsigned long long)d << 32) | (uint32_t)iR;
      best = key < best ? key : best;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const unsigned long long p = __shfl_xor(best, o); best = p < best ? p : best; }
  const int bestDist = best == ~0ull ? 256 : (int)(best >> 32);
  const int TH_HIGH = 100, thOrbDist = 75;
  if (bestDist < TH_HIGH && bestDist < thOrbDist) {  // wave-uniform
    const int bestIdxR = (int)(uint32_t)best;
    const float uR0 = kpsR[bestIdxR].x;
    const float scaleFactor = sg.inv_scale[levelL];
    const float scaleduL = roundf(__fmul_rn(kpL.x, scaleFactor));
    const float scaledvL = roundf(__fmul_rn(kpL.y, scaleFactor));
    const float scaleduR0 = roundf(__fmul_rn(uR0, scaleFactor));
    const int wnd = 5, Ls = 5;
    const float iniu = __fsub_rn(__fadd_rn(scaleduR0, (float)Ls), (float)wnd);
    const float endu = __fadd_rn(__fadd_rn(__fadd_rn(scaleduR0, (float)Ls), (float)wnd), 1.0f);
    if (!(iniu < 0 || endu >= (float)sg.w[levelL])) {
      const int r0 = (int)__fsub_rn(scaledvL, (float)wnd), cL0 = (int)__fsub_rn(scaleduL, (float)wnd);
      const uint8_t* IL = sg.left[levelL];
      const uint8_t* IR = sg.right[levelL];
      const int pL = sg.pitchL[levelL], pR = sg.pitchR[levelL];
      for (int it = lane; it < 121; it += 64) {
        const int si = it / 11, rr = it - si * 11;
        const int cR0 = (int)__fsub_rn(__fadd_rn(scaleduR0, (float)(si - Ls)), (float)wnd);
        const uint8_t* a = IL + (size_t)(r0 + rr) * pL + cL0;
        const uint8_t* b = IR + (size_t)(r0 + rr) * pR + cR0;
        int acc = 0;
#pragma unroll
        for (int cc = 0; cc < 11; cc++) { const int d = (int)a[cc] - (int)b[cc]; acc += d < 0 ? -d : d; }
        s_part[wv][it] = acc;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (lane == 0) {
        float vDists[11];
        int bestS = 0x7fffffff, bestincR = 0;
        for (int si = 0; si < 11; si++) {
          int acc = 0;
          for (int rr = 0; rr < 11; rr++) acc += s_part[wv][si * 11 + rr];
          const float dist = (float)acc;
          if (dist < (float)bestS) { bestS = acc; bestincR = si - Ls; }
          vDists[si] = dist;
        }
        if (!(bestincR == -Ls || bestincR == Ls)) {
          const float dist1 = vDists[Ls + bestincR - 1], dist2 = vDists[Ls + bestincR], dist3 = vDists[Ls + bestincR + 1];
          const float den = __fmul_rn(2.0f, __fsub_rn(__fadd_rn(dist1, dist3), __fmul_rn(2.0f, dist2)));
          const float deltaR = __fdiv_rn(__fsub_rn(dist1, dist3), den);
          if (!(deltaR < -1 || deltaR > 1)) {
            float bestuR = __fmul_rn(sg.scale[levelL], __fadd_rn(__fadd_rn(scaleduR0, (float)bestincR), deltaR));
            float disparity = __fsub_rn(uL, bestuR);
            if (disparity >= minD && disparity < maxD) {
              if (disparity <= 0) { disparity = (float)0.01; bestuR = (float)__dsub_rn((double)uL, 0.01); }
              out_d = __fdiv_rn(mbf, disparity);
              out_u = bestuR;
              out_sad = bestS;
            }
          }
        }
      }
    }
  }
  if (lane == 0) { uRight[iL] = out_u; depth[iL] = out_d; sad[iL] = out_sad; }
}

// The median filter of the accepted matches (src/Frame.cc:969-981): only the value of the (n/2)-th smallest SAD
// matters (all matches with SAD >= 1.5 * 1.4 * median are removed), so no sort is needed — one workgroup counts ranks.
// sad values staged in LDS when they fit (the median is found by rank counting: every thread walks all of them); `h_u / h_d / h_done`
// (optional): the caller-visible mapped copy of the two result arrays and the word the waiting host polls
constexpr int kStereoLdsSad = 12288;
__global__ __launch_bounds__(1024) void k_stereo_filter(int N, float* __restrict__ uRight, float* __restrict__ depth,
                                                        const int32_t* __restrict__ sad, int32_t* __restrict__ kept_out,
                                                        float* __restrict__ h_u, float* __restrict__ h_d,
                                                        unsigned long long* __restrict__ h_done) {
  __shared__ int s_n, s_median, s_kept;
  __shared__ int32_t s_sad[kStereoLdsSad];
  const int t = threadIdx.x;
  if (t == 0) { s_n = 0; s_median = -1; s_kept = 0; }
  const bool lds = N <= kStereoLdsSad;   // block-uniform
  if (lds) for (int i = t; i < N; i += 1024) s_sad[i] = sad[i];
  __syncthreads();
  const int32_t* S = lds ? s_sad : sad;
  int mine = 0;
  for (int i = t; i < N; i += 1024) mine += S[i] >= 0;
  if (mine) atomicAdd(&s_n, mine);
  __syncthreads();
  const int n = s_n;
  if (n > 0) {   // block-uniform
    const int k = n / 2;
    for (int i = t; i < N; i += 1024) {
      const int v = S[i];
      if (v < 0) continue;
      int lo = 0, eq = 0;
      for (int j = 0; j < N; j++) { const int u = S[j]; lo += (u >= 0 && u < v); eq += (u == v); }
      if (lo <= k && k < lo + eq) s_median = v;   // every thread that hits writes the same value
    }
    __syncthreads();
    const float thDist = __fmul_rn(1.5f * 1.4f, (float)s_median);
    int kept = 0;
    for (int i = t; i < N; i += 1024) {
      const int v = S[i];
      if (v < 0) continue;
      if ((float)v < thDist) kept++;
      else { uRight[i] = -1.0f; depth[i] = -1.0f; }
    }
    if (kept) atomicAdd(&s_kept, kept);
  }
  __syncthreads();
  if (t == 0) *kept_out = s_kept;
  if (h_done) {   // block-uniform: every thread mirrors the entries it finalised, then ONE system-scope release and the done word
    for (int i = t; i < N; i += 1024) { h_u[i] = uRight[i]; h_d[i] = depth[i]; }
    __builtin_amdgcn_s_waitcnt(0);   // this wave's stores have left
    __syncthreads();
    if (t == 0) {
      __threadfence_system();
      __hip_atomic_store(h_done, 0x100000000ull | (unsigned long long)(uint32_t)s_kept, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// input blob from mapped pinned memory into HBM (k_stereo_match gathers right keypoints and descriptors at random)
__global__ __launch_bounds__(256) void k_stereo_stage_in(const uint4* __restrict__ src, uint4* __restrict__ dst, int n16) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) dst[i] = src[i];
}

// Scratch buffers of one entry-point call: slices of the context's arena (rewound by ArenaScope at the call's start).
static thread_local DeviceArena* t_arena = nullptr;
struct ArenaScope {
  explicit ArenaScope(orbx_ctx* ctx) { ctx->arena.rewind(); t_arena = &ctx->arena; }
  ~ArenaScope() { t_arena = nullptr; }
};
template <typename T>
struct DBuf {
  T* p = nullptr;
  hipError_t alloc(size_t n) {
    hipError_t e = hipSuccess;
    p = (T*)t_arena->alloc(std::max<size_t>(n, 1) * sizeof(T), &e);
    return e;
  }
};

struct GridOnDevice {
  DBuf<orbx_keypoint> kps;
  DBuf<uint16_t> sorted;
  DBuf<int32_t> cell_start;
  float minX = 0, minY = 0, invW = 0, invH = 0;
};

static int build_grid(orbx_ctx* ctx, const orbx_keypoint* kps, int n, float min_x, float min_y, float max_x, float max_y,
                      GridOnDevice& g) {
  if (n > kGridMaxPts) return set_err(ctx, ORBX_E_CAPACITY, "frame grid: more than 32768 keypoints");
  g.minX = min_x; g.minY = min_y;
  g.invW = (float)kGridCols / (float)(max_x - min_x);   // src/Frame.cc:159-160
  g.invH = (float)kGridRows / (float)(max_y - min_y);
  ORBX_HIP(ctx, g.kps.alloc(n)); ORBX_HIP(ctx, g.sorted.alloc(n)); ORBX_HIP(ctx, g.cell_start.alloc(kGridCells + 1));
  if (n) ORBX_HIP(ctx, hipMemcpyAsync(g.kps.p, kps, sizeof(orbx_keypoint) * n, hipMemcpyHostToDevice, ctx->stream));
  int npad = 2;
  while (npad < n) npad <<= 1;
  if ((size_t)npad * 4 > 64 * 1024) {
    if (ensure_dynamic_lds((const void*)k_frame_grid, npad * 4) != hipSuccess)
      return set_err(ctx, ORBX_E_CAPACITY, "frame grid: keypoints do not fit the LDS sort");
  }
  hipLaunchKernelGGL(k_frame_grid, dim3(1), dim3(1024), (size_t)npad * 4, ctx->stream, g.kps.p, n, g.minX, g.minY, g.invW, g.invH,
                     npad, g.sorted.p, g.cell_start.p);
  ORBX_HIP(ctx, hipGetLastError());
  return ORBX_OK;
}

struct AreaGates {   // optional per-keypoint / per-query gates, host pointers (nullptr = off)
  const uint8_t* kp_skip = nullptr;
  const float* kp_uright = nullptr;
  const float* qxr = nullptr;
  int n = 0;
};

struct AreaQueries {
  DBuf<float> x, y, r;
  DBuf<int32_t> lmin, lmax, counts, row_ptr, cand;
  int nq = 0, nnz = 0;
};

// count -> scan -> fill; leaves row_ptr / cand on the device, nnz on the host
static int run_area(orbx_ctx* ctx, const GridOnDevice& g, const float* qx, const float* qy, const float* qr, const int32_t* qmin,
                    const int32_t* qmax, int nq, AreaQueries& a, const AreaGates& gates = AreaGates()) {
  a.nq = nq;
  DBuf<uint8_t> d_skip; DBuf<float> d_ur, d_qxr;
  if (gates.kp_skip && gates.n) {
    ORBX_HIP(ctx, d_skip.alloc(gates.n));
    ORBX_HIP(ctx, hipMemcpyAsync(d_skip.p, gates.kp_skip, gates.n, hipMemcpyHostToDevice, ctx->stream));
  }
  if (gates.kp_uright && gates.qxr && gates.n && nq) {
    ORBX_HIP(ctx, d_ur.alloc(gates.n)); ORBX_HIP(ctx, d_qxr.alloc(nq));
    ORBX_HIP(ctx, hipMemcpyAsync(d_ur.p, gates.kp_uright, sizeof(float) * gates.n, hipMemcpyHostToDevice, ctx->stream));
    ORBX_HIP(ctx, hipMemcpyAsync(d_qxr.p, gates.qxr, sizeof(float) * nq, hipMemcpyHostToDevice, ctx->stream));
  }
  ORBX_HIP(ctx, a.x.alloc(nq)); ORBX_HIP(ctx, a.y.alloc(nq)); ORBX_HIP(ctx, a.r.alloc(nq));
  ORBX_HIP(ctx, a.lmin.alloc(nq)); ORBX_HIP(ctx, a.lmax.alloc(nq)); ORBX_HIP(ctx, a.counts.alloc(nq)); ORBX_HIP(ctx, a.row_ptr.alloc(nq + 1));
  if (nq == 0) { ORBX_HIP(ctx, hipMemsetAsync(a.row_ptr.p, 0, sizeof(int32_t), ctx->stream)); a.nnz = 0; return ORBX_OK; }
  const size_t fb = sizeof(float) * nq, ib = sizeof(int32_t) * nq;
  ORBX_HIP(ctx, hipMemcpyAsync(a.x.p, qx, fb, hipMemcpyHostToDevice, ctx->stream));
  ORBX_HIP(ctx, hipMemcpyAsync(a.y.p, qy, fb, hipMemcpyHostToDevice, ctx->stream));
  ORBX_HIP(ctx, hipMemcpyAsync(a.r.p, qr, fb, hipMemcpyHostToDevice, ctx->stream));
  ORBX_HIP(ctx, hipMemcpyAsync(a.lmin.p, qmin, ib, hipMemcpyHostToDevice, ctx->stream));
  ORBX_HIP(ctx, hipMemcpyAsync(a.lmax.p, qmax, ib, hipMemcpyHostToDevice, ctx->stream));
  const dim3 grid((nq + 3) / 4), block(256);
  hipLaunchKernelGGL(k_features_in_area<false>, grid, block, 0, ctx->stream, g.kps.p, g.sorted.p, g.cell_start.p, g.minX, g.minY, g.invW,
                     g.invH, a.x.p, a.y.p, a.r.p, a.lmin.p, a.lmax.p, nq, a.counts.p, (const int32_t*)nullptr, (int32_t*)nullptr,
                     (const uint8_t*)d_skip.p, (const float*)d_ur.p, (const float*)d_qxr.p);
  hipLaunchKernelGGL(k_scan_excl, dim3(1), dim3(1024), 0, ctx->stream, a.counts.p, nq, a.row_ptr.p);
  ORBX_HIP(ctx, hipGetLastError());
  int32_t nnz = 0;
  ORBX_HIP(ctx, hipMemcpyAsync(&nnz, a.row_ptr.p + nq, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  ORBX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  a.nnz = nnz;
  ORBX_HIP(ctx, a.cand.alloc(nnz));
  if (nnz) {
    hipLaunchKernelGGL(k_features_in_area<true>, grid, block, 0, ctx->stream, g.kps.p, g.sorted.p, g.cell_start.p, g.minX, g.minY, g.invW,
                       g.invH, a.x.p, a.y.p, a.r.p, a.lmin.p, a.lmax.p, nq, (int32_t*)nullptr, a.row_ptr.p, a.cand.p,
                       (const uint8_t*)d_skip.p, (const float*)d_ur.p, (const float*)d_qxr.p);
    ORBX_HIP(ctx, hipGetLastError());
  }
  return ORBX_OK;
}

// Window query + every candidate's Hamming distance + per-query best / second, in one pass over the device: the shared
// core of the guided searches.  Host vectors out.
struct WindowResult {
  std::vector<int32_t> row_ptr, cand, dist, best_idx, best_dist, second_idx, second_dist;
  int nnz = 0;
};

static int window_search_core(orbx_ctx* ctx, const orbx_keypoint* kps, const uint8_t* desc, int n, float min_x, float min_y, float max_x,
                              float max_y, const AreaGates& gates, const float* qx, const float* qy, const float* qr, const int32_t* qmin,
                              const int32_t* qmax, const uint8_t* qdesc, int nq, bool want_best, WindowResult& out) {
  out.row_ptr.assign(nq + 1, 0);
  out.nnz = 0;
  if (want_best) { out.best_idx.assign(nq, -1); out.best_dist.assign(nq, 256); out.second_idx.assign(nq, -1); out.second_dist.assign(nq, 256); }
  if (nq == 0 || n == 0) return ORBX_OK;
  // the grid of Frame::AssignFeaturesToGrid is assigned on the device (src/Frame.cc:159-160 for the cell sizes)
  orbx_grid g;
  g.min_x = min_x; g.min_y = min_y;
  g.inv_w = (float)kGridCols / (float)(max_x - min_x);
  g.inv_h = (float)kGridRows / (float)(max_y - min_y);
  g.cell_start = nullptr; g.cell_idx = nullptr;
  const bool stereo_gate = gates.kp_uright && gates.qxr;
  size_t cap = std::max<size_t>(out.cand.size(), 16384);
  for (int attempt = 0; attempt < 2; attempt++) {
    out.cand.resize(cap); out.dist.resize(cap);
    const int rc = window_call(ctx, "window search", kps, desc, n, &g, gates.kp_skip, stereo_gate ? gates.kp_uright : nullptr, nullptr, 0, qx, qy,
                               qr, qmin, qmax, stereo_gate ? gates.qxr : nullptr, qdesc, nq, true, out.row_ptr.data(), out.cand.data(),
                               out.dist.data(), (int)cap, want_best ? out.best_idx.data() : nullptr, want_best ? out.best_dist.data() : nullptr,
                               want_best ? out.second_idx.data() : nullptr, want_best ? out.second_dist.data() : nullptr);
    if (rc >= 0) { out.nnz = rc; return ORBX_OK; }
    if (rc != ORBX_E_CAPACITY || attempt) return rc;
    cap = (size_t)out.row_ptr[nq] + 64;   // complete on ORBX_E_CAPACITY
  }
  return ORBX_OK;
}

}  // namespace orbx

using namespace orbx;

extern "C" {

int orbx_features_in_area(orbx_ctx* ctx, const orbx_keypoint* kps, int n, float min_x, float min_y, float max_x, float max_y,
                          const float* qx, const float* qy, const float* qr, const int32_t* qmin_level, const int32_t* qmax_level,
                          int nq, int32_t* row_ptr, int32_t* cand, int cand_cap) {
  if (!ctx || n < 0 || nq < 0 || !row_ptr || (n > 0 && !kps) || (nq > 0 && (!qx || !qy || !qr || !qmin_level || !qmax_level)) ||
      !(max_x > min_x) || !(max_y > min_y))
    return ctx ? set_err(ctx, ORBX_E_INVALID, "orbx_features_in_area: bad arguments") : ORBX_E_INVALID;
  ORBX_HIP(ctx, hipSetDevice(ctx->device));
  ArenaScope scope(ctx);
  GridOnDevice g;
  int rc = build_grid(ctx, kps, n, min_x, min_y, max_x, max_y, g);
  if (rc != ORBX_OK) return rc;
  AreaQueries a;
  rc = run_area(ctx, g, qx, qy, qr, qmin_level, qmax_level, nq, a);
  if (rc != ORBX_OK) return rc;
  if (a.nnz > cand_cap) return set_err(ctx, ORBX_E_CAPACITY, "orbx_features_in_area: candidate buffer too small");
  ORBX_HIP(ctx, hipMemcpyAsync(row_ptr, a.row_ptr.p, sizeof(int32_t) * (nq + 1), hipMemcpyDeviceToHost, ctx->stream));
  if (a.nnz) ORBX_HIP(ctx, hipMemcpyAsync(cand, a.cand.p, sizeof(int32_t) * a.nnz, hipMemcpyDeviceToHost, ctx->stream));
  ORBX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return a.nnz;
}

int orbx_search_for_initialization(orbx_ctx* ctx, const orbx_keypoint* kps1, const uint8_t* desc1, int n1, const orbx_keypoint* kps2,
                                   const uint8_t* desc2, int n2, float min_x, float min_y, float max_x, float max_y, float* prev_xy,
                                   int window_size, float nn_ratio, int check_orientation, int32_t* matches12, int* nmatches_out) {
  if (!ctx || n1 < 0 || n2 < 0 || !nmatches_out || (n1 > 0 && (!kps1 || !desc1 || !prev_xy || !matches12)) ||
      (n2 > 0 && (!kps2 || !desc2)) || !(max_x > min_x) || !(max_y > min_y))
    return ctx ? set_err(ctx, ORBX_E_INVALID, "orbx_search_for_initialization: bad arguments") : ORBX_E_INVALID;
  *nmatches_out = 0;
  for (int i = 0; i < n1; i++) matches12[i] = -1;
  if (n1 == 0 || n2 == 0) return ORBX_OK;
  ORBX_HIP(ctx, hipSetDevice(ctx->device));
  // queries: the level-0 keypoints of F1 (src/ORBmatcher.cc:664-666), window around vbPrevMatched, level1..level1
  std::vector<int> qi;
  std::vector<float> qx, qy, qr;
  std::vector<int32_t> ql;
  std::vector<uint8_t> qd;
  for (int i1 = 0; i1 < n1; i1++) {
    if (kps1[i1].octave > 0) continue;
    qi.push_back(i1);
    qx.push_back(prev_xy[2 * i1]); qy.push_back(prev_xy[2 * i1 + 1]); qr.push_back((float)window_size);
    ql.push_back(kps1[i1].octave);
    qd.insert(qd.end(), desc1 + (size_t)i1 * 32, desc1 + (size_t)i1 * 32 + 32);
  }
  const int nq = (int)qi.size();
  if (nq == 0) return ORBX_OK;
  ArenaScope scope(ctx);
  GridOnDevice g;
  int rc = build_grid(ctx, kps2, n2, min_x, min_y, max_x, max_y, g);
  if (rc != ORBX_OK) return rc;
  AreaQueries a;
  rc = run_area(ctx, g, qx.data(), qy.data(), qr.data(), ql.data(), ql.data(), nq, a);
  if (rc != ORBX_OK) return rc;
  std::vector<int32_t> row_ptr(nq + 1), cand(std::max(a.nnz, 1)), dist(std::max(a.nnz, 1));
  ORBX_HIP(ctx, hipMemcpyAsync(row_ptr.data(), a.row_ptr.p, sizeof(int32_t) * (nq + 1), hipMemcpyDeviceToHost, ctx->stream));
  if (a.nnz) {
    // every candidate distance on the device (ORBmatcher::DescriptorDistance, src/ORBmatcher.cc:685)
    DBuf<uint8_t> dq, dt;
    DBuf<int32_t> ddist;
    ORBX_HIP(ctx, dq.alloc((size_t)nq * 32)); ORBX_HIP(ctx, dt.alloc((size_t)n2 * 32)); ORBX_HIP(ctx, ddist.alloc(a.nnz));
    ORBX_HIP(ctx, hipMemcpyAsync(dq.p, qd.data(), (size_t)nq * 32, hipMemcpyHostToDevice, ctx->stream));
    ORBX_HIP(ctx, hipMemcpyAsync(dt.p, desc2, (size_t)n2 * 32, hipMemcpyHostToDevice, ctx->stream));
    rc = orbx_nn_csr_device(ctx, dq.p, nq, dt.p, n2, a.row_ptr.p, a.cand.p, 0, nullptr, nullptr, nullptr, nullptr, ddist.p, ctx->stream);
    if (rc != ORBX_OK) return rc;
    ORBX_HIP(ctx, hipMemcpyAsync(cand.data(), a.cand.p, sizeof(int32_t) * a.nnz, hipMemcpyDeviceToHost, ctx->stream));
    ORBX_HIP(ctx, hipMemcpyAsync(dist.data(), ddist.p, sizeof(int32_t) * a.nnz, hipMemcpyDeviceToHost, ctx->stream));
    ORBX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  } else {
    ORBX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  // ---- host replay of the greedy bookkeeping in the original query order (src/ORBmatcher.cc:661-763)
  const int TH_LOW = 50, HISTO_LENGTH = 30;
  int nmatches = 0;
  std::vector<int> rotHist[30];
  const float factor = 1.0f / HISTO_LENGTH;
  std::vector<int> vMatchedDistance(n2, INT_MAX), vnMatches21(n2, -1);
  for (int q = 0; q < nq; q++) {
    const int i1 = qi[q];
    if (row_ptr[q] == row_ptr[q + 1]) continue;
    int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
    for (int c = row_ptr[q]; c < row_ptr[q + 1]; c++) {
      const int i2 = cand[c], d = dist[c];
      if (vMatchedDistance[i2] <= d) continue;
      if (d < bestDist) { bestDist2 = bestDist; bestDist = d; bestIdx2 = i2; }
      else if (d < bestDist2) { bestDist2 = d; }
    }
    if (bestDist <= TH_LOW) {
      if (bestDist < (float)bestDist2 * nn_ratio) {
        if (vnMatches21[bestIdx2] >= 0) { matches12[vnMatches21[bestIdx2]] = -1; nmatches--; }
        matches12[i1] = bestIdx2;
        vnMatches21[bestIdx2] = i1;
        vMatchedDistance[bestIdx2] = bestDist;
        nmatches++;
        if (check_orientation) {
          float rot = kps1[i1].angle - kps2[bestIdx2].angle;
          if (rot < 0.0) rot += 360.0f;
          int bin = (int)std::round(rot * factor);
          if (bin == HISTO_LENGTH) bin = 0;
          if (bin >= 0 && bin < HISTO_LENGTH) rotHist[bin].push_back(i1);
        }
      }
    }
  }
  if (check_orientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < HISTO_LENGTH; i++) {  // ComputeThreeMaxima, src/ORBmatcher.cc:2012-2053
      const int s = (int)rotHist[i].size();
      if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
      else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
      else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (int idx1 : rotHist[i])
        if (matches12[idx1] >= 0) { matches12[idx1] = -1; nmatches--; }
    }
  }
  for (int i1 = 0; i1 < n1; i1++)
    if (matches12[i1] >= 0) { prev_xy[2 * i1] = kps2[matches12[i1]].x; prev_xy[2 * i1 + 1] = kps2[matches12[i1]].y; }
  *nmatches_out = nmatches;
  return ORBX_OK;
}

int orbx_window_search(orbx_ctx* ctx, const orbx_keypoint* kps, const uint8_t* desc, int n, float min_x, float min_y, float max_x,
                       float max_y, const uint8_t* kp_skip, const float* kp_uright, const float* qx, const float* qy, const float* qr,
                       const int32_t* qmin_level, const int32_t* qmax_level, const uint8_t* q_desc, const float* q_xr, int nq,
                       int32_t* row_ptr, int32_t* cand, int32_t* dist, int cand_cap, int32_t* best_idx, int32_t* best_dist,
                       int32_t* second_idx, int32_t* second_dist) {
  if (!ctx || n < 0 || nq < 0 || !row_ptr || (n > 0 && (!kps || !desc)) ||
      (nq > 0 && (!qx || !qy || !qr || !qmin_level || !qmax_level || !q_desc)) || !(max_x > min_x) || !(max_y > min_y) || cand_cap < 0)
    return ctx ? set_err(ctx, ORBX_E_INVALID, "orbx_window_search: bad arguments") : ORBX_E_INVALID;
  orbx_grid g;
  g.min_x = min_x; g.min_y = min_y;
  g.inv_w = (float)kGridCols / (float)(max_x - min_x);   // src/Frame.cc:159-160
  g.inv_h = (float)kGridRows / (float)(max_y - min_y);
  g.cell_start = nullptr; g.cell_idx = nullptr;
  const bool stereo_gate = kp_uright && q_xr;
  return window_call(ctx, "orbx_window_search", kps, desc, n, &g, kp_skip, stereo_gate ? kp_uright : nullptr, nullptr, 0, qx, qy, qr, qmin_level,
                     qmax_level, stereo_gate ? q_xr : nullptr, q_desc, nq, cand || dist, row_ptr, cand, dist, cand_cap, best_idx, best_dist,
                     second_idx, second_dist);
}

int orbx_search_by_projection(orbx_ctx* ctx, const orbx_keypoint* kps_un, const uint8_t* desc, const float* u_right, int32_t* kp_obs, int n,
                              float min_x, float min_y, float max_x, float max_y, const float* scale_factors, int nlevels,
                              const uint8_t* mp_in_view, const float* mp_proj_x, const float* mp_proj_y, const float* mp_proj_xr,
                              const float* mp_view_cos, const int32_t* mp_level, const uint8_t* mp_desc, const int32_t* mp_obs, int nmp,
                              float th, float nn_ratio, int32_t* kp_match, int* nmatches_out) {
  if (!ctx || n < 0 || nmp < 0 || nlevels <= 0 || !scale_factors || !nmatches_out || (n > 0 && (!kps_un || !desc || !kp_obs || !kp_match)) ||
      (nmp > 0 && (!mp_in_view || !mp_proj_x || !mp_proj_y || !mp_view_cos || !mp_level || !mp_desc || !mp_obs)) || !(max_x > min_x) ||
      !(max_y > min_y) || (u_right && nmp > 0 && !mp_proj_xr))
    return ctx ? set_err(ctx, ORBX_E_INVALID, "orbx_search_by_projection: bad arguments") : ORBX_E_INVALID;
  *nmatches_out = 0;
  for (int i = 0; i < n; i++) kp_match[i] = -1;
  if (n == 0 || nmp == 0) return ORBX_OK;
  ORBX_HIP(ctx, hipSetDevice(ctx->device));
  // one window query per map point in view (src/ORBmatcher.cc:49-74)
  const bool bFactor = th != 1.0;
  std::vector<int> qi;
  std::vector<float> qx, qy, qr, qxr;
  std::vector<int32_t> qlo, qhi;
  std::vector<uint8_t> qd;
  for (int i = 0; i < nmp; i++) {
    if (!mp_in_view[i]) continue;
    const int lvl = mp_level[i];
    if (lvl < 0 || lvl >= nlevels) return set_err(ctx, ORBX_E_INVALID, "orbx_search_by_projection: predicted level out of range");
    float r = mp_view_cos[i] > 0.998 ? 2.5f : 4.0f;   // RadiusByViewingCos, :214-220
    if (bFactor) r *= th;
    qi.push_back(i);
    qx.push_back(mp_proj_x[i]); qy.push_back(mp_proj_y[i]); qr.push_back(r * scale_factors[lvl]);
    qxr.push_back(mp_proj_xr ? mp_proj_xr[i] : 0.f);
    qlo.push_back(lvl - 1); qhi.push_back(lvl);
    qd.insert(qd.end(), mp_desc + (size_t)i * 32, mp_desc + (size_t)i * 32 + 32);
  }
  const int nq = (int)qi.size();
  if (nq == 0) return ORBX_OK;
  std::vector<uint8_t> skip(n);
  for (int i = 0; i < n; i++) skip[i] = kp_obs[i] > 0;   // bound to an observed map point: never a candidate (:81-83)
  ArenaScope scope(ctx);
  AreaGates gates;
  gates.kp_skip = skip.data(); gates.kp_uright = u_right; gates.qxr = u_right ? qxr.data() : nullptr; gates.n = n;
  WindowResult w;
  int rc = window_search_core(ctx, kps_un, desc, n, min_x, min_y, max_x, max_y, gates, qx.data(), qy.data(), qr.data(), qlo.data(),
                              qhi.data(), qd.data(), nq, false, w);
  if (rc != ORBX_OK) return rc;
  // host replay of the order-dependent part (:76-140): a keypoint bound to an observed map point by an EARLIER query of
  // this call is no candidate for the later ones
  const int TH_HIGH = 100;
  int nmatches = 0;
  for (int q = 0; q < nq; q++) {
    const int iMP = qi[q];
    bool any = false;   // vIndices.empty() is decided before the occupancy / stereo gates; those only `continue`
    int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
    for (int c = w.row_ptr[q]; c < w.row_ptr[q + 1]; c++) {
      const int idx = w.cand[c], d = w.dist[c];
      any = true;
      if (kp_obs[idx] > 0) continue;
      if (d < bestDist) { bestDist2 = bestDist; bestDist = d; bestLevel2 = bestLevel; bestLevel = kps_un[idx].octave; bestIdx = idx; }
      else if (d < bestDist2) { bestLevel2 = kps_un[idx].octave; bestDist2 = d; }
    }
    (void)any;
    if (bestDist <= TH_HIGH) {
      if (bestLevel == bestLevel2 && (float)bestDist > nn_ratio * (float)bestDist2) continue;
      if (bestLevel != bestLevel2 || (float)bestDist <= nn_ratio * (float)bestDist2) {
        kp_match[bestIdx] = iMP;
        kp_obs[bestIdx] = mp_obs[iMP];
        nmatches++;
      }
    }
  }
  *nmatches_out = nmatches;
  return ORBX_OK;
}

int orbx_search_by_projection_last(orbx_ctx* ctx, const orbx_keypoint* kps_un, const uint8_t* desc, const float* u_right, int32_t* kp_obs,
                                   int n, float min_x, float min_y, float max_x, float max_y, const float* scale_factors, int nlevels,
                                   float mbf, const uint8_t* lp_valid, const float* lp_u, const float* lp_v, const float* lp_invz,
                                   const int32_t* lp_octave, const float* lp_angle, const uint8_t* lp_desc, const int32_t* lp_obs, int nlast,
                                   float th, int direction, int check_orientation, int32_t* kp_match, int* nmatches_out) {
  if (!ctx || n < 0 || nlast < 0 || nlevels <= 0 || !scale_factors || !nmatches_out || direction < 0 || direction > 2 ||
      (n > 0 && (!kps_un || !desc || !kp_obs || !kp_match)) ||
      (nlast > 0 && (!lp_valid || !lp_u || !lp_v || !lp_octave || !lp_angle || !lp_desc || !lp_obs)) || (u_right && nlast > 0 && !lp_invz) ||
      !(max_x > min_x) || !(max_y > min_y))
    return ctx ? set_err(ctx, ORBX_E_INVALID, "orbx_search_by_projection_last: bad arguments") : ORBX_E_INVALID;
  *nmatches_out = 0;
  for (int i = 0; i < n; i++) kp_match[i] = -1;
  if (n == 0 || nlast == 0) return ORBX_OK;
  ORBX_HIP(ctx, hipSetDevice(ctx->device));
  std::vector<int> qi;
  std::vector<float> qx, qy, qr, qxr;
  std::vector<int32_t> qlo, qhi;
  std::vector<uint8_t> qd;
  for (int i = 0; i < nlast; i++) {
    if (!lp_valid[i]) continue;
    const int oct = lp_octave[i];
    if (oct < 0 || oct >= nlevels) return set_err(ctx, ORBX_E_INVALID, "orbx_search_by_projection_last: octave out of range");
    qi.push_back(i);
    qx.push_back(lp_u[i]); qy.push_back(lp_v[i]); qr.push_back(th * scale_factors[oct]);   // :1724
    qxr.push_back(u_right ? lp_u[i] - mbf * lp_invz[i] : 0.f);                              // :1754
    if (direction == 1) { qlo.push_back(oct); qhi.push_back(-1); }                           // :1728-1733
    else if (direction == 2) { qlo.push_back(0); qhi.push_back(oct); }
    else { qlo.push_back(oct - 1); qhi.push_back(oct + 1); }
    qd.insert(qd.end(), lp_desc + (size_t)i * 32, lp_desc + (size_t)i * 32 + 32);
  }
  const int nq = (int)qi.size();
  if (nq == 0) return ORBX_OK;
  std::vector<uint8_t> skip(n);
  for (int i = 0; i < n; i++) skip[i] = kp_obs[i] > 0;
  ArenaScope scope(ctx);
  AreaGates gates;
  gates.kp_skip = skip.data(); gates.kp_uright = u_right; gates.qxr = u_right ? qxr.data() : nullptr; gates.n = n;
  WindowResult w;
  int rc = window_search_core(ctx, kps_un, desc, n, min_x, min_y, max_x, max_y, gates, qx.data(), qy.data(), qr.data(), qlo.data(),
                              qhi.data(), qd.data(), nq, false, w);
  if (rc != ORBX_OK) return rc;
  // host replay in the last frame's keypoint order (:1738-1790), then the rotation filter (:1862-1882)
  const int TH_HIGH = 100, HISTO_LENGTH = 30;
  const float factor = 1.0f / HISTO_LENGTH;
  std::vector<int> rotHist[30];
  int nmatches = 0;
  for (int q = 0; q < nq; q++) {
    const int i = qi[q];
    int bestDist = 256, bestIdx2 = -1;
    for (int c = w.row_ptr[q]; c < w.row_ptr[q + 1]; c++) {
      const int i2 = w.cand[c], d = w.dist[c];
      if (kp_obs[i2] > 0) continue;
      if (d < bestDist) { bestDist = d; bestIdx2 = i2; }
    }
    if (bestDist <= TH_HIGH) {
      kp_match[bestIdx2] = i;
      kp_obs[bestIdx2] = lp_obs[i];
      nmatches++;
      if (check_orientation) {
        float rot = lp_angle[i] - kps_un[bestIdx2].angle;
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)std::round(rot * factor);
        if (bin == HISTO_LENGTH) bin = 0;
        if (bin >= 0 && bin < HISTO_LENGTH) rotHist[bin].push_back(bestIdx2);
      }
    }
  }
  if (check_orientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < HISTO_LENGTH; i++) {  // ComputeThreeMaxima, src/ORBmatcher.cc:2012-2053
      const int s = (int)rotHist[i].size();
      if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
      else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
      else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (int idx : rotHist[i]) { kp_match[idx] = -2; kp_obs[idx] = -1; nmatches--; }
    }
  }
  *nmatches_out = nmatches;
  return ORBX_OK;
}

int orbx_search_by_bow(orbx_ctx* ctx, const uint8_t* kf_desc, const float* kf_angle, const uint8_t* kf_valid, int nkf,
                       const uint32_t* kf_fv_node, const int32_t* kf_fv_ptr, const uint32_t* kf_fv_idx, int n_kf_nodes,
                       const uint8_t* f_desc, const float* f_angle, int nf, const uint32_t* f_fv_node, const int32_t* f_fv_ptr,
                       const uint32_t* f_fv_idx, int n_f_nodes, float nn_ratio, int check_orientation, int32_t* match_kf,
                       int* nmatches_out) {
  if (!ctx || nkf < 0 || nf < 0 || n_kf_nodes < 0 || n_f_nodes < 0 || !nmatches_out || (nf > 0 && (!f_desc || !f_angle || !match_kf)) ||
      (nkf > 0 && (!kf_desc || !kf_angle || !kf_valid)) || (n_kf_nodes > 0 && (!kf_fv_node || !kf_fv_ptr || !kf_fv_idx)) ||
      (n_f_nodes > 0 && (!f_fv_node || !f_fv_ptr || !f_fv_idx)))
    return ctx ? set_err(ctx, ORBX_E_INVALID, "orbx_search_by_bow: bad arguments") : ORBX_E_INVALID;
  *nmatches_out = 0;
  for (int i = 0; i < nf; i++) match_kf[i] = -1;
  if (nkf == 0 || nf == 0 || n_kf_nodes == 0 || n_f_nodes == 0) return ORBX_OK;
  for (int i = 1; i < n_kf_nodes; i++) if (kf_fv_node[i] <= kf_fv_node[i - 1]) return set_err(ctx, ORBX_E_INVALID, "orbx_search_by_bow: keyframe nodes must ascend");
  for (int i = 1; i < n_f_nodes; i++) if (f_fv_node[i] <= f_fv_node[i - 1]) return set_err(ctx, ORBX_E_INVALID, "orbx_search_by_bow: frame nodes must ascend");
  // queries in the reference's order (:244-262): common nodes ascending, the keyframe's features of the node in list order,
  // those with a good map point; candidates = the frame's features of the same node, in list order
  std::vector<int32_t> qkf, row_ptr(1, 0), cand;
  std::vector<uint8_t> qd;
  int a = 0, b = 0;
  while (a < n_kf_nodes && b < n_f_nodes) {
    if (kf_fv_node[a] == f_fv_node[b]) {
      for (int i = kf_fv_ptr[a]; i < kf_fv_ptr[a + 1]; i++) {
        const uint32_t k = kf_fv_idx[i];
        if (k >= (uint32_t)nkf) return set_err(ctx, ORBX_E_INVALID, "orbx_search_by_bow: keyframe feature index out of range");
        if (!kf_valid[k]) continue;
        qkf.push_back((int32_t)k);
        qd.insert(qd.end(), kf_desc + (size_t)k * 32, kf_desc + (size_t)k * 32 + 32);
        for (int j = f_fv_ptr[b]; j < f_fv_ptr[b + 1]; j++) {
          if (f_fv_idx[j] >= (uint32_t)nf) return set_err(ctx, ORBX_E_INVALID, "orbx_search_by_bow: frame feature index out of range");
          cand.push_back((int32_t)f_fv_idx[j]);
        }
        row_ptr.push_back((int32_t)cand.size());
      }
      a++; b++;
    } else if (kf_fv_node[a] < f_fv_node[b]) a++;
    else b++;
  }
  const int nq = (int)qkf.size(), nnz = (int)cand.size();
  if (nq == 0 || nnz == 0) return ORBX_OK;
  ORBX_HIP(ctx, hipSetDevice(ctx->device));
  ArenaScope scope(ctx);
  DBuf<uint8_t> dq, dt;
  DBuf<int32_t> drp, dc, dd;
  ORBX_HIP(ctx, dq.alloc((size_t)nq * 32)); ORBX_HIP(ctx, dt.alloc((size_t)nf * 32)); ORBX_HIP(ctx, drp.alloc(nq + 1));
  ORBX_HIP(ctx, dc.alloc(nnz)); ORBX_HIP(ctx, dd.alloc(nnz));
  hipStream_t st = ctx->stream;
  ORBX_HIP(ctx, hipMemcpyAsync(dq.p, qd.data(), (size_t)nq * 32, hipMemcpyHostToDevice, st));
  ORBX_HIP(ctx, hipMemcpyAsync(dt.p, f_desc, (size_t)nf * 32, hipMemcpyHostToDevice, st));
  ORBX_HIP(ctx, hipMemcpyAsync(drp.p, row_ptr.data(), sizeof(int32_t) * (nq + 1), hipMemcpyHostToDevice, st));
  ORBX_HIP(ctx, hipMemcpyAsync(dc.p, cand.data(), sizeof(int32_t) * nnz, hipMemcpyHostToDevice, st));
  int rc = orbx_nn_csr_device(ctx, dq.p, nq, dt.p, nf, drp.p, dc.p, 0, nullptr, nullptr, nullptr, nullptr, dd.p, st);
  if (rc != ORBX_OK) return rc;
  std::vector<int32_t> dist(nnz);
  ORBX_HIP(ctx, hipMemcpyAsync(dist.data(), dd.p, sizeof(int32_t) * nnz, hipMemcpyDeviceToHost, st));
  ORBX_HIP(ctx, hipStreamSynchronize(st));
  // host replay (:264-330): a frame feature matched by an earlier keyframe feature is no candidate any more
  const int TH_LOW = 50, HISTO_LENGTH = 30;
  const float factor = 1.0f / HISTO_LENGTH;
  std::vector<int> rotHist[30];
  int nmatches = 0;
  for (int q = 0; q < nq; q++) {
    int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
    for (int c = row_ptr[q]; c < row_ptr[q + 1]; c++) {
      const int iF = cand[c], d = dist[c];
      if (match_kf[iF] >= 0) continue;
      if (d < bestDist1) { bestDist2 = bestDist1; bestDist1 = d; bestIdxF = iF; }
      else if (d < bestDist2) { bestDist2 = d; }
    }
    if (bestDist1 <= TH_LOW && (float)bestDist1 < nn_ratio * (float)bestDist2) {
      match_kf[bestIdxF] = qkf[q];
      if (check_orientation) {
        float rot = kf_angle[qkf[q]] - f_angle[bestIdxF];
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)std::round(rot * factor);
        if (bin == HISTO_LENGTH) bin = 0;
        if (bin >= 0 && bin < HISTO_LENGTH) rotHist[bin].push_back(bestIdxF);
      }
      nmatches++;
    }
  }
  if (check_orientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < HISTO_LENGTH; i++) {  // ComputeThreeMaxima, src/ORBmatcher.cc:2012-2053
      const int s = (int)rotHist[i].size();
      if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
      else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
      else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (int idx : rotHist[i]) { match_kf[idx] = -1; nmatches--; }
    }
  }
  *nmatches_out = nmatches;
  return ORBX_OK;
}

int orbx_stereo_matches(orbx_ctx* left, orbx_ctx* right, const orbx_keypoint* kpsL, const uint8_t* descL, int nL,
                        const orbx_keypoint* kpsR, const uint8_t* descR, int nR, float mb, float mbf, float* u_right, float* depth,
                        int* nmatches) {
  if (!left || !right || nL < 0 || nR < 0 || (nL > 0 && (!kpsL || !descL || !u_right || !depth)) || (nR > 0 && (!kpsR || !descR)) ||
      !(mb > 0))
    return left ? set_err(left, ORBX_E_INVALID, "orbx_stereo_matches: bad arguments") : ORBX_E_INVALID;
  orbx_ctx* ctx = left;
  if (nmatches) *nmatches = 0;
  for (int i = 0; i < nL; i++) { u_right[i] = -1.0f; depth[i] = -1.0f; }
  if (nL == 0 || nR == 0) return ORBX_OK;
  if (!left->d_geo || !right->d_geo || left->last_nframes < 1 || right->last_nframes < 1 || left->device != right->device ||
      left->geo.rows != right->geo.rows || left->geo.cols != right->geo.cols || left->nlevels != right->nlevels)
    return set_err(ctx, ORBX_E_INVALID, "orbx_stereo_matches: both extractors must have processed a frame of the same shape on one GPU");
  ORBX_HIP(ctx, hipSetDevice(ctx->device));
  // both pyramids may still be in flight (context stream, aux streams, or the caller's stream of a device-resident batch)
  ORBX_HIP(ctx, sync_ctx(left));
  ORBX_HIP(ctx, sync_ctx(right));
  StereoGeom sg;
  std::memset(&sg, 0, sizeof(sg));
  sg.nlevels = left->nlevels;
  for (int l = 0; l < left->nlevels; l++) {
    const LevelGeom& L = left->geo.lv[l];
    const LevelGeom& R = right->geo.lv[l];
    sg.w[l] = L.w; sg.h[l] = L.h;
    if (l == 0) {
      sg.left[l] = left->last_imgs; sg.pitchL[l] = (int)left->last_row_stride;
      sg.right[l] = right->last_imgs; sg.pitchR[l] = (int)right->last_row_stride;
    } else {
      sg.left[l] = left->d_pyr + L.plane_off; sg.pitchL[l] = L.pitch;
      sg.right[l] = right->d_pyr + R.plane_off; sg.pitchR[l] = R.pitch;
    }
    sg.scale[l] = left->scale[l]; sg.inv_scale[l] = left->inv_scale[l];
  }
  ArenaScope scope(ctx);
  hipStream_t st = ctx->stream;
  // one packed blob in pinned memory (keypoints and descriptors of both sides), pulled into HBM by a copy kernel; the filter kernel
  // writes the two result arrays into the mapped output and the host polls a done word — no copy node, no stream synchronisation
  // ("window_direct" = 0 selects copies + synchronisation instead)
  BlobLayout in, out;
  const size_t o_kl = in.add(sizeof(orbx_keypoint) * (size_t)nL), o_kr = in.add(sizeof(orbx_keypoint) * (size_t)nR),
               o_dl = in.add((size_t)nL * 32), o_dr = in.add((size_t)nR * 32);
  const size_t p_u = out.add(sizeof(float) * (size_t)nL), p_d = out.add(sizeof(float) * (size_t)nL), p_done = out.add(16);
  uint8_t* h = nullptr;
  ORBX_HIP(ctx, host_stage(ctx, in.size + out.size, &h));
  std::memcpy(h + o_kl, kpsL, sizeof(orbx_keypoint) * (size_t)nL);
  std::memcpy(h + o_kr, kpsR, sizeof(orbx_keypoint) * (size_t)nR);
  std::memcpy(h + o_dl, descL, (size_t)nL * 32);
  std::memcpy(h + o_dr, descR, (size_t)nR * 32);
  DBuf<uint8_t> din;
  DBuf<float> du, dd;
  DBuf<int32_t> dsad, dkept;
  ORBX_HIP(ctx, din.alloc(in.size));
  ORBX_HIP(ctx, du.alloc(nL)); ORBX_HIP(ctx, dd.alloc(nL)); ORBX_HIP(ctx, dsad.alloc(nL)); ORBX_HIP(ctx, dkept.alloc(1));
  const orbx_keypoint* dkl = (const orbx_keypoint*)(din.p + o_kl);
  const orbx_keypoint* dkr = (const orbx_keypoint*)(din.p + o_kr);
  uint8_t* hdev = nullptr;
  const bool direct = ctx->window_direct && hipHostGetDevicePointer((void**)&hdev, h, 0) == hipSuccess && hdev != nullptr;
  if (!direct) (void)hipGetLastError();
  int32_t kept = 0;
  if (direct) {
    uint8_t* hout = h + in.size;
    volatile unsigned long long* done = (volatile unsigned long long*)(hout + p_done);
    __atomic_store_n(done, 0ull, __ATOMIC_RELEASE);
    const int n16 = (int)((in.size + 15) / 16);
    hipLaunchKernelGGL(k_stereo_stage_in, dim3(std::min((n16 + 255) / 256, 512)), dim3(256), 0, st, (const uint4*)hdev, (uint4*)din.p, n16);
    hipLaunchKernelGGL(k_stereo_match, dim3((nL + 3) / 4), dim3(256), 0, st, sg, dkl, din.p + o_dl, nL, dkr, din.p + o_dr, nR, mb, mbf, du.p, dd.p, dsad.p);
    hipLaunchKernelGGL(k_stereo_filter, dim3(1), dim3(1024), 0, st, nL, du.p, dd.p, dsad.p, dkept.p, (float*)(hdev + in.size + p_u),
                       (float*)(hdev + in.size + p_d), (unsigned long long*)(hdev + in.size + p_done));
    ORBX_HIP(ctx, hipGetLastError());
    unsigned long long word = 0;
    for (unsigned spin = 1;; spin++) {
      if ((word = __atomic_load_n(done, __ATOMIC_ACQUIRE)) != 0) break;
      if ((spin & 0x3fff) == 0) {
        const hipError_t qe = hipStreamQuery(st);
        if (qe == hipSuccess) {
          if ((word = __atomic_load_n(done, __ATOMIC_ACQUIRE)) != 0) break;
          return set_err(ctx, ORBX_E_DEVICE, "orbx_stereo_matches: the pass finished without publishing its results");
        }
        if (qe != hipErrorNotReady) { ORBX_HIP(ctx, qe); }
      }
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
    }
    kept = (int32_t)(uint32_t)word;
    std::memcpy(u_right, hout + p_u, sizeof(float) * (size_t)nL);
    std::memcpy(depth, hout + p_d, sizeof(float) * (size_t)nL);
  } else {
    ORBX_HIP(ctx, hipMemcpyAsync(din.p, h, in.size, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_stereo_match, dim3((nL + 3) / 4), dim3(256), 0, st, sg, dkl, din.p + o_dl, nL, dkr, din.p + o_dr, nR, mb, mbf, du.p, dd.p, dsad.p);
    hipLaunchKernelGGL(k_stereo_filter, dim3(1), dim3(1024), 0, st, nL, du.p, dd.p, dsad.p, dkept.p, (float*)nullptr, (float*)nullptr,
                       (unsigned long long*)nullptr);
    ORBX_HIP(ctx, hipGetLastError());
    ORBX_HIP(ctx, hipMemcpyAsync(u_right, du.p, sizeof(float) * nL, hipMemcpyDeviceToHost, st));
    ORBX_HIP(ctx, hipMemcpyAsync(depth, dd.p, sizeof(float) * nL, hipMemcpyDeviceToHost, st));
    ORBX_HIP(ctx, hipMemcpyAsync(&kept, dkept.p, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    ORBX_HIP(ctx, hipStreamSynchronize(st));
  }
  if (nmatches) *nmatches = kept;
  return ORBX_OK;
}

}  // extern "C"
