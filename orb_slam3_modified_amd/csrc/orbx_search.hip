// orbx guided search: the frame grid (Frame::AssignFeaturesToGrid / GetFeaturesInArea, src/Frame.cc:385-416,
// :657-735) built and queried on the GPU, and ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:648-763) on
// top of it: candidate generation + all Hamming distances on the device, the order-dependent greedy bookkeeping
// (queries are NOT independent in this routine) replayed on the host in the original query order.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <vector>

#include "orbx_internal.h"

namespace orbx {

constexpr int kGridCols = 64, kGridRows = 48, kGridCells = kGridCols * kGridRows;  // include/Frame.h:44-45
constexpr int kGridMaxPts = 8192;

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
                                                          int32_t* __restrict__ cand) {
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

template <typename T>
struct DBuf {
  T* p = nullptr;
  ~DBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t n) { return hipMalloc((void**)&p, std::max<size_t>(n, 1) * sizeof(T)); }
};

struct GridOnDevice {
  DBuf<orbx_keypoint> kps;
  DBuf<uint16_t> sorted;
  DBuf<int32_t> cell_start;
  float minX = 0, minY = 0, invW = 0, invH = 0;
};

static int build_grid(orbx_ctx* ctx, const orbx_keypoint* kps, int n, float min_x, float min_y, float max_x, float max_y,
                      GridOnDevice& g) {
  if (n > kGridMaxPts) return set_err(ctx, ORBX_E_CAPACITY, "frame grid: more than 8192 keypoints");
  g.minX = min_x; g.minY = min_y;
  g.invW = (float)kGridCols / (float)(max_x - min_x);   // src/Frame.cc:159-160
  g.invH = (float)kGridRows / (float)(max_y - min_y);
  ORBX_HIP(ctx, g.kps.alloc(n)); ORBX_HIP(ctx, g.sorted.alloc(n)); ORBX_HIP(ctx, g.cell_start.alloc(kGridCells + 1));
  if (n) ORBX_HIP(ctx, hipMemcpyAsync(g.kps.p, kps, sizeof(orbx_keypoint) * n, hipMemcpyHostToDevice, ctx->stream));
  int npad = 2;
  while (npad < n) npad <<= 1;
  hipLaunchKernelGGL(k_frame_grid, dim3(1), dim3(1024), (size_t)npad * 4, ctx->stream, g.kps.p, n, g.minX, g.minY, g.invW, g.invH,
                     npad, g.sorted.p, g.cell_start.p);
  ORBX_HIP(ctx, hipGetLastError());
  return ORBX_OK;
}

struct AreaQueries {
  DBuf<float> x, y, r;
  DBuf<int32_t> lmin, lmax, counts, row_ptr, cand;
  int nq = 0, nnz = 0;
};

// count -> scan -> fill; leaves row_ptr / cand on the device, nnz on the host
static int run_area(orbx_ctx* ctx, const GridOnDevice& g, const float* qx, const float* qy, const float* qr, const int32_t* qmin,
                    const int32_t* qmax, int nq, AreaQueries& a) {
  a.nq = nq;
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
                     g.invH, a.x.p, a.y.p, a.r.p, a.lmin.p, a.lmax.p, nq, a.counts.p, (const int32_t*)nullptr, (int32_t*)nullptr);
  hipLaunchKernelGGL(k_scan_excl, dim3(1), dim3(1024), 0, ctx->stream, a.counts.p, nq, a.row_ptr.p);
  ORBX_HIP(ctx, hipGetLastError());
  int32_t nnz = 0;
  ORBX_HIP(ctx, hipMemcpyAsync(&nnz, a.row_ptr.p + nq, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  ORBX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  a.nnz = nnz;
  ORBX_HIP(ctx, a.cand.alloc(nnz));
  if (nnz) {
    hipLaunchKernelGGL(k_features_in_area<true>, grid, block, 0, ctx->stream, g.kps.p, g.sorted.p, g.cell_start.p, g.minX, g.minY, g.invW,
                       g.invH, a.x.p, a.y.p, a.r.p, a.lmin.p, a.lmax.p, nq, (int32_t*)nullptr, a.row_ptr.p, a.cand.p);
    ORBX_HIP(ctx, hipGetLastError());
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

}  // extern "C"
