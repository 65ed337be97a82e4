// orbx extractor kernels for gfx950 (CDNA4, wave64).  Integer/bitwise work, no MFMA.
//
//   k_resize      : one pyramid level from the previous one (OpenCV INTER_LINEAR 8u fixed point), 64x64 output tiles,
//                   source rectangle staged in LDS.
//   k_fast_cells  : per (frame, 35-px cell): LDS-staged tile -> necessary test on packed 16-bit lanes -> exact FAST-9/16
//                   score of the survivors -> 3x3 NMS inside the cell -> iniTh/minTh selection -> ordered compaction
//                   (bitmap + popcount prefix).
//   k_quadtree    : per (frame, level): DistributeOctTree as flat node list + segment partition (ballot ranks),
//                   libstdc++-exact sort for the tie order; node arrays in LDS, or in HBM for huge level quotas.
//   k_assemble    : per frame: output slot of every keypoint (mono side ascending / lapping side descending).
//   k_blur7       : 7x7 fixed-point Gaussian of every level, 64x58 tiles, v_dot4_u32_u8 horizontal and v_dot2_u32_u16
//                   vertical pass.
//   k_describe    : K keypoints per wave: IC moments from dword rows (v_dot4 with byte masks, DPP reduction) -> angle,
//                   cos/sin once per K; 512 taps gathered from the blurred level -> 256 steered tests, one wave ballot =
//                   8 descriptor bytes.
//   k_color_to_gray: cv::cvtColor(...2GRAY) behind the upload.
//
// Float code relies on -ffp-contract=off (no FMA fusion) and IEEE division; see DESIGN.md "bit-exactness".
#include <hip/hip_runtime.h>

#include "glibc_sincosf.h"
#include "gnu_sort.h"
#include "orbx_internal.h"

namespace orbx {

__constant__ __align__(16) int8_t c_pattern[1024] = {
#include "orb_pattern.inc"
};

// n / d for block-uniform operands without the ~25-instruction VALU division sequence: M = ceil(2^32 / d) from the
// host (0 encodes d == 1); exact while n * d < 2^32 (checked on the host for every use).
__device__ __forceinline__ int fast_div(int n, uint32_t M) { return M ? (int)__umulhi((uint32_t)n, M) : n; }

// Full-rate 24-bit multiply, pinned with inline asm: the compiler re-widens __mul24 to the quarter-rate v_mul_lo_u32
// whenever it cannot prove the operand ranges itself.  Both operands must fit 24 signed bits.
// CAUTION (measured the hard way): the hazard recogniser does not look inside inline asm.  Feed these helpers only values
// produced by plain VALU arithmetic — never the direct result of v_dot4 / v_dot2, a transcendental or a DPP / readlane op,
// whose consumers need wait states the compiler will not insert in front of an asm statement.
__device__ __forceinline__ int mul_i24(int a, int b) {
  int r;
  asm("v_mul_i32_i24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__device__ __forceinline__ uint32_t mul_u24(uint32_t a, uint32_t b) {   // low 32 bits of a 24 x 24-bit product, full rate
  uint32_t r;
  asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// XCD-aware block order.  Workgroup b of a 1-D grid runs on XCD b % 8 (observed dispatch rule, used for speed only)
// and every XCD has its own 4 MiB L2, so neighbouring tiles / cells of one frame should share an XCD: the grid is
// padded to 8*chunk blocks and block b works on logical item (b % 8) * chunk + b / 8, i.e. XCD k owns the contiguous
// logical range [k*chunk, (k+1)*chunk) = a contiguous run of whole frames.  Returns -1 for the padding blocks.
__device__ __forceinline__ int xcd_logical_block(int n_items) {
  const int chunk = (int)(gridDim.x >> 3);
  const int L = (int)(blockIdx.x & 7u) * chunk + (int)(blockIdx.x >> 3);
  return L < n_items ? L : -1;
}

// The single-frame path's upload as a kernel: rows of `row16` 16-byte units from the caller's image in mapped pinned host memory (tight
// rows) into the staging image (pitch `dst_pitch`).  A copy NODE costs 11.6 us of DMA for 300 KB plus a 7.7 us hand-over to the first
// kernel; wide loads from a kernel pull the same bytes over PCIe without the hand-over.
__global__ __launch_bounds__(256) void k_upload_rows(const uint4* __restrict__ src, uint8_t* __restrict__ dst, int rows, int row16, int dst_pitch) {
  const int n = rows * row16;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int r = i / row16, c = i - r * row16;
    *(uint4*)(dst + (size_t)r * dst_pitch + 16 * c) = src[i];
  }
}

// ------------------------------------------------------------------------------------------------
// K1: cv::resize INTER_LINEAR, CV_8UC1 (src/ORBextractor.cc:1183), one pyramid level from the previous one.
// The byte-gather formulation is bound by the texture-address unit (one VMEM instruction per source byte), so a
// workgroup stages the source rectangle of its 64x64 output tile into LDS with aligned dword loads and gathers the
// four bilinear taps of every output pixel from LDS; each lane produces 4 horizontally adjacent pixels of 4
// consecutive rows (the decoded column table is reused, one dword store per row).  Arithmetic: int32 fixed point exactly as OpenCV.  A source whose rows are not dword
// aligned is staged with byte loads instead (same LDS layout, same results).
// ------------------------------------------------------------------------------------------------
constexpr int kRT_W = 64, kRT_H = 64, kRT_RPT = 4;  // output tile of k_resize; rows per thread

// one 64x64 output tile (logical item L of the launch: frame-major, then tile rows); 256 threads; `smem` = the workgroup's dynamic LDS
__device__ __forceinline__ void resize_tile(uint8_t* smem, const int L, const uint8_t* __restrict__ src, long long src_frame_stride,
                                            int src_pitch, int sw, uint8_t* __restrict__ dst,
                                            long long dst_frame_stride, int dst_pitch, int dw, int dh,
                                            const XTab* __restrict__ xt, const XTab* __restrict__ yt, int nbx,
                                            int nby, int lds_pitch, int lds_rows, uint32_t m_tiles,
                                            uint32_t m_nbx, uint32_t m_lp4, int nph) {
  const int t = threadIdx.x;
  const int frame = fast_div(L, m_tiles), rem = L - frame * (nbx * nby);
  const int by = fast_div(rem, m_nbx), bx = rem - by * nbx;
  const int x0 = bx * kRT_W, y0 = by * kRT_H;
  const int xl = min(x0 + kRT_W, dw) - 1, yl = min(y0 + kRT_H, dh) - 1;  // last output column / row of the tile
  // source rectangle (tables are monotonic): columns [X0, X1], rows [Y0, Y1]
  const XTab txa = xt[x0], txb = xt[xl], tya = yt[y0], tyb = yt[yl];
  const int X0 = (int)txa.s0 & ~3, X1 = max((int)txb.s0, (int)txb.s1);
  const int Y0 = tya.s0, Y1 = max((int)tyb.s0, (int)tyb.s1);
  const int nrows = Y1 - Y0 + 1, ncolb = X1 - X0 + 1;
  const uint8_t* S = src + (long long)frame * src_frame_stride + (long long)Y0 * src_pitch + X0;
  const bool al = ((src_pitch & 3) == 0) && ((((unsigned long long)src) & 3) == 0) && ((src_frame_stride & 3) == 0);
  if (nrows > lds_rows || ncolb > lds_pitch) return;  // host sized the tile from the same tables: cannot happen
  if (al) {
    // whole LDS rows (lds_pitch >= the widest source rectangle of the level): LDS dword index == loop index
    // thread = (dword column c, row phase of nph = 256 / lp4): the column-only work once, then down the column nph rows at a time
    const int lp4 = lds_pitch >> 2, swr = (sw + 3) & ~3;
    const int rph = fast_div(t, m_lp4), c = t - rph * lp4;
    // LDS-DMA loads: the LDS dword index of (row rph + k nph, column c) is t + k nph lp4 — lane-linear, as global_load_lds writes
    // (wave-uniform base + lane * 4); no staging registers, no ds_write pass
    {
      const bool mine = rph < nph && X0 + 4 * c < swr;
      const uint8_t* sp = S + 4 * c;
      const int stride = __mul24(nph, lp4);
      for (int r0 = 0; r0 < nrows; r0 += nph) {   // block-uniform trip count
        const int r = r0 + rph;
        uint32_t* dp = (uint32_t*)smem + __mul24(r0, lp4) + (t & ~63);
        (void)stride;
        if (mine && r < nrows)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sp + (uint32_t)__mul24(r, src_pitch)),
                                           (__attribute__((address_space(3))) void*)dp, 4, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  } else {
#pragma unroll 1  // cold path: keep it out of the register budget
    for (int r = t >> 6; r < nrows; r += 4) {
#pragma unroll 1
      for (int c = t & 63; c < ncolb; c += 64) smem[r * lds_pitch + c] = S[(long long)r * src_pitch + c];
    }
  }
  __syncthreads();
  const int x4 = x0 + (t & 15) * 4, dy0 = y0 + (t >> 4) * kRT_RPT;
  if (dy0 >= dh || x4 >= dw) return;
  XTab tx[4];
  if (x4 + 3 < dw) {
    const uint4 q0 = *(const uint4*)(xt + x4), q1 = *(const uint4*)(xt + x4 + 2);
    tx[0] = *(const XTab*)&q0.x; tx[1] = *(const XTab*)&q0.z; tx[2] = *(const XTab*)&q1.x; tx[3] = *(const XTab*)&q1.z;
  } else {
#pragma unroll
    for (int i = 0; i < 4; i++) tx[i] = xt[min(x4 + i, dw - 1)];
  }
  int c0[4], c1[4], a0[4], a1[4];
#pragma unroll
  for (int i = 0; i < 4; i++) { c0[i] = (int)tx[i].s0 - X0; c1[i] = (int)tx[i].s1 - X0; a0[i] = tx[i].a0; a1[i] = tx[i].a1; }
  uint8_t* outp = dst + (long long)frame * dst_frame_stride;  // uniform; per-lane offsets stay 32-bit (plane < 2^31 bytes)
#pragma unroll
  for (int rr = 0; rr < kRT_RPT; rr++) {
    const int dy = dy0 + rr;
    if (dy < dh) {
      const XTab ty = yt[dy];
      const uint8_t* R0 = smem + __mul24((int)ty.s0 - Y0, lds_pitch);
      const uint8_t* R1 = smem + __mul24((int)ty.s1 - Y0, lds_pitch);
      const int b0 = ty.a0, b1 = ty.a1;
      uint32_t packed = 0;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        // every factor fits 24 bits (bytes, 12-bit weights, h >> 4 < 2^15): full-rate v_mul/mad_i32_i24, not the
        // quarter-rate 32-bit multiply
        const int h0 = __mul24((int)R0[c0[i]], a0[i]) + __mul24((int)R0[c1[i]], a1[i]);
        const int h1 = __mul24((int)R1[c0[i]], a0[i]) + __mul24((int)R1[c1[i]], a1[i]);
        const int v = (((mul_i24(b0, h0 >> 4) >> 16) + (mul_i24(b1, h1 >> 4) >> 16) + 2) >> 2) & 0xff;
        packed |= (uint32_t)v << (8 * i);
      }
      *(uint32_t*)(outp + (uint32_t)(__mul24(dy, dst_pitch) + x4)) = packed;
    }
  }
}

__global__ __launch_bounds__(256) void k_resize(const uint8_t* __restrict__ src, long long src_frame_stride,
                                                int src_pitch, int sw, uint8_t* __restrict__ dst,
                                                long long dst_frame_stride, int dst_pitch, int dw, int dh,
                                                const XTab* __restrict__ xt, const XTab* __restrict__ yt, int nbx,
                                                int nby, int nitems, int lds_pitch, int lds_rows, uint32_t m_tiles,
                                                uint32_t m_nbx, uint32_t m_lp4, int nph) {
  extern __shared__ __align__(16) uint8_t smem[];
  const int L = xcd_logical_block(nitems);
  if (L < 0) return;  // block-uniform
  resize_tile(smem, L, src, src_frame_stride, src_pitch, sw, dst, dst_frame_stride, dst_pitch, dw, dh, xt, yt, nbx, nby, lds_pitch, lds_rows,
                     m_tiles, m_nbx, m_lp4, nph);
}

// Batch frames whose rows are not dword-aligned (odd strides: KITTI's 1241- and 1226-px rows, ROI views) into an aligned copy: every kernel
// that reads level 0 — the first resize, FAST, blur, the descriptors' orientation patch — has a slow byte-wise path for such sources
// (measured on 1241 x 376: resize +46 %, blur +40 %, FAST +14 %, the step +17 %), and one pass over the pixels costs less than that.
// grid = (dwords of a row / 256, rows, frames); the padding bytes of the copy are zero.
__global__ __launch_bounds__(256) void k_realign_rows(const uint8_t* __restrict__ src, long long src_row_stride, long long src_frame_stride,
                                                      uint8_t* __restrict__ dst, int dst_pitch, int rows, int cols) {
  const int c4 = (int)(blockIdx.x * 256 + threadIdx.x), r = (int)blockIdx.y, f = (int)blockIdx.z;
  if (4 * c4 >= dst_pitch) return;
  const uint8_t* s = src + (long long)f * src_frame_stride + (long long)r * src_row_stride + 4 * c4;
  uint32_t v = 0;
  if (4 * c4 + 3 < cols) __builtin_memcpy(&v, s, 4);   // no alignment promised: the compiler picks what the target guarantees
  else
#pragma unroll
    for (int j = 0; j < 4; j++) v |= (uint32_t)((4 * c4 + j < cols) ? s[j] : 0) << (8 * j);
  *(uint32_t*)(dst + ((long long)f * rows + r) * dst_pitch + 4 * c4) = v;
}

// ------------------------------------------------------------------------------------------------
// K1, small batches: K consecutive pyramid levels in ONE launch.  Inside the single-frame graph every launch costs about 5 us whatever
// it does, and the seven dependent resizes were 38 us of a 150 us frame.  Here a workgroup owns one 64x64 tile of the LAST level of
// the group and computes, in LDS, the rectangles of the intermediate levels that tile needs (at scale 1.2 a halo of a few pixels: 6 %
// redundant work per level), writing them to the pyramid on the way.  Neighbouring workgroups write the overlap of their rectangles
// with identical bytes (every pixel is the same function of the same source pixels); rectangles are dword-aligned so that no store
// touches a byte its workgroup did not compute; together the rectangles cover every pixel of the intermediate levels (the host
// checks this on the tables before it takes this path).  Same arithmetic as resize_tile: bytes identical to the separate launches.
// ------------------------------------------------------------------------------------------------
struct ChainLevel { const XTab* xt; const XTab* yt; uint8_t* dst; int pitch, w, h; };
template <int K>
struct ChainArgs {
  const uint8_t* src; long long src_frame_stride, dst_frame_stride; int src_pitch, sw;
  ChainLevel lv[K];          // destination levels, in order
  int nbx, nby, tile;        // tiles of the last level (tile x tile pixels: kRT_W, or smaller for long chains of small levels)
  int buf_off[K], buf_pitch[K];   // LDS rectangles: [0] = source, [k] = destination level k - 1 of the group (k < K)
};

__device__ __forceinline__ uint32_t chain_px4(const uint8_t* __restrict__ prev, int pp, int ox, int oy, const XTab* __restrict__ xt,
                                              const XTab ty, int x4, int w) {
  const uint8_t* R0 = prev + ((int)ty.s0 - oy) * pp;
  const uint8_t* R1 = prev + ((int)ty.s1 - oy) * pp;
  const int b0 = ty.a0, b1 = ty.a1;
  uint32_t packed = 0;
  XTab txs[4];
  if (x4 + 3 < w) {   // the level tables are 32-byte aligned and x4 is a multiple of 4: two 16-byte loads
    const uint4 q0 = *(const uint4*)(xt + x4), q1 = *(const uint4*)(xt + x4 + 2);
    txs[0] = *(const XTab*)&q0.x; txs[1] = *(const XTab*)&q0.z; txs[2] = *(const XTab*)&q1.x; txs[3] = *(const XTab*)&q1.z;
  } else {
#pragma unroll
    for (int i = 0; i < 4; i++) txs[i] = xt[min(x4 + i, w - 1)];
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const XTab tx = txs[i];
    const int c0 = (int)tx.s0 - ox, c1 = (int)tx.s1 - ox, a0 = tx.a0, a1 = tx.a1;
    const int h0 = (int)R0[c0] * a0 + (int)R0[c1] * a1;
    const int h1 = (int)R1[c0] * a0 + (int)R1[c1] * a1;
    const int v = ((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2) & 0xff;
    packed |= (uint32_t)v << (8 * i);
  }
  return packed;
}

template <int K>
__global__ __launch_bounds__(1024) void k_resize_chain(const ChainArgs<K> a) {
  extern __shared__ __align__(16) uint8_t smem[];
  const int t = threadIdx.x, T = blockDim.x, frame = blockIdx.y;
  const int by = (int)blockIdx.x / a.nbx, bx = (int)blockIdx.x - by * a.nbx;
  // rectangles, inclusive: [k] for k = K (the tile of the last level) down to 0 (the source); columns of k < K span whole dwords
  int X0[K + 1], X1[K + 1], Y0[K + 1], Y1[K + 1];
  X0[K] = bx * a.tile; X1[K] = min(bx * a.tile + a.tile, a.lv[K - 1].w) - 1;
  Y0[K] = by * a.tile; Y1[K] = min(by * a.tile + a.tile, a.lv[K - 1].h) - 1;
#pragma unroll
  for (int k = K; k >= 1; k--) {
    const ChainLevel& L = a.lv[k - 1];
    const XTab ta = L.xt[X0[k]], tb = L.xt[min(X1[k], L.w - 1)], ua = L.yt[Y0[k]], ub = L.yt[Y1[k]];
    X0[k - 1] = (int)ta.s0 & ~3; X1[k - 1] = max((int)tb.s0, (int)tb.s1) | 3;
    Y0[k - 1] = ua.s0; Y1[k - 1] = max((int)ub.s0, (int)ub.s1);
  }
  // ---- stage the source rectangle
  {
    const uint8_t* S = a.src + (long long)frame * a.src_frame_stride + (long long)Y0[0] * a.src_pitch + X0[0];
    uint8_t* B = smem + a.buf_off[0];
    const int bp = a.buf_pitch[0], nd = (X1[0] - X0[0] + 1) >> 2, nr = Y1[0] - Y0[0] + 1, swr = (a.sw + 3) & ~3;
    const bool al = ((a.src_pitch & 3) == 0) && ((((unsigned long long)a.src) & 3) == 0) && ((a.src_frame_stride & 3) == 0);
    const uint32_t m = nd > 1 ? (uint32_t)(((1ull << 32) + nd - 1) / nd) : 0u;
    for (int i = t; i < nd * nr; i += T) {
      const int r = fast_div(i, m), c = i - r * nd;
      uint32_t v = 0;
      if (X0[0] + 4 * c < swr) {
        if (al) v = *(const uint32_t*)(S + (long long)r * a.src_pitch + 4 * c);
        else {
          const uint8_t* q = S + (long long)r * a.src_pitch + 4 * c;
#pragma unroll
          for (int j = 0; j < 4; j++) v |= (uint32_t)((X0[0] + 4 * c + j < a.sw) ? q[j] : 0) << (8 * j);
        }
      }
      *(uint32_t*)(B + r * bp + 4 * c) = v;
    }
  }
  __syncthreads();
  // ---- level by level
#pragma unroll
  for (int k = 1; k <= K; k++) {
    const ChainLevel& L = a.lv[k - 1];
    const uint8_t* prev = smem + a.buf_off[k - 1];
    const int pp = a.buf_pitch[k - 1];
    const int nd = (X1[k] - X0[k] + 4) >> 2, nr = Y1[k] - Y0[k] + 1;   // k == K: the tile may end inside a dword (the pad bytes repeat the last pixel)
    uint8_t* outp = L.dst + (long long)frame * a.dst_frame_stride;
    const uint32_t m = nd > 1 ? (uint32_t)(((1ull << 32) + nd - 1) / nd) : 0u;
    for (int i = t; i < nd * nr; i += T) {
      const int r = fast_div(i, m), c = i - r * nd;
      const int y = Y0[k] + r, x4 = X0[k] + 4 * c;
      const uint32_t packed = chain_px4(prev, pp, X0[k - 1], Y0[k - 1], L.xt, L.yt[y], x4, L.w);
      if (x4 < ((L.w + 3) & ~3)) *(uint32_t*)(outp + (long long)y * L.pitch + x4) = packed;
      if (k < K) *(uint32_t*)(smem + a.buf_off[k] + r * a.buf_pitch[k] + 4 * c) = packed;
    }
    if (k < K) __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// K2: FAST score + per-cell NMS + threshold selection + ordered compaction.
// ------------------------------------------------------------------------------------------------
// S(p) = max over the 16 nine-pixel arcs of min(v - p_k), and the same for (p_k - v), minus 1:
// the value OpenCV's cornerScore<16> returns for any threshold at which p is a corner.
// With d_k = v - p_k:  max_arc min d = v - min_arc max p  and  min_arc max d = v - max_arc min p, so the score is
// max(v - minmax9(p), maxmin9(p) - v) - 1 and needs no subtraction per circle pixel; a 9-arc extremum is
// 3-input extrema of 3-input extrema (v_min3_u32 / v_max3_u32): 16 + 16 + 8 instructions per polarity.
__device__ __forceinline__ uint32_t umin3(uint32_t a, uint32_t b, uint32_t c) { return min(min(a, b), c); }
__device__ __forceinline__ uint32_t umax3(uint32_t a, uint32_t b, uint32_t c) { return max(max(a, b), c); }

__device__ __forceinline__ int fast_score16(int v, const uint32_t (&p)[16]) {
  uint32_t M3[16], m3[16];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    M3[k] = umax3(p[k], p[(k + 1) & 15], p[(k + 2) & 15]);
    m3[k] = umin3(p[k], p[(k + 1) & 15], p[(k + 2) & 15]);
  }
  uint32_t M9[16], m9[16];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    M9[k] = umax3(M3[k], M3[(k + 3) & 15], M3[(k + 6) & 15]);
    m9[k] = umin3(m3[k], m3[(k + 3) & 15], m3[(k + 6) & 15]);
  }
  uint32_t minmax = umin3(M9[0], M9[1], M9[2]), maxmin = umax3(m9[0], m9[1], m9[2]);
#pragma unroll
  for (int k = 3; k < 15; k += 2) {
    minmax = umin3(minmax, M9[k], M9[k + 1]);
    maxmin = umax3(maxmin, m9[k], m9[k + 1]);
  }
  minmax = min(minmax, M9[15]);
  maxmin = max(maxmin, m9[15]);
  return max(v - (int)minmax, (int)maxmin - v) - 1;
}

// Work-efficient structure (the kernel is VALU-bound): only ~5 % of the pixels are corners at minTh, so
// (1) every pixel takes a 9-value necessary test (four opposite pairs of the circle — the compass and the diagonal
//     ones: a 9-arc always contains one pixel of each opposite pair), evaluated for 4 horizontally adjacent pixels per
//     lane from 11 aligned LDS dwords (byte windows via v_alignbyte), no per-pixel index arithmetic; it passes ~6 %
//     of the pixels (two pairs alone pass ~11 %), which halves the exact-score work;
// (2) the survivors are compacted into an LDS list of packed (y, x) and only they pay for the exact
//     16-pixel score; (3) NMS and the iniTh/minTh selection run on the list; (4) the ordered (row-major) output
//     order is rebuilt from a bitmap + popcount prefix instead of a pass over all pixels.
// list entry: x | y << 7 (detection-domain coordinates, both < 128), bit 15 = NMS survivor.
typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2_t as_u16x2(uint32_t x) { u16x2_t r; __builtin_memcpy(&r, &x, 4); return r; }
__device__ __forceinline__ uint32_t as_u32(u16x2_t x) { uint32_t r; __builtin_memcpy(&r, &x, 4); return r; }

// The necessary test of stage B for TWO pixels at once on packed 16-bit lanes (v_pk_min/max_u16, saturating v_pk_add/sub).
// Every operand carries its pixel in the HIGH byte of each 16-bit lane; the low byte may hold anything: min / max of such
// lanes order by the high byte first, and the two threshold comparisons are arranged so that the low byte cannot flip them
// (dark: (v - t) << 8 > lane  <=>  v - t > pixel;  bright: lane > ((v + t) << 8 | 0xff)  <=>  pixel > v + t, with v + t saturating).
// Returns a dword whose 16-bit lanes are non-zero exactly for the pixels that pass.
__device__ __forceinline__ uint32_t fast_pretest_pk(uint32_t c, uint32_t p0, uint32_t p8, uint32_t p4, uint32_t p12, uint32_t p2,
                                                    uint32_t p10, uint32_t p6, uint32_t p14, uint32_t t_hi) {
  const u16x2_t a0 = as_u16x2(p0), a8 = as_u16x2(p8), a4 = as_u16x2(p4), a12 = as_u16x2(p12);
  const u16x2_t a2 = as_u16x2(p2), a10 = as_u16x2(p10), a6 = as_u16x2(p6), a14 = as_u16x2(p14);
  const u16x2_t maxmin = __builtin_elementwise_max(
      __builtin_elementwise_max(__builtin_elementwise_min(a0, a8), __builtin_elementwise_min(a4, a12)),
      __builtin_elementwise_max(__builtin_elementwise_min(a2, a10), __builtin_elementwise_min(a6, a14)));
  const u16x2_t minmax = __builtin_elementwise_min(
      __builtin_elementwise_min(__builtin_elementwise_max(a0, a8), __builtin_elementwise_max(a4, a12)),
      __builtin_elementwise_min(__builtin_elementwise_max(a2, a10), __builtin_elementwise_max(a6, a14)));
  const u16x2_t T = as_u16x2(t_hi);
  const u16x2_t lo = __builtin_elementwise_sub_sat(as_u16x2(c & 0xff00ff00u), T);   // max(v - t, 0) << 8
  const u16x2_t hi = __builtin_elementwise_add_sat(as_u16x2(c | 0x00ff00ffu), T);   // min(v + t, 255) << 8 | 0xff  (0xffff when v + t > 255)
  return as_u32(__builtin_elementwise_sub_sat(lo, maxmin)) | as_u32(__builtin_elementwise_sub_sat(minmax, hi));
}

// The same test with every pixel in the LOW byte of its 16-bit lane (high byte zero): the even pixels of a dword are isolated with
// one v_and_b32 (a full-rate instruction on gfx950, profiles/valu_ceiling_r2.txt) instead of the half-rate shift that would move
// them into the high byte; v + t cannot overflow 16 bits, so only v - t saturates.
__device__ __forceinline__ uint32_t fast_pretest_pk_lo(uint32_t c, uint32_t p0, uint32_t p8, uint32_t p4, uint32_t p12, uint32_t p2,
                                                       uint32_t p10, uint32_t p6, uint32_t p14, uint32_t t_lo) {
  const u16x2_t a0 = as_u16x2(p0), a8 = as_u16x2(p8), a4 = as_u16x2(p4), a12 = as_u16x2(p12);
  const u16x2_t a2 = as_u16x2(p2), a10 = as_u16x2(p10), a6 = as_u16x2(p6), a14 = as_u16x2(p14);
  const u16x2_t maxmin = __builtin_elementwise_max(
      __builtin_elementwise_max(__builtin_elementwise_min(a0, a8), __builtin_elementwise_min(a4, a12)),
      __builtin_elementwise_max(__builtin_elementwise_min(a2, a10), __builtin_elementwise_min(a6, a14)));
  const u16x2_t minmax = __builtin_elementwise_min(
      __builtin_elementwise_min(__builtin_elementwise_max(a0, a8), __builtin_elementwise_max(a4, a12)),
      __builtin_elementwise_min(__builtin_elementwise_max(a2, a10), __builtin_elementwise_max(a6, a14)));
  const u16x2_t T = as_u16x2(t_lo);
  const u16x2_t lo = __builtin_elementwise_sub_sat(as_u16x2(c), T);   // max(v - t, 0)
  const u16x2_t hi = as_u16x2(c) + T;                                  // v + t <= 510
  return as_u32(__builtin_elementwise_sub_sat(lo, maxmin)) | as_u32(__builtin_elementwise_sub_sat(minmax, hi));
}

// one cell (logical item L of a launch over cells [cell_base, cell_base + ncells_sub) of every frame); T threads
// Two passes (flags bit 1; the batch kernel's default since round 6): the cell loop of the reference LITERALLY — cv::FAST at iniThFAST, and only
// where that leaves the cell without a keypoint cv::FAST again at minThFAST (src/ORBextractor.cc:826-850) — stages B, C, D run at iniTh first.
// Four cells in five hold an iniTh corner (tools/fast_pass_rates.py: 78 % on the synthetic stream, 80 % on natural crops) and iniTh's candidate
// list is a third (natural) to two thirds (synthetic) of minTh's, so the 130-instruction exact score, the NMS and the compaction run over fewer
// trips; the fifth cell pays stage B twice.  Measured (profiles/fast_passes_r6.txt): natural crops k_fast_cells 0.520 -> 0.385 ms per 256 frames,
// the step 232 k -> 259 k features/ms; 1024 x 1024: +2 %; the synthetic stream: the kernel alone 0.412 -> 0.383 ms but +2 % VALU instructions
// (a quarter of its cells are flat and pay stage B twice for nothing), i.e. -1 % on the two-lane step, inside the run-to-run spread.
// One pass (flags bit 1 clear: the fused single-frame launch, where the slowest cell sets the kernel's time; "fast_passes" = 1): stage B at minTh,
// the threshold chosen afterwards — the same keypoints by the closed form of SURVEY.md section 8(c)-F.
template <int T, int PITCH, bool PK>
__device__ __forceinline__ void fast_cell(uint8_t* smem, const int L, const DeviceGeom* __restrict__ g, const CellGeom* __restrict__ cells,
                                          const uint8_t* __restrict__ imgs, long long img_row_stride,
                                          long long img_frame_stride, const uint8_t* __restrict__ pyr,
                                          long long pyr_frame_bytes, uint32_t* __restrict__ cand,
                                          int32_t* __restrict__ cell_cnt, int ini_th, int min_th, int tile_rows,
                                          int cell_base, int ncells_sub, uint32_t m_ncells_sub, int flags,
                                          int list_cap, int nwords) {
  // LDS: [16 B pad][tile_rows][PITCH] raw pixels (+ alignment shift xo) | [tile_rows][PITCH] scores with a 1-px
  // zero frame | list | bitmap | word prefix.  PITCH is a compile-time constant so every circle / neighbour access is an
  // immediate offset.  Everything is sized by the launch for the cells it covers (tile_rows, list_cap, nwords = 64 or 256): the
  // LDS footprint of a workgroup decides how many of them a CU holds, and this kernel lives on residency.
  const int stage_dma = flags & 1;          // the tile by LDS-DMA loads
  const bool twopass = (flags & 2) != 0;    // block-uniform (kernel argument)
  uint8_t* tile = smem + 16;
  uint8_t* sc = tile + tile_rows * PITCH;
  uint16_t* list = (uint16_t*)(sc + tile_rows * PITCH);  // candidate pixels (bit 15: NMS survivor)
  uint32_t* bitmap = (uint32_t*)(list + list_cap);        // list_cap is a multiple of 8
  int* wpre = (int*)(bitmap + nwords);
  constexpr int NW = T / 64, WPT = 256 / T, P4 = PITCH / 4;
  __shared__ int wave_tot[NW];
  __shared__ int s_cnt;

  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  // this launch covers cells [cell_base, cell_base + ncells_sub) of every frame (level 0 runs as its own launch,
  // concurrently with the pyramid chain)
  const int frame = fast_div(L, m_ncells_sub), cell = cell_base + (L - frame * ncells_sub);
  const CellGeom cg = cells[cell];
  const uint8_t* img;
  int pitch;  // < 2^23 (checked on the host): row offsets are 24-bit multiplies
  if (cg.level == 0) { img = imgs + (long long)frame * img_frame_stride; pitch = (int)img_row_stride; }
  else { img = pyr + (long long)frame * pyr_frame_bytes + cg.plane_off; pitch = cg.pitch; }
  const int cw = cg.cw, ch = cg.ch, dw = cw - 6, dh = ch - 6;
  // ---- A: stage the sub-image with aligned dword loads when the source allows it
  const bool al = ((pitch & 3) == 0) && ((((unsigned long long)img) & 3) == 0);
  const int xo = al ? (cg.x0 & 3) : 0;
  if (PITCH == 64 && al && stage_dma) {   // a 64-byte tile pitch: one wave-instruction = four whole rows
    // "fast_stage_dma": the same dwords straight into the tile by LDS-DMA (global_load_lds: no staging VGPRs, no ds_write pass); lane <->
    // (row r0 + t / 16, dword t % 16), so the LDS image of one wave-instruction is 64 consecutive dwords = four tile rows.  The zeroing
    // below runs while the loads are in flight; the barrier that ends stage A waits for them (vmcnt).
    const uint8_t* src = img + (long long)cg.y0 * pitch + (cg.x0 - xo);
    const int ndw = (xo + cw + 3) >> 2;
    for (int r0 = 0; r0 < ch; r0 += T / 16) {
      const int r = r0 + (t >> 4), c = t & 15;
      uint8_t* dst = tile + (size_t)(r0 * 16 + (t & ~63)) * 4;   // wave-uniform; the hardware adds lane * 4
      if (r < ch && c < ndw)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (uint32_t)(__mul24(r, pitch) + 4 * c)),
                                         (__attribute__((address_space(3))) void*)dst, 4, 0, 0);
    }
  } else if (PITCH != 64 && al && stage_dma) {
    // the 96-byte pitch of wide cells: the tile as a FLAT dword stream — dword
    // d = T * trip + t lands at LDS dword d (lane-linear, as global_load_lds requires) and is fetched from row d / P4, column d % P4 of the
    // source (per-lane SOURCE addresses are free; the division is by a compile-time constant); pad columns (c >= ndw) are not loaded
    const uint8_t* src = img + (long long)cg.y0 * pitch + (cg.x0 - xo);
    const int ndw = (xo + cw + 3) >> 2, total = ch * P4;
    for (int d0 = 0; d0 < total; d0 += T) {
      const uint32_t d = (uint32_t)(d0 + t), r = d / (uint32_t)P4, c = d - r * (uint32_t)P4;
      uint8_t* dst = tile + (size_t)(d0 + (t & ~63)) * 4;   // wave-uniform; the hardware adds lane * 4
      if ((int)d < total && (int)c < ndw)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (uint32_t)(__mul24((int)r, pitch) + 4 * (int)c)),
                                         (__attribute__((address_space(3))) void*)dst, 4, 0, 0);
    }
  } else if (al) {
    const uint8_t* src = img + (long long)cg.y0 * pitch + (cg.x0 - xo);
    const int ndw = (xo + cw + 3) >> 2;
    for (int r = t >> 4; r < ch; r += T / 16)
      for (int c = t & 15; c < ndw; c += 16)
        ((uint32_t*)tile)[r * P4 + c] = *(const uint32_t*)(src + (uint32_t)(__mul24(r, pitch) + 4 * c));
  } else {
    const uint8_t* src = img + (long long)cg.y0 * pitch + cg.x0;
#pragma unroll 1  // cold path: keep it out of the register budget
    for (int r = wv; r < ch; r += NW) {
#pragma unroll 1
      for (int c = lane; c < cw; c += 64) tile[r * PITCH + c] = src[(long long)r * pitch + c];
    }
  }
  for (int i = t; i < ((dh + 2) * P4 + 3) >> 2; i += T) ((uint4*)sc)[i] = make_uint4(0u, 0u, 0u, 0u);  // sc is 16-byte aligned
  for (int i = t; i < nwords; i += T) bitmap[i] = 0;
  if (t == 0) s_cnt = 0;
  if (stage_dma) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tile's LDS-DMA loads (block-uniform)
  __syncthreads();
  int TH = twopass ? ini_th : min_th;   // threshold of the current pass
  int n1 = 0;
#pragma unroll 1
  for (int pass = 0;; pass++) {
  // ---- B: necessary test, 4 pixels (one aligned LDS dword of centres) per lane and step; branch-free
  {
    const int kmin = (3 + xo) >> 2, kmax = (cw - 4 + xo) >> 2, ng = kmax - kmin + 1;
    const int nit = dh * ng;
    const uint32_t magic = (65536u + (uint32_t)ng - 1u) / (uint32_t)ng;  // i / ng == (i * magic) >> 16 for i < 65536 / ng
    for (int i0 = 0; i0 < nit; i0 += T) {
      const int act = (i0 + t) < nit;
      const int i = act ? i0 + t : nit - 1;
      const int ry = (int)(mul_u24((uint32_t)i, magic) >> 16);   // i < 2^16, magic <= 2^16
      const int k = kmin + (i - ry * ng);
      const uint32_t* rowp = (const uint32_t*)tile + (ry + 3) * P4 + k;
      const uint32_t dC = rowp[0], dL = rowp[-1], dR = rowp[1], dU = rowp[-3 * P4], dD = rowp[3 * P4];
      const uint32_t Q4 = __builtin_amdgcn_alignbyte(dR, dC, 3);   // x+3 of pixel j in byte j
      const uint32_t Q12 = __builtin_amdgcn_alignbyte(dC, dL, 1);  // x-3 of pixel j in byte j
      constexpr uint32_t LO = 0x00ff00ffu;
      const uint32_t t_hi = (uint32_t)TH * 0x01000100u, t_lo = (uint32_t)TH * 0x00010001u;
      // the two diagonal pairs (2,10) and (6,14): rows +-2, columns +-2
      const uint32_t eC = rowp[2 * P4], eL = rowp[2 * P4 - 1], eR = rowp[2 * P4 + 1];
      const uint32_t fC = rowp[-2 * P4], fL = rowp[-2 * P4 - 1], fR = rowp[-2 * P4 + 1];
      const uint32_t Q2 = __builtin_amdgcn_alignbyte(eR, eC, 2), Q14 = __builtin_amdgcn_alignbyte(eC, eL, 2);
      const uint32_t Q6 = __builtin_amdgcn_alignbyte(fR, fC, 2), Q10 = __builtin_amdgcn_alignbyte(fC, fL, 2);
      const int c0 = 4 * k - xo - 3;                               // detection-domain x of pixel 0
      bool ps[4];
      if constexpr (PK) {
        // pixels 1 and 3 sit in the high bytes of the raw dwords' 16-bit lanes (the low bytes are don't-cares there); pixels 0
        // and 2 are the low bytes, isolated by a mask
        uint32_t rO = fast_pretest_pk(dC, dD, dU, Q4, Q12, Q2, Q10, Q6, Q14, t_hi);
        uint32_t rE = fast_pretest_pk_lo(dC & LO, dD & LO, dU & LO, Q4 & LO, Q12 & LO, Q2 & LO, Q10 & LO, Q6 & LO, Q14 & LO, t_lo);
        // validity (x inside the detection domain, lane active) on the same lanes: (unsigned)(c0 + j) < dw, c0 >= -3
        const uint32_t c0a = act ? (uint32_t)c0 : 0x4000u;
        const u16x2_t C0 = as_u16x2(__builtin_amdgcn_perm(c0a, c0a, 0x01000100u));
        const u16x2_t DW = as_u16x2((uint32_t)dw * 0x10001u);
        const u16x2_t vE = __builtin_elementwise_sub_sat(DW, C0 + as_u16x2(0x00020000u));   // x = c0, c0 + 2
        const u16x2_t vO = __builtin_elementwise_sub_sat(DW, C0 + as_u16x2(0x00030001u));   // x = c0 + 1, c0 + 3
        rE = as_u32(__builtin_elementwise_min(as_u16x2(rE), vE));
        rO = as_u32(__builtin_elementwise_min(as_u16x2(rO), vO));
        ps[0] = (rE & 0xffffu) != 0; ps[2] = rE > 0xffffu;
        ps[1] = (rO & 0xffffu) != 0; ps[3] = rO > 0xffffu;
      } else
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int v = (dC >> (8 * j)) & 0xff, p0 = (dD >> (8 * j)) & 0xff, p8 = (dU >> (8 * j)) & 0xff;
        const int p4 = (Q4 >> (8 * j)) & 0xff, p12 = (Q12 >> (8 * j)) & 0xff;
        const int p2 = (Q2 >> (8 * j)) & 0xff, p10 = (Q10 >> (8 * j)) & 0xff, p6 = (Q6 >> (8 * j)) & 0xff, p14 = (Q14 >> (8 * j)) & 0xff;
        const bool dark = max(max(min(p0, p8), min(p4, p12)), max(min(p2, p10), min(p6, p14))) < v - TH;
        const bool bright = min(min(max(p0, p8), max(p4, p12)), min(max(p2, p10), max(p6, p14))) > v + TH;
        const bool valid = (unsigned)(c0 + j) < (unsigned)dw;
        ps[j] = (dark | bright) & valid & (act != 0);
      }
      const unsigned long long b0 = __ballot(ps[0]), b1 = __ballot(ps[1]), b2 = __ballot(ps[2]), b3 = __ballot(ps[3]);
      const int n0 = __popcll(b0), n1_ = __popcll(b1), n2 = __popcll(b2), n3 = __popcll(b3);
      const int tot = n0 + n1_ + n2 + n3;
      if (tot) {  // wave-uniform
        int base = 0;
        if (lane == 0) base = atomicAdd(&s_cnt, tot);
        base = __builtin_amdgcn_readfirstlane(base);
        // slot of pixel j of this lane: base + (passes of pixels < j in the wave) + (passes of pixel j in lower lanes);
        // v_mbcnt accumulates onto a scalar start value, so each slot costs two VALU instructions
        const int ent = c0 + (ry << 7);  // c0 may be negative for the first group; pixel j is only listed when c0 + j >= 0
        const int o0 = __builtin_amdgcn_mbcnt_hi((uint32_t)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b0, base));
        const int o1 = __builtin_amdgcn_mbcnt_hi((uint32_t)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b1, base + n0));
        const int o2 = __builtin_amdgcn_mbcnt_hi((uint32_t)(b2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b2, base + n0 + n1_));
        const int o3 = __builtin_amdgcn_mbcnt_hi((uint32_t)(b3 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b3, base + n0 + n1_ + n2));
        if (ps[0]) list[o0] = (uint16_t)ent;
        if (ps[1]) list[o1] = (uint16_t)(ent + 1);
        if (ps[2]) list[o2] = (uint16_t)(ent + 2);
        if (ps[3]) list[o3] = (uint16_t)(ent + 3);
      }
    }
  }
  __syncthreads();
  n1 = s_cnt;
  // ---- C: exact score of the listed pixels
  for (int e = t; e < n1; e += T) {
    const int ent = list[e];
    const int x = ent & 127, y = ent >> 7;
    const uint8_t* c0 = tile + y * PITCH + (x + xo);  // top-left of the 7x7 window; centre = c0[3 * PITCH + 3]
    const int v = c0[3 * PITCH + 3];
    uint32_t p[16];
    p[0] = c0[6 * PITCH + 3];  p[1] = c0[6 * PITCH + 4];  p[2] = c0[5 * PITCH + 5];  p[3] = c0[4 * PITCH + 6];
    p[4] = c0[3 * PITCH + 6];  p[5] = c0[2 * PITCH + 6];  p[6] = c0[1 * PITCH + 5];  p[7] = c0[4];
    p[8] = c0[3];              p[9] = c0[2];              p[10] = c0[1 * PITCH + 1]; p[11] = c0[2 * PITCH];
    p[12] = c0[3 * PITCH];     p[13] = c0[4 * PITCH];     p[14] = c0[5 * PITCH + 1]; p[15] = c0[6 * PITCH + 2];
    const int s = fast_score16(v, p);
    if (s >= TH && s > 0) sc[(y + 1) * PITCH + (x + 1)] = (uint8_t)s;
  }
  __syncthreads();
  // ---- D: 3x3 strict NMS inside the cell (neighbours outside the detection domain are 0)
  int any_ini = 0;
  for (int e = t; e < n1; e += T) {
    const int ent = list[e];
    const int x = ent & 127, y = ent >> 7;
    const uint8_t* q = sc + y * PITCH + x;  // top-left of the 3x3 window
    const int s = q[PITCH + 1];
    if (s > 0) {
      const int m = max(max(max((int)q[0], (int)q[1]), max((int)q[2], (int)q[PITCH])),
                        max(max((int)q[PITCH + 2], (int)q[2 * PITCH]), max((int)q[2 * PITCH + 1], (int)q[2 * PITCH + 2])));
      if (s > m) {
        list[e] = (uint16_t)(ent | 0x8000);
        any_ini |= twopass ? 1 : (int)(s >= ini_th);
      }
    }
  }
  const int use_ini = __syncthreads_or(any_ini);   // (also the barrier after which s_cnt may be reset and the list rewritten)
  if (!twopass) { TH = use_ini ? ini_th : min_th; break; }
  // two passes: every stored score is >= TH, so any NMS survivor is a keypoint of this pass
  if (use_ini || pass == 1 || ini_th == min_th) break;   // block-uniform
  TH = min_th;
  if (t == 0) s_cnt = 0;
  __syncthreads();
  }
  // ---- E: bitmap of the selected survivors (bit index = row-major pixel index)
  for (int e = t; e < n1; e += T) {
    const int le = list[e];
    if (le & 0x8000) {
      const int x = le & 127, y = (le >> 7) & 127;
      const int i = y * dw + x;
      if (sc[(y + 1) * PITCH + (x + 1)] >= TH) {
        atomicOr(&bitmap[i >> 5], 1u << (i & 31));
      }
    }
  }
  __syncthreads();
  // ---- F: exclusive popcount prefix over the bitmap words
  const int npx = dw * dh;
  if (npx <= 2048) {  // block-uniform; the usual case: <= 64 words, one wave scans them, the others go straight on
    if (wv == 0) {
      const int c = __popc(bitmap[lane]);  // words beyond the cell are zero
      int inc = c;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o); if (lane >= o) inc += u; }
      wpre[lane] = inc - c;
      if (lane == 63) cell_cnt[(long long)frame * g->ncells_total + cell] = inc;
    }
  } else {  // thread t <-> words t*WPT ..
    int cw_[WPT], c = 0;
#pragma unroll
    for (int k = 0; k < WPT; k++) { cw_[k] = __popc(bitmap[t * WPT + k]); c += cw_[k]; }
    int inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o); if (lane >= o) inc += u; }
    if (lane == 63) wave_tot[wv] = inc;
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < NW; k++) { const int wt = wave_tot[k]; off += k < wv ? wt : 0; tot += wt; }
    int run = off + inc - c;
#pragma unroll
    for (int k = 0; k < WPT; k++) { wpre[t * WPT + k] = run; run += cw_[k]; }
    if (t == 0) cell_cnt[(long long)frame * g->ncells_total + cell] = tot;
  }
  __syncthreads();
  // ---- G: row-major rank of every selected survivor -> its slot
  uint32_t* slot = cand + (long long)frame * g->cand_total + cg.slot_off;
  for (int e = t; e < n1; e += T) {
    const int le = list[e];
    if (le & 0x8000) {
      const int x = le & 127, y = (le >> 7) & 127;
      const int s = sc[(y + 1) * PITCH + (x + 1)];
      if (s >= TH) {
        const int i = y * dw + x;
        const int rank = wpre[i >> 5] + __popc(bitmap[i >> 5] & ((1u << (i & 31)) - 1u));
        slot[rank] = pack_pt(x + 3 + cg.relx, y + 3 + cg.rely, s);
      }
    }
  }
}

template <int T, int PITCH, bool PK = false>
__global__ __launch_bounds__(T) void k_fast_cells(const DeviceGeom* __restrict__ g, const CellGeom* __restrict__ cells,
                                                  const uint8_t* __restrict__ imgs, long long img_row_stride,
                                                  long long img_frame_stride, const uint8_t* __restrict__ pyr,
                                                  long long pyr_frame_bytes, uint32_t* __restrict__ cand,
                                                  int32_t* __restrict__ cell_cnt, int ini_th, int min_th, int tile_rows,
                                                  int nitems, int cell_base, int ncells_sub, uint32_t m_ncells_sub, int flags,
                                                  int list_cap, int nwords) {
  extern __shared__ __align__(16) uint8_t smem[];
  const int L = xcd_logical_block(nitems);
  if (L < 0) return;  // block-uniform
  fast_cell<T, PITCH, PK>(smem, L, g, cells, imgs, img_row_stride, img_frame_stride, pyr, pyr_frame_bytes, cand, cell_cnt, ini_th, min_th, tile_rows,
                          cell_base, ncells_sub, m_ncells_sub, flags, list_cap, nwords);
}

// ------------------------------------------------------------------------------------------------
// K3: quadtree distribution (src/ORBextractor.cc:555-779), see tests/support/quadtree_model.cpp for the
// sequential statement of the same array formulation.
// ------------------------------------------------------------------------------------------------
struct QNode { int16_t x0, y0, x1, y1; int32_t start, count; };

#ifdef ORBX_QT_PROFILE  // temporary phase timing of frame 0 (cycles): [level][slot]
__device__ long long g_qt_prof[kMaxLevels * 8];
#define QT_T0() long long _qt_t = wall_clock64()
#define QT_ACC(slot) do { if (threadIdx.x == 0 && blockIdx.y == 0) { const long long _n = wall_clock64(); g_qt_prof[blockIdx.x * 8 + (slot)] += _n - _qt_t; _qt_t = _n; } } while (0)
#else
#define QT_T0() do {} while (0)
#define QT_ACC(slot) do {} while (0)
#endif

constexpr unsigned long long kM21 = (1ull << 21) - 1ull;

__device__ __forceinline__ unsigned long long block_excl_scan(unsigned long long* a, int n, unsigned long long* wt) {
  const int T = blockDim.x, t = threadIdx.x, lane = t & 63, w = t >> 6, NW = T >> 6;
  const int ipt = (n + T - 1) / T;
  const int b = min(t * ipt, n), e = min(b + ipt, n);
  unsigned long long local = 0;
  for (int i = b; i < e; i++) local += a[i];
  unsigned long long inc = local;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned long long v = __shfl_up(inc, o);
    if (lane >= o) inc += v;
  }
  if (lane == 63) wt[w] = inc;
  __syncthreads();
  unsigned long long off = 0, total = 0;
  for (int k = 0; k < NW; k++) { const unsigned long long v = wt[k]; off += k < w ? v : 0; total += v; }
  unsigned long long run = off + inc - local;
  for (int i = b; i < e; i++) { const unsigned long long v = a[i]; a[i] = run; run += v; }
  __syncthreads();
  return total;
}

// One value per thread: exclusive prefix over the workgroup and the total, ONE barrier.  wt2 holds two sets of wave totals and
// `par` selects one; consecutive calls alternate, so a wave still reading the previous call's totals is never overwritten.
__device__ __forceinline__ unsigned long long block_scan1(unsigned long long v, unsigned long long* wt2, int par, unsigned long long& total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, NW = blockDim.x >> 6;
  unsigned long long inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned long long u = __shfl_up(inc, o);
    if (lane >= o) inc += u;
  }
  unsigned long long* wt = wt2 + (par << 3);
  if (lane == 63) wt[w] = inc;
  __syncthreads();
  unsigned long long off = 0, tot = 0;
  for (int k = 0; k < NW; k++) { const unsigned long long x = wt[k]; off += k < w ? x : 0; tot += x; }
  total = tot;
  return off + inc - v;
}

// LDS written by some lanes of a wave and read by other lanes of the same wave: order the accesses.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One wave: the closed form of libstdc++'s Hoare partition step on v[first, last) with the pivot already moved to
// v[first] (gnu_sort.h: partition_closed_form is the sequential statement).  ia / ir: per-wave LDS index scratch.
__device__ __forceinline__ int wave_partition(unsigned long long* v, int first, int last, uint16_t* ia, uint16_t* ir) {
  const int lane = threadIdx.x & 63;
  const unsigned long long lt = (1ull << lane) - 1ull;
  const uint32_t kp = (uint32_t)(v[first] >> 32);
  int nL = 0, nR = 0;
  for (int base = first + 1; base < last; base += 64) {
    const int i = base + lane;
    const bool valid = i < last;
    const uint32_t k = valid ? (uint32_t)(v[i] >> 32) : 0u;
    const bool isL = valid && k >= kp, isR = valid && k <= kp;
    const unsigned long long bL = __ballot(isL), bR = __ballot(isR);
    if (isL) ia[nL + __popcll(bL & lt)] = (uint16_t)i;
    if (isR) ir[nR + __popcll(bR & lt)] = (uint16_t)i;
    nL += __popcll(bL);
    nR += __popcll(bR);
  }
  wave_lds_sync();
  const int lim = min(nL, nR + 1);
  int cnt = 0;
  for (int j0 = 1; j0 <= lim; j0 += 64) {
    const int j = j0 + lane;
    bool ok = false;
    if (j <= lim) {
      const int bj = j <= nR ? (int)ir[nR - j] : first;
      ok = (int)ia[j - 1] < bj;
    }
    cnt += __popcll(__ballot(ok));
  }
  for (int j = 1 + lane; j <= cnt; j += 64) {
    const int a = ia[j - 1], b = ir[nR - j];
    const unsigned long long ta = v[a], tb = v[b];
    v[a] = tb;
    v[b] = ta;
  }
  const int J = cnt + 1;
  const int rprev = J == 1 ? last : (int)ir[nR - (J - 1)];
  const int cut = (J <= nL && (int)ia[J - 1] < rprev) ? (int)ia[J - 1] : rprev;
  wave_lds_sync();
  return cut;
}

// One wave, one introsort step (median-of-3 + Hoare partition, returns the cut) of a segment with at most 256 interior
// elements held in four registers per lane (register r, lane e <-> position first + 1 + 64 r + e): the ranks of the closed form
// come from per-register ballots and their scalar popcounts, the swap partners meet through the ia / ir scratch.
__device__ __forceinline__ int wave_step_regs(unsigned long long* v, int first, int last, uint16_t* ia, uint16_t* ir) {
  const int lane = threadIdx.x & 63;
  const unsigned long long ltm = (1ull << lane) - 1ull, gtm = ~(ltm | (1ull << lane));
  const int f = __builtin_amdgcn_readfirstlane(first), l = __builtin_amdgcn_readfirstlane(last);
  unsigned long long mine[4];
#pragma unroll
  for (int r = 0; r < 4; r++) { const int i = f + 1 + 64 * r + lane; mine[r] = i < l ? v[i] : 0ull; }
  const int pa = f + 1, pb = f + ((l - f) >> 1), pc = l - 1;
  const unsigned long long vf = v[f], va = v[pa], vb = v[pb], vc = v[pc];
  const uint32_t ka = (uint32_t)(va >> 32), kb = (uint32_t)(vb >> 32), kc = (uint32_t)(vc >> 32);
  int pm;
  if (ka < kb) pm = kb < kc ? pb : (ka < kc ? pc : pa);
  else pm = ka < kc ? pa : (kb < kc ? pc : pb);
  const unsigned long long piv = pm == pa ? va : (pm == pb ? vb : vc);
  const uint32_t kp = (uint32_t)(piv >> 32);
  unsigned long long bL[4], bR[4];
  bool isL[4], isR[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int i = f + 1 + 64 * r + lane;
    if (i == pm) mine[r] = vf;
    const uint32_t k = (uint32_t)(mine[r] >> 32);
    isL[r] = i < l && k >= kp;
    isR[r] = i < l && k <= kp;
    bL[r] = __ballot(isL[r]);
    bR[r] = __ballot(isR[r]);
  }
  int preL[4], sufR[4];
  preL[0] = 0; preL[1] = __popcll(bL[0]); preL[2] = preL[1] + __popcll(bL[1]); preL[3] = preL[2] + __popcll(bL[2]);
  sufR[3] = 0; sufR[2] = __popcll(bR[3]); sufR[1] = sufR[2] + __popcll(bR[2]); sufR[0] = sufR[1] + __popcll(bR[1]);
  int rankL[4], rankR[4];
  bool partL[4], partR[4];
  unsigned long long pL[4], pR[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int i = f + 1 + 64 * r + lane;
    rankL[r] = preL[r] + __popcll(bL[r] & ltm);   // a's before me
    rankR[r] = sufR[r] + __popcll(bR[r] & gtm);   // b's after me
    partL[r] = isL[r] && rankR[r] >= rankL[r] + 1;
    partR[r] = isR[r] && rankL[r] >= rankR[r] + 1;
    pL[r] = __ballot(partL[r]);
    pR[r] = __ballot(partR[r]);
    if (partL[r]) ia[rankL[r]] = (uint16_t)i;
    if (partR[r]) ir[rankR[r]] = (uint16_t)i;
    if (i < l) v[i] = mine[r];   // the arrangement after the median swap: what the partners read
  }
  if (lane == 0) v[f] = piv;
  wave_lds_sync();
  unsigned long long other[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    other[r] = 0ull;
    if (partL[r]) other[r] = v[ir[rankL[r]]];
    if (partR[r]) other[r] = v[ia[rankR[r]]];
  }
  wave_lds_sync();
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int i = f + 1 + 64 * r + lane;
    if (partL[r] || partR[r]) v[i] = other[r];
  }
  int aJ = 0x7fffffff, rprev = l;
#pragma unroll
  for (int r = 3; r >= 0; r--) {
    const unsigned long long rest = bL[r] & ~pL[r];
    if (rest) aJ = f + 64 * r + __ffsll((long long)rest);
    if (pR[r]) rprev = f + 64 * r + __ffsll((long long)pR[r]);
  }
  wave_lds_sync();
  return aJ < rprev ? aJ : rprev;
}

// One wave runs the whole introsort loop of a segment whose interior (first, last) fits the wave (last - first - 1 <= 64):
// lane e <-> position first + 1 + e, the elements live in registers.  Median-of-3 by three lane reads; the Hoare step from
// two ballots — with a[j] the j-th position (from the left) whose key is >= the pivot's and b[j] the j-th (from the right) whose
// key is <= it, the pair (a[j], b[j]) is swapped iff a[j] < b[j], which for the lane at a[j] reads "at least j such b after me"
// and for the lane at b[j] "at least j such a before me" (an element equal to the pivot is both, but cannot take part as both);
// the cut is the first a that did not swap, or the last b that did, whichever comes first (gnu_sort.h, partition_closed_form).
// Sub-segments longer than 16 are taken up by the same wave (at most two can be pending inside 65 positions); the others get
// their final-segment marks.  ia / ir: per-segment LDS scratch (last - first entries each), seg: as in block_gnu_sort.
__device__ __forceinline__ void wave_introsort_small(unsigned long long* v, int first, int last, int depth, uint32_t* seg,
                                                     uint16_t* ia0, uint16_t* ir0, int seg0) {
  const int lane = threadIdx.x & 63;
  const unsigned long long ltm = (1ull << lane) - 1ull, gtm = ~(ltm | (1ull << lane));
  int f = __builtin_amdgcn_readfirstlane(first), l = __builtin_amdgcn_readfirstlane(last), d = __builtin_amdgcn_readfirstlane(depth);
  unsigned long long st0 = 0, st1 = 0;  // pending (f | l << 16 | d << 32)
  int sp = 0;
  while (true) {
    if (d == 0) {  // depth limit: heapsort fallback, result in final order
      if (lane == 0) orbx_sort::heap_sort(v, f, l);
      for (int i = f + lane; i < l; i += 64) seg[i] = (uint32_t)i | (uint32_t)(i + 1) << 16;
      wave_lds_sync();
    } else {
      d--;
      uint16_t* ia = ia0 + (f - seg0);
      uint16_t* ir = ir0 + (f - seg0);
      const int i = f + 1 + lane;
      const bool valid = i < l;
      unsigned long long mine = valid ? v[i] : 0ull;
      const unsigned long long vf = v[f];
      // std::__move_median_to_first(first, first + 1, mid, last - 1)
      const int lb = ((l - f) >> 1) - 1, lc = l - f - 2;
      const uint32_t mk = (uint32_t)(mine >> 32), ml = (uint32_t)mine;
      const uint32_t ka = __builtin_amdgcn_readlane(mk, 0), kb = __builtin_amdgcn_readlane(mk, lb), kc = __builtin_amdgcn_readlane(mk, lc);
      int lm;
      if (ka < kb) lm = kb < kc ? lb : (ka < kc ? lc : 0);
      else lm = ka < kc ? 0 : (kb < kc ? lc : lb);
      const uint32_t kp = __builtin_amdgcn_readlane(mk, lm);
      const unsigned long long piv = (unsigned long long)kp << 32 | __builtin_amdgcn_readlane(ml, lm);
      if (lane == lm) mine = vf;
      // std::__unguarded_partition(first + 1, last, first), closed form
      const uint32_t k = (uint32_t)(mine >> 32);
      const bool isL = valid && k >= kp, isR = valid && k <= kp;
      const unsigned long long bL = __ballot(isL), bR = __ballot(isR);
      const int nLbefore = __popcll(bL & ltm), nRafter = __popcll(bR & gtm);
      const bool partL = isL && nRafter >= nLbefore + 1;
      const bool partR = isR && nLbefore >= nRafter + 1;
      const unsigned long long pL = __ballot(partL), pR = __ballot(partR);
      if (partL) ia[nLbefore] = (uint16_t)lane;
      if (partR) ir[nRafter] = (uint16_t)lane;
      wave_lds_sync();
      int partner = lane;
      if (partL) partner = ir[nLbefore];
      if (partR) partner = ia[nRafter];
      const unsigned long long other = __shfl(mine, partner);
      if (partL || partR) mine = other;
      const unsigned long long rest = bL & ~pL;
      const int aJ = rest ? f + __ffsll((long long)rest) : 0x7fffffff;
      const int rprev = pR ? f + __ffsll((long long)pR) : l;
      const int cut = aJ < rprev ? aJ : rprev;
      if (valid) v[i] = mine;
      if (lane == 0) v[f] = piv;
      // children [f, cut) and [cut, l)
      const bool bigL = cut - f > 16, bigR = l - cut > 16;
      if (!bigL) for (int q = f + lane; q < cut; q += 64) seg[q] = (uint32_t)f | (uint32_t)cut << 16;
      if (!bigR) for (int q = cut + lane; q < l; q += 64) seg[q] = (uint32_t)cut | (uint32_t)l << 16;
      wave_lds_sync();
      if (bigL && bigR) {
        const unsigned long long e = (unsigned long long)cut | (unsigned long long)l << 16 | (unsigned long long)d << 32;
        if (sp == 0) st0 = e; else st1 = e;
        sp++;
        l = cut;
        continue;
      }
      if (bigL) { l = cut; continue; }
      if (bigR) { f = cut; continue; }
    }
    if (sp == 0) break;
    sp--;
    const unsigned long long e = sp == 0 ? st0 : st1;
    f = (int)(e & 0xffffull); l = (int)((e >> 16) & 0xffffull); d = (int)(e >> 32);
  }
}

// position of the k-th set bit of m, counted from bit 0 (k = 0: the lowest); the bit must exist
__device__ __forceinline__ int select_bit64(unsigned long long m, int k) {
  uint32_t w = (uint32_t)m;
  int base = 0;
  const int c = __popc(w);
  if (k >= c) { k -= c; w = (uint32_t)(m >> 32); base = 32; }
#pragma unroll
  for (int sft = 16; sft >= 1; sft >>= 1) {
    const uint32_t lowm = w & ((1u << sft) - 1u);
    const int cl = __popc(lowm);
    if (k >= cl) { k -= cl; w >>= sft; base += sft; } else w = lowm;
  }
  return base;
}

// One wave, a block of at most 64 consecutive positions [first, last): the whole introsort loop of that block with the elements in
// registers (lane e <-> position first + e) and EVERY pending segment of the block advancing in the same step — the steps of a block are
// its recursion depth (3-4), not its number of segments (7-15), and a step touches no LDS: the median candidates, the pivot and the swap
// partners travel by lane permutes, and a lane finds its partner arithmetically (the closed form of partition_closed_form pairs the j-th
// key >= pivot from the left with the j-th key <= pivot from the right: "the j-th set bit" of a ballot restricted to the lane's segment).
// Same permutation as wave_introsort_small (gnu_sort.h is the sequential statement).  seg: as in block_gnu_sort.
__device__ __forceinline__ void wave_introsort_sync(unsigned long long* v, int first, int last, int depth, uint32_t* seg) {
  const int lane = threadIdx.x & 63;
  const int base = __builtin_amdgcn_readfirstlane(first), n = __builtin_amdgcn_readfirstlane(last) - base;
  const bool in = lane < n;
  unsigned long long mine = in ? v[base + lane] : 0ull;
  int sf = 0, sl = n, sd = __builtin_amdgcn_readfirstlane(depth);   // this lane's segment [sf, sl) in lane indices, its remaining depth
  bool act = in && n > 16;
  const unsigned long long ltm = (1ull << lane) - 1ull, gtm = ~(ltm | (1ull << lane));
  while (true) {
    if (__ballot(act) == 0ull) break;
    unsigned long long heap = __ballot(act && sd == 0);
    if (heap) {   // depth limit (practically never): the sequential heapsort on those segments, through LDS
      if (in) v[base + lane] = mine;
      wave_lds_sync();
      while (heap) {
        const int p = __ffsll((long long)heap) - 1;
        const int hf = __builtin_amdgcn_readlane(sf, p), hl = __builtin_amdgcn_readlane(sl, p);
        if (lane == 0) orbx_sort::heap_sort(v, base + hf, base + hl);
        const unsigned long long hm = (hl >= 64 ? ~0ull : ((1ull << hl) - 1ull)) & ~((1ull << hf) - 1ull);
        heap &= ~hm;
        if (lane >= hf && lane < hl) { sf = lane; sl = lane + 1; act = false; }   // in final order: every element its own segment
      }
      wave_lds_sync();
      if (in) mine = v[base + lane];
      continue;
    }
    if (act) sd--;
    uint32_t key = (uint32_t)(mine >> 32);
    // std::__move_median_to_first(first, first + 1, mid, last - 1), then the pivot sits at `first`
    const int pa = sf + 1, pb = sf + ((sl - sf) >> 1), pc = sl - 1;
    const uint32_t ka = __shfl(key, pa), kb = __shfl(key, pb), kc = __shfl(key, pc);
    int pm;
    if (ka < kb) pm = kb < kc ? pb : (ka < kc ? pc : pa);
    else pm = ka < kc ? pa : (kb < kc ? pc : pb);
    const unsigned long long piv = __shfl(mine, pm), vf = __shfl(mine, sf);
    const uint32_t kp = (uint32_t)(piv >> 32);
    if (act) { if (lane == sf) mine = piv; else if (lane == pm) mine = vf; }
    key = (uint32_t)(mine >> 32);
    // std::__unguarded_partition(first + 1, last, first), closed form, inside the lane's own segment
    const bool inr = act && lane > sf;
    const bool isL = inr && key >= kp, isR = inr && key <= kp;
    const unsigned long long segm = (sl >= 64 ? ~0ull : ((1ull << sl) - 1ull)) & ~((2ull << sf) - 1ull);   // bits (sf, sl)
    const unsigned long long mL = __ballot(isL) & segm, mR = __ballot(isR) & segm;
    const int rankL = __popcll(mL & ltm), rankR = __popcll(mR & gtm);
    const bool partL = isL && rankR >= rankL + 1, partR = isR && rankL >= rankR + 1;
    const unsigned long long pL = __ballot(partL), pR = __ballot(partR) & segm;
    // partner: the key <= pivot with exactly rankL such keys after it / the key >= pivot with exactly rankR such keys before it
    const int partner = partL ? select_bit64(mR, __popcll(mR) - 1 - rankL) : partR ? select_bit64(mL, rankR) : lane;
    const unsigned long long other = __shfl(mine, partner);
    if (partL || partR) mine = other;
    // the cut: the first key >= pivot that did not swap, or the lowest position that received one, whichever comes first
    const unsigned long long rest = mL & ~pL;
    const int aJ = rest ? __ffsll((long long)rest) - 1 : 0x7fffffff;
    const int rprev = pR ? __ffsll((long long)pR) - 1 : sl;
    const int cut = aJ < rprev ? aJ : rprev;
    if (act) {
      if (lane < cut) sl = cut; else sf = cut;
      act = sl - sf > 16;
    }
  }
  if (in) { v[base + lane] = mine; seg[base + lane] = (uint32_t)(base + sf) | (uint32_t)(base + sl) << 16; }
  wave_lds_sync();
}

// std::sort(v, v + n) with the reference's (count, UL.x) comparator, exact libstdc++ permutation (ties included),
// by the whole workgroup: level-synchronous introsort loop (one wave per pending segment and round), then a stable
// rank inside every final segment (== __final_insertion_sort).  tmp: n elements; seg: n words; q0/q1: n/16+2 entries each;
// idx: 2 * n uint16 (n < 65536).  In the workgroup's node scratch (LDS, or HBM for huge levels).  Ends with a barrier.
__device__ __forceinline__ void block_gnu_sort(unsigned long long* v, int n, unsigned long long* tmp, uint32_t* seg,
                                               unsigned long long* q0, unsigned long long* q1, uint16_t* idx, int* sh_cnt) {
  const int T = blockDim.x, t = threadIdx.x, lane = t & 63, w = t >> 6, NW = T >> 6;
  QT_T0();
  for (int i = t; i < n; i += T) seg[i] = (uint32_t)n << 16;  // lo = 0, hi = n
  // (Round 5 measured the alternative for vectors of <= 257 elements — wave 0 alone, segment after segment, no level-synchronous rounds: the
  // quadtree launch of a single frame 36.5 -> 38.5 us, a 256-frame batch 103.7 -> 105.0 us.  The rounds' barriers cost less than taking the
  // sibling segments one after the other.  HISTORY.md.)
  if (n > 16) {
    if (t == 0) { q0[0] = (unsigned long long)n << 16 | (unsigned long long)(2 * (31 - __clz(n))) << 32; *sh_cnt = 0; }  // first | last << 16 | depth << 32
    __syncthreads();
    int ncur = 1;
    uint16_t* ia = idx;      // segments of one round are disjoint: every wave indexes the shared scratch by position
    uint16_t* ir = idx + n;
    while (ncur > 0) {
      for (int sidx = w; sidx < ncur; sidx += NW) {
        const unsigned long long pk = q0[sidx];
        const int f = (int)(pk & 0xffffull), l = (int)((pk >> 16) & 0xffffull), d = (int)(pk >> 32);
        if (l - f <= 64) {  // a block of <= 64 positions: all its segments step together, in registers, without LDS
          wave_introsort_sync(v, f, l, d, seg);
          continue;
        }
        if (l - f <= 65) {  // fits a wave's registers: this wave finishes the segment and everything below it
          wave_introsort_small(v, f, l, d, seg, ia, ir, 0);
          continue;
        }
        if (d == 0) {  // depth limit: heapsort fallback (practically never), result already in final order
          if (lane == 0) orbx_sort::heap_sort(v, f, l);
          for (int i = f + lane; i < l; i += 64) seg[i] = (uint32_t)i | (uint32_t)(i + 1) << 16;
          continue;
        }
        int cut;
        if (l - f <= 257) {
          cut = wave_step_regs(v, f, l, ia + f, ir + f);
        } else {
          if (lane == 0) orbx_sort::move_median_to_first(v, f, f + 1, f + (l - f) / 2, l - 1);
          wave_lds_sync();
          cut = wave_partition(v, f, l, ia + f, ir + f);
        }
#pragma unroll
        for (int c = 0; c < 2; c++) {
          const int cf = c ? cut : f, cl = c ? l : cut;
          if (cl - cf > 16) {
            if (lane == 0) q1[atomicAdd(sh_cnt, 1)] = (unsigned long long)cf | (unsigned long long)cl << 16 | (unsigned long long)(d - 1) << 32;
          } else {
            for (int i = cf + lane; i < cl; i += 64) seg[i] = (uint32_t)cf | (uint32_t)cl << 16;
          }
        }
      }
      __syncthreads();
      ncur = *sh_cnt;
      __syncthreads();
      if (t == 0) *sh_cnt = 0;
      { unsigned long long* tq = q0; q0 = q1; q1 = tq; }
      __syncthreads();
    }
  }
  __syncthreads();
  QT_ACC(5);
  for (int i = t; i < n; i += T) {
    const unsigned long long e = v[i];
    const uint32_t k = (uint32_t)(e >> 32), sg = seg[i];
    const int lo = (int)(sg & 0xffffu), hi = (int)(sg >> 16);
    // a final segment holds at most 16 elements: sixteen independent reads (clamped past its end) instead of a dependent loop
    uint32_t kj[16];
#pragma unroll
    for (int u = 0; u < 16; u++) kj[u] = (uint32_t)(v[min(lo + u, hi - 1)] >> 32);
    int rank = lo;
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const int j = lo + u;
      rank += (j < hi) && ((kj[u] < k) || (kj[u] == k && j < i));
    }
    tmp[rank] = e;
  }
  __syncthreads();
  for (int i = t; i < n; i += T) v[i] = tmp[i];
  if (t == 0) *sh_cnt = 0;   // callers count on it
  __syncthreads();
  QT_ACC(6);
#ifdef ORBX_QT_PROFILE
  if (threadIdx.x == 0 && blockIdx.y == 0) g_qt_prof[blockIdx.x * 8 + 7] += 100;   // number of sorts (x 1 us in the tool's print)
#endif
}

// One wave: child counts of `nd` (DivideNode, src/ORBextractor.cc:480-536) and optional stable scatter cur->nxt.
__device__ __forceinline__ int4 wave_split(const QNode nd, const uint32_t* cur, uint32_t* nxt, bool scatter) {
  const int lane = threadIdx.x & 63;
  const int sx = nd.x0 + ((nd.x1 - nd.x0 + 1) >> 1);  // ceil(float(w)/2)
  const int sy = nd.y0 + ((nd.y1 - nd.y0 + 1) >> 1);
  int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  for (int o = 0; o < nd.count; o += 64) {
    const int e = o + lane;
    int chd = 4;
    if (e < nd.count) {
      const uint32_t p = cur[nd.start + e];
      chd = (pt_x(p) < sx ? 0 : 1) + (pt_y(p) < sy ? 0 : 2);
    }
    c0 += __popcll(__ballot(chd == 0));
    c1 += __popcll(__ballot(chd == 1));
    c2 += __popcll(__ballot(chd == 2));
    c3 += __popcll(__ballot(chd == 3));
  }
  if (scatter) {
    int r0 = nd.start, r1 = r0 + c0, r2 = r1 + c1, r3 = r2 + c2;
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int o = 0; o < nd.count; o += 64) {
      const int e = o + lane;
      int chd = 4;
      uint32_t p = 0;
      if (e < nd.count) {
        p = cur[nd.start + e];
        chd = (pt_x(p) < sx ? 0 : 1) + (pt_y(p) < sy ? 0 : 2);
      }
      const unsigned long long b0 = __ballot(chd == 0), b1 = __ballot(chd == 1), b2 = __ballot(chd == 2),
                               b3 = __ballot(chd == 3);
      if (chd < 4) {
        const unsigned long long mine = chd == 0 ? b0 : chd == 1 ? b1 : chd == 2 ? b2 : b3;
        const int basep = chd == 0 ? r0 : chd == 1 ? r1 : chd == 2 ? r2 : r3;
        nxt[basep + __popcll(mine & lt)] = p;
      }
      r0 += __popcll(b0); r1 += __popcll(b1); r2 += __popcll(b2); r3 += __popcll(b3);
    }
  }
  return make_int4(c0, c1, c2, c3);
}

__device__ __forceinline__ QNode child_node(const QNode nd, int c, int start, int count) {
  const int sx = nd.x0 + ((nd.x1 - nd.x0 + 1) >> 1);
  const int sy = nd.y0 + ((nd.y1 - nd.y0 + 1) >> 1);
  QNode r;
  r.x0 = (c & 1) ? sx : nd.x0; r.x1 = (c & 1) ? nd.x1 : sx;
  r.y0 = (c & 2) ? sy : nd.y0; r.y1 = (c & 2) ? nd.y1 : sy;
  r.start = start; r.count = count;
  return r;
}

// One thread: the same child counts / stable scatter for a small node (count <= kQtThreadNode).  In the sorted phase the
// list holds many nodes of a few points each; a wave per node would spend a whole ballot round trip on every one of them.
constexpr int kQtThreadNode = 16;
__device__ __forceinline__ int4 thread_split_count(const QNode nd, const uint32_t* cur) {
  const int sx = nd.x0 + ((nd.x1 - nd.x0 + 1) >> 1);
  const int sy = nd.y0 + ((nd.y1 - nd.y0 + 1) >> 1);
  uint32_t acc = 0;  // four 8-bit counters
  uint32_t p[kQtThreadNode];  // every read issued before the first use: one LDS round trip, not one per point
#pragma unroll
  for (int e = 0; e < kQtThreadNode; e++) p[e] = cur[nd.start + min(e, nd.count - 1)];
#pragma unroll
  for (int e = 0; e < kQtThreadNode; e++) {
    const int chd = (pt_x(p[e]) < sx ? 0 : 1) + (pt_y(p[e]) < sy ? 0 : 2);
    acc += e < nd.count ? 1u << (chd << 3) : 0u;
  }
  return make_int4((int)(acc & 0xffu), (int)((acc >> 8) & 0xffu), (int)((acc >> 16) & 0xffu), (int)(acc >> 24));
}
__device__ __forceinline__ void thread_split_scatter(const QNode nd, const int4 c, uint32_t* cur, uint32_t* nxt) {
  const int sx = nd.x0 + ((nd.x1 - nd.x0 + 1) >> 1);
  const int sy = nd.y0 + ((nd.y1 - nd.y0 + 1) >> 1);
  uint32_t run = ((uint32_t)c.x << 8) | ((uint32_t)(c.x + c.y) << 16) | ((uint32_t)(c.x + c.y + c.z) << 24);  // next free offset of each child
  // the node's points live in registers between the read and the two writes: no scratch buffer, no copy back
  uint32_t p[kQtThreadNode];
#pragma unroll
  for (int e = 0; e < kQtThreadNode; e++) p[e] = cur[nd.start + min(e, nd.count - 1)];
#pragma unroll
  for (int e = 0; e < kQtThreadNode; e++) {
    if (e < nd.count) {
      const int sh = ((pt_x(p[e]) < sx ? 0 : 1) + (pt_y(p[e]) < sy ? 0 : 2)) << 3;
      cur[nd.start + (int)((run >> sh) & 0xffu)] = p[e];
      run += 1u << sh;
    }
  }
}

// Full pass over LDS-resident points, thread per point: the packed (4 x 16 bit) exclusive count of the four child classes over
// the points [0, pos) of the current arrangement, from the per-chunk prefix `cpre` and the per-chunk class ballots `cbal`
// (chunk = 64 consecutive positions).  pos may equal n: chunk index n >> 6 is then one past the last written chunk, whose
// ballots are masked out by lt = 0 (the tables hold one spare entry).
__device__ __forceinline__ unsigned long long qt_prefix_at(const unsigned long long* cpre, const unsigned long long* cbal, int pos) {
  const int c = pos >> 6, l = pos & 63;
  const unsigned long long lt = (1ull << l) - 1ull;
  const unsigned long long* b = cbal + 4 * c;
  return cpre[c] + ((unsigned long long)__popcll(b[0] & lt) | (unsigned long long)__popcll(b[1] & lt) << 16 |
                    (unsigned long long)__popcll(b[2] & lt) << 32 | (unsigned long long)__popcll(b[3] & lt) << 48);
}

__device__ __forceinline__ unsigned long long expand_elem(const QNode n, int pos) {
  const uint32_t key = ((uint32_t)n.count << 13) | (uint32_t)(n.x0 & 0x1fff);
  return ((unsigned long long)key << 32) | (uint32_t)pos;
}

// Body of k_quadtree; cur / nxt are the two point buffers of this (frame, level): LDS when the level's candidates fit
// (LP, the usual case: every access is then an LDS access instead of an L2 round trip), global memory otherwise.
template <bool LP>
__device__ __forceinline__ void quadtree_body(const DeviceGeom* __restrict__ g, const CellGeom* __restrict__ cells,
                                              const uint32_t* __restrict__ fcand, uint32_t* gcur, uint32_t* gnxt,
                                              uint32_t* lcur, uint32_t* lnxt, uint32_t* __restrict__ lvl_kp,
                                              int32_t* __restrict__ lvl_n, int node_cap, int scan_cap, uint8_t* smem, int n,
                                              unsigned long long* wt, int* sh_cnt, int* sh_jstar_p, int level_base, int pts_cap) {
  // LP with lcur == nullptr: the OVERFLOW form of the LDS algorithm — a level with more candidates than the LDS point buffers hold (up to
  // twice as many, at most 4096) keeps its points in the HBM ping-pong buffers (L2-resident) and uses the whole LDS point area for the
  // node-index arrays and chunk tables of that many points: the same thread-per-point passes, slower point accesses — instead of the
  // wave-per-node fallback below (natural images put 1 600 - 2 300 corners on level 0 where the synthetic stream puts 800)
  const bool overflow = LP && lcur == nullptr;
  uint32_t* cur = (LP && !overflow) ? lcur : gcur;
  uint32_t* nxt = (LP && !overflow) ? lnxt : gnxt;
  if (overflow) pts_cap *= 2;
  // LP only: node index of every point (list position of the node that holds it) and the chunk tables of the full passes
  uint16_t* nid = overflow ? (uint16_t*)lnxt : (uint16_t*)(lnxt + pts_cap);   // overflow: lnxt = start of the LDS point area
  uint16_t* nidn = nid + pts_cap;
  unsigned long long* cpre = (unsigned long long*)(nidn + pts_cap);
  unsigned long long* cbal = cpre + (pts_cap >> 6) + 1;
  unsigned long long* cpx = cbal + 4 * ((pts_cap >> 6) + 1);   // fused pass: exclusive chunk prefix (cpre then keeps the raw counts)
  int spar = 0;                                                // parity of block_scan1's wave-total buffer
  QNode* LA = (QNode*)smem;
  QNode* LB = LA + node_cap;
  unsigned long long* EA = (unsigned long long*)(LB + node_cap);
  unsigned long long* EB = EA + node_cap;
  unsigned long long* scan = EB + node_cap;
  int4* kids = (int4*)(scan + scan_cap);
  int* flag = (int*)(kids + node_cap);
  int& sh_jstar = *sh_jstar_p;

  const int T = blockDim.x, t = threadIdx.x, lane = t & 63, w = t >> 6, NW = T >> 6;
  // level_base bit 8: level-major launch (grid = frames x levels: all workgroups of the largest level start first — longest first)
  const int level = (level_base & 0xff) + ((level_base & 0x100) ? (int)blockIdx.y : (int)blockIdx.x), frame = (level_base & 0x100) ? (int)blockIdx.x : (int)blockIdx.y;
  const DeviceLevel& lv = g->lv[level];
  const int N = lv.quota;
  // node arrays hold node_cap entries; the list never grows beyond min(N + 3, n) nodes (it stops at N, and every node
  // keeps at least one point).  The host may have clamped node_cap (16-bit positions in the sort): refuse, loudly, what
  // does not fit (negative count -> ORBX_E_CAPACITY)
  if (min(N + 4 * kMaxRoots + 8, n + 4 * kMaxRoots + 8) > node_cap) {  // block-uniform
    if (t == 0) lvl_n[frame * g->nlevels + level] = -max(n, 1);
    return;
  }
  QT_T0();
  // ---- gather the per-cell lists in cell order (vToDistributeKeys, src/ORBextractor.cc:863-868): `scan` holds the
  // exclusive prefix of the cell counts (computed by the kernel); element e finds its cell by binary search
  for (int c = t; c < lv.ncells; c += T) {   // a thread per cell copies the cell's (few) candidates
    const int base = (int)scan[c], cnt = (c + 1 < lv.ncells ? (int)scan[c + 1] : n) - base;
    const uint32_t* src = fcand + cells[lv.cell_begin + c].slot_off;
    for (int k0 = 0; k0 < cnt; k0 += 8) {   // eight independent loads, then the stores (a plain copy loop would wait for each load)
      uint32_t r[8];
#pragma unroll
      for (int k = 0; k < 8; k++) r[k] = src[min(k0 + k, cnt - 1)];
#pragma unroll
      for (int k = 0; k < 8; k++) if (k0 + k < cnt) cur[base + k0 + k] = r[k];
    }
  }
  if constexpr (LP) for (int e = t; e < n; e += T) nid[e] = 0;
  // the fused first passes (below) count, per 64-point chunk, the points of every depth-3 class: the table starts from zero
  const int fz_nch = (n + 63) >> 6;
  const bool fused_ok = LP && !(level_base & 0x200) && lv.nroots == 1 && n <= 2048 && fz_nch * 128 <= node_cap * (int)sizeof(QNode) &&
                        fz_nch * 64 <= node_cap * (int)sizeof(int4) && node_cap >= 64;   // block-uniform
  if (fused_ok) for (int e = t; e < fz_nch * 16; e += T) ((uint32_t*)kids)[e] = 0u;
  __syncthreads();
  int nL = 0, nE = 0;
  // ---- root nodes (src/ORBextractor.cc:559-601)
  const int H = lv.h - 2 * kBorder;
  if (lv.nroots == 1) {
    if (t == 0) { QNode r; r.x0 = lv.root_x0[0]; r.y0 = 0; r.x1 = lv.root_x1[0]; r.y1 = H; r.start = 0; r.count = n; LA[0] = r; }
    nL = 1;
  } else {
    if (t < kMaxRoots) sh_cnt[t] = 0;
    __syncthreads();
    for (int i = t; i < n; i += T) {
      const int b = (int)__fdiv_rn((float)pt_x(cur[i]), lv.hX);
      atomicAdd(&sh_cnt[b], 1);
    }
    __syncthreads();
    if (w == 0) {  // stable bucket scatter by one wave
      int run[kMaxRoots], kix[kMaxRoots];
      int acc = 0, kk = 0;
      for (int b = 0; b < lv.nroots; b++) { run[b] = acc; acc += sh_cnt[b]; kix[b] = kk; kk += sh_cnt[b] > 0; }
      const unsigned long long lt = (1ull << lane) - 1ull;
      for (int o = 0; o < n; o += 64) {
        const int e = o + lane;
        int bk = -1;
        uint32_t p = 0;
        if (e < n) { p = cur[e]; bk = (int)__fdiv_rn((float)pt_x(p), lv.hX); }
        for (int b = 0; b < lv.nroots; b++) {
          const unsigned long long m = __ballot(bk == b);
          if (bk == b) {
            const int dst = run[b] + __popcll(m & lt);
            nxt[dst] = p;
            if constexpr (LP) nidn[dst] = (uint16_t)kix[b];
          }
          run[b] += __popcll(m);
        }
      }
      if (lane == 0) {
        int k = 0, off = 0;
        for (int b = 0; b < lv.nroots; b++) {
          if (sh_cnt[b] > 0) {
            QNode r; r.x0 = lv.root_x0[b]; r.y0 = 0; r.x1 = lv.root_x1[b]; r.y1 = H; r.start = off; r.count = sh_cnt[b];
            LA[k++] = r;
          }
          off += sh_cnt[b];
        }
        sh_jstar = k;
      }
    }
    __syncthreads();
    nL = sh_jstar;
    { uint32_t* tmp = cur; cur = nxt; nxt = tmp; }
    { uint16_t* tn = nid; nid = nidn; nidn = tn; }
    __syncthreads();
  }
  __syncthreads();

  QT_ACC(0);
  bool finish = false, sorted_phase = false;
  if constexpr (LP) {
    if (fused_ok) {
      // ======== the first (up to three) full passes in ONE pass over the points.  The quadrant a point falls into at depth 1, 2, 3 is a
      // function of its coordinates and the root alone (DivideNode's bounds are midpoints, src/ORBextractor.cc:480-536), so one sweep
      // classifies every point to depth 3 and counts the classes per 64-point chunk; ONE wave then replays the reference's list
      // evolution pass by pass on those counts (which nodes split, where their children enter the list, the stop tests of :683-690)
      // with a lane per depth-3 class, and one more sweep moves the points (stable, by the class of the depth actually reached).
      // Three barriers instead of twelve.  State on exit = the state the generic passes below would have left.
      uint8_t* cc = (uint8_t*)kids;                 // [nch][64] points of (chunk, class)          (zeroed in the gather phase)
      uint16_t* cpf = (uint16_t*)LB;                // [nch][64] exclusive prefix over the chunks, per class (later: per depth-D class)
      uint16_t* tpos = (uint16_t*)flag;             // [64] list position of the node that holds class l
      uint16_t* tstart = tpos + 64;                 // [64] first position of class l's depth-D group in the sorted point array
      const int RX0 = lv.root_x0[0], RX1 = lv.root_x1[0];
      const unsigned long long ltm = (1ull << lane) - 1ull;
      // ---- A: classify, count
      for (int c = w; c < fz_nch; c += NW) {
        const int i = (c << 6) + lane;
        int code = 0;
        const bool valid = i < n;
        if (valid) {
          const uint32_t p = cur[i];
          const int px = pt_x(p), py = pt_y(p);
          int x0 = RX0, x1 = RX1, y0 = 0, y1 = H;
#pragma unroll
          for (int d = 0; d < 3; d++) {
            const int sx = x0 + ((x1 - x0 + 1) >> 1), sy = y0 + ((y1 - y0 + 1) >> 1);
            const int cx = px < sx ? 0 : 1, cy = py < sy ? 0 : 1;
            if (cx) x0 = sx; else x1 = sx;
            if (cy) y0 = sy; else y1 = sy;
            code = (code << 2) | cx | (cy << 1);
          }
          nid[i] = (uint16_t)code;
        }
        unsigned long long m = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 6; b++) {
          const unsigned long long bb = __ballot((code >> b) & 1);
          m &= ((code >> b) & 1) ? bb : ~bb;
        }
        if (valid && (m & ltm) == 0ull) cc[(c << 6) + code] = (uint8_t)__popcll(m);   // the first lane of every class present in the chunk
      }
      __syncthreads();
      // ---- B: one wave replays the passes on the counts
      int* fz = (int*)wt;   // [0] nL  [1] nE  [2] D | finish << 8 | sorted << 9   (wt is free here: no block scan is in flight)
      if (w == 0) {
        // exclusive prefix over the chunks, per depth-3 class (lane = class); h3 = points of the class
        int h3 = 0;
        for (int c = 0; c < fz_nch; c++) { const int v = cc[(c << 6) + lane]; cpf[(c << 6) + lane] = (uint16_t)h3; h3 += v; }
        // class counts at depth 2 (quads of lanes) and depth 1 (groups of 16), in every lane of the group
        int h2 = h3; h2 += __shfl_xor(h2, 1); h2 += __shfl_xor(h2, 2);
        int h1 = h2; h1 += __shfl_xor(h1, 4); h1 += __shfl_xor(h1, 8);
        int inc = h3;   // lexicographic start of every depth-3 class in the sorted point array
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o); if (lane >= o) inc += u; }
        const int start3 = inc - h3;
        // per lane: the list node its class currently belongs to
        bool alive = true;
        int ndepth = 0, ncount = n, npos = 0, eidx = -1;
        int curL = 1, curE = 0, D = 0;
        bool fin = false, srt = false;
#pragma unroll 1
        for (int p = 1; p <= 3 && !fin && !srt; p++) {
          const int sp = 2 * (3 - p);                       // lanes of one depth-p class share lane >> sp
          const int hp = p == 1 ? h1 : p == 2 ? h2 : h3;     // points of my depth-p class
          const bool split = alive && ncount > 1, nomore = alive && ncount == 1;
          // my parent's four children
          const int pbase = (lane >> (sp + 2)) << 2, cme = (lane >> sp) & 3;
          int k = 0, q = 0, ci = 0, qi = 0;
#pragma unroll
          for (int c4 = 0; c4 < 4; c4++) {
            const int sc = __shfl(hp, (pbase | c4) << sp);
            k += sc > 0; q += sc > 1;
            ci += (c4 < cme) && sc > 0; qi += (c4 < cme) && sc > 1;
          }
          // one value per list node, at the lane of its position: children (k), expandable children (q), no-more flag — then their
          // exclusive prefixes in list order.  The representative lane of a node is the first lane of its class group.
          const int gmask = (1 << (2 * (3 - ndepth))) - 1;
          const bool rep = alive && (lane & gmask) == 0;
          uint16_t* inv = tstart;   // scratch until the tables are written: position -> representative lane
          if (rep) inv[npos] = (uint16_t)lane;
          wave_lds_sync();
          const int src = lane < curL ? (int)inv[lane] : 0;
          wave_lds_sync();
          const int packed = split ? (k | (q << 8)) : nomore ? (1 << 16) : 0;
          int vpos = __shfl(packed, src);                    // the node at list position `lane`
          if (lane >= curL) vpos = 0;
          int vinc = vpos;
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(vinc, o); if (lane >= o) vinc += u; }
          const int vex = vinc - vpos, vtot = __shfl(vinc, 63);
          const int mine = __shfl(vex, npos);                // prefixes at my node's position
          const int totalKids = vtot & 0xff, totalQ = (vtot >> 8) & 0xff, totalM = vtot >> 16;
          const int kpre = mine & 0xff, qpre = (mine >> 8) & 0xff, spre = mine >> 16;
          // my class after this pass
          if (split) {
            if (hp == 0) alive = false;
            else {
              ndepth = p; ncount = hp;
              npos = totalKids - (kpre + k) + (k - 1 - ci);
              eidx = hp > 1 ? qpre + qi : -1;
            }
          } else if (nomore) { npos = totalKids + spre; eidx = -1; }
          const int newL = totalKids + totalM;
          D = p;
          fin = newL >= N || newL == curL;
          srt = !fin && newL + 3 * totalQ > N;
          curL = newL; curE = totalQ;
        }
        // ---- the state the generic passes would have left: list, expandable list, tables for the point sweep
        const int sD = 2 * (3 - D);
        const int gmask = (1 << (2 * (3 - ndepth))) - 1;
        const bool rep = alive && (lane & gmask) == 0;
        if (rep) {
          int x0 = RX0, x1 = RX1, y0 = 0, y1 = H;
          for (int d = 1; d <= ndepth; d++) {
            const int c4 = (lane >> (2 * (3 - d))) & 3;
            const int sx = x0 + ((x1 - x0 + 1) >> 1), sy = y0 + ((y1 - y0 + 1) >> 1);
            if (c4 & 1) x0 = sx; else x1 = sx;
            if (c4 & 2) y0 = sy; else y1 = sy;
          }
          QNode nd;
          nd.x0 = (int16_t)x0; nd.y0 = (int16_t)y0; nd.x1 = (int16_t)x1; nd.y1 = (int16_t)y1; nd.start = start3; nd.count = ncount;
          LA[npos] = nd;
          if (ndepth == D && eidx >= 0) EA[eidx] = expand_elem(nd, npos);
        }
        wave_lds_sync();   // inv (in tstart) has been read by everyone
        tpos[lane] = (uint16_t)npos;
        tstart[lane] = (uint16_t)__shfl(start3, (lane >> sD) << sD);
        if (D < 3)   // the points are sorted by their depth-D class: prefix over the chunks per depth-D class
          for (int c = 0; c < fz_nch; c++) {
            int v = cpf[(c << 6) + lane];
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2);
            if (D < 2) { v += __shfl_xor(v, 4); v += __shfl_xor(v, 8); }
            cpf[(c << 6) + lane] = (uint16_t)v;
          }
        if (lane == 0) { fz[0] = curL; fz[1] = curE; fz[2] = D | (fin ? 0x100 : 0) | (srt ? 0x200 : 0); }
      }
      __syncthreads();
      nL = fz[0]; nE = fz[1];
      const int D = fz[2] & 0xff;
      finish = (fz[2] & 0x100) != 0; sorted_phase = (fz[2] & 0x200) != 0;
      // ---- C: the points move (stable inside a depth-D class) and learn their node's list position
      for (int c = w; c < fz_nch; c += NW) {
        const int i = (c << 6) + lane;
        const bool valid = i < n;
        const int code = valid ? (int)nid[i] : 0;
        unsigned long long m = __ballot(valid);
        for (int b = 5; b >= 6 - 2 * D; b--) {
          const unsigned long long bb = __ballot((code >> b) & 1);
          m &= ((code >> b) & 1) ? bb : ~bb;
        }
        if (valid) {
          const int dst = (int)tstart[code] + (int)cpf[(c << 6) + code] + __popcll(m & ltm);
          nxt[dst] = cur[i];
          nidn[dst] = tpos[code];
        }
      }
      __syncthreads();
      { uint16_t* tn = nid; nid = nidn; nidn = tn; }
      { uint32_t* tp = cur; cur = nxt; nxt = tp; }
      QT_ACC(1);
    }
  }
  while (!finish) {
    if (!sorted_phase) {
      // ======== full pass: split every node holding more than one point (src/ORBextractor.cc:612-681)
      const int prevSize = nL;
      const int nch = (n + 63) >> 6;
      if (LP && nL <= T) {   // block-uniform
        // ---- the usual case, four barriers: at most one node per thread, so the node's scan value, its prefix and its children
        // stay in that thread's registers (no scan arrays, no separate prefix pass over the chunks)
        for (int c = w; c < nch; c += NW) {
          const int i = (c << 6) + lane;
          int chd = 4;
          if (i < n) {
            const QNode nd = LA[nid[i]];
            if (nd.count > 1) {
              const uint32_t p = cur[i];
              chd = (pt_x(p) < nd.x0 + ((nd.x1 - nd.x0 + 1) >> 1) ? 0 : 1) + (pt_y(p) < nd.y0 + ((nd.y1 - nd.y0 + 1) >> 1) ? 0 : 2);
            }
          }
          const unsigned long long b0 = __ballot(chd == 0), b1 = __ballot(chd == 1), b2 = __ballot(chd == 2), b3 = __ballot(chd == 3);
          if (lane == 0) {
            cbal[4 * c] = b0; cbal[4 * c + 1] = b1; cbal[4 * c + 2] = b2; cbal[4 * c + 3] = b3;
            cpre[c] = (unsigned long long)__popcll(b0) | (unsigned long long)__popcll(b1) << 16 |
                      (unsigned long long)__popcll(b2) << 32 | (unsigned long long)__popcll(b3) << 48;
          }
        }
        __syncthreads();
        // every wave scans the (<= 64) chunk counts for itself; wave 0 leaves the exclusive prefix in LDS for the scatter below
        const unsigned long long cv = lane < nch ? cpre[lane] : 0ull;
        unsigned long long cinc = cv;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const unsigned long long u = __shfl_up(cinc, o);
          if (lane >= o) cinc += u;
        }
        const unsigned long long cex = cinc - cv;
        if (w == 0 && lane < nch) cpx[lane] = cex;
        QNode nd;
        nd.count = 0;
        unsigned long long sv = 0, s0 = 0, dcnt = 0;
        {
          int ps = 0, pe = 0;
          if (t < nL) { nd = LA[t]; ps = nd.start; pe = nd.start + nd.count; }
          // prefix at both ends of the node (all lanes take part in the lane reads; position n lies one chunk past the last)
          const int c0i = ps >> 6, c1i = pe >> 6;
          const unsigned long long e0 = __shfl(cex, min(c0i, nch - 1)), e1a = __shfl(cex, min(c1i, nch - 1)), tot = __shfl(cinc, nch - 1);
          if (t < nL) {
            sv = 1ull << 42;
            if (nd.count > 1) {
              const unsigned long long l0 = (1ull << (ps & 63)) - 1ull, l1 = (1ull << (pe & 63)) - 1ull;
              const unsigned long long* q0 = cbal + 4 * c0i;
              const unsigned long long* q1 = cbal + 4 * c1i;   // may be the spare entry: masked by l1 = 0
              s0 = e0 + ((unsigned long long)__popcll(q0[0] & l0) | (unsigned long long)__popcll(q0[1] & l0) << 16 |
                         (unsigned long long)__popcll(q0[2] & l0) << 32 | (unsigned long long)__popcll(q0[3] & l0) << 48);
              const unsigned long long s1 = (c1i < nch ? e1a : tot) +
                        ((unsigned long long)__popcll(q1[0] & l1) | (unsigned long long)__popcll(q1[1] & l1) << 16 |
                         (unsigned long long)__popcll(q1[2] & l1) << 32 | (unsigned long long)__popcll(q1[3] & l1) << 48);
              dcnt = s1 - s0;
              const int c0 = (int)(dcnt & 0xffffu), c1 = (int)((dcnt >> 16) & 0xffffu), c2 = (int)((dcnt >> 32) & 0xffffu), c3 = (int)(dcnt >> 48);
              const unsigned long long k = (c0 > 0) + (c1 > 0) + (c2 > 0) + (c3 > 0);
              const unsigned long long q = (c0 > 1) + (c1 > 1) + (c2 > 1) + (c3 > 1);
              sv = k | (q << 21);
            }
          }
        }
        unsigned long long tot;
        const unsigned long long pre = block_scan1(sv, wt, spar, tot);
        spar ^= 1;
        const int totalKids = (int)(tot & kM21), nToExpand = (int)((tot >> 21) & kM21), nNoMore = (int)(tot >> 42);
        if (t < nL) {
          const int kpre = (int)(pre & kM21), qpre = (int)((pre >> 21) & kM21), spre = (int)(pre >> 42);
          unsigned long long cpos = 0;
          if (nd.count > 1) {
            const int cnt[4] = {(int)(dcnt & 0xffffu), (int)((dcnt >> 16) & 0xffffu), (int)((dcnt >> 32) & 0xffffu), (int)(dcnt >> 48)};
            const int k = (cnt[0] > 0) + (cnt[1] > 0) + (cnt[2] > 0) + (cnt[3] > 0);
            int ci = 0, qi = 0, st = nd.start;
#pragma unroll
            for (int ch = 0; ch < 4; ch++) {
              if (cnt[ch] > 0) {
                const int pos = totalKids - (kpre + k) + (k - 1 - ci);
                const QNode cn = child_node(nd, ch, st, cnt[ch]);
                LB[pos] = cn;
                cpos |= (unsigned long long)pos << (16 * ch);
                if (cnt[ch] > 1) { EB[qpre + qi] = expand_elem(cn, pos); qi++; }
                ci++;
              }
              st += cnt[ch];
            }
            kids[t] = make_int4((int)(uint32_t)dcnt, (int)(uint32_t)(dcnt >> 32), (int)(uint32_t)s0, (int)(uint32_t)(s0 >> 32));
          } else {
            LB[totalKids + spre] = nd;
            cpos = (unsigned long long)(totalKids + spre);
          }
          scan[t] = cpos;
        }
        __syncthreads();
        for (int i = t; i < n; i += T) {
          const int id = nid[i];
          const QNode pn = LA[id];
          const uint32_t p = cur[i];
          const unsigned long long cpos = scan[id];
          if (pn.count > 1) {
            const int chd = (pt_x(p) < pn.x0 + ((pn.x1 - pn.x0 + 1) >> 1) ? 0 : 1) + (pt_y(p) < pn.y0 + ((pn.y1 - pn.y0 + 1) >> 1) ? 0 : 2);
            const int sh = chd << 4;
            const int4 kp = kids[id];
            const unsigned long long cnts = (unsigned long long)(uint32_t)kp.x | (unsigned long long)(uint32_t)kp.y << 32;
            const unsigned long long base = (unsigned long long)(uint32_t)kp.z | (unsigned long long)(uint32_t)kp.w << 32;
            const int c = i >> 6;
            const int mine = (int)((cpx[c] >> sh) & 0xffffu) + __popcll(cbal[4 * c + chd] & ((1ull << (i & 63)) - 1ull));
            const unsigned long long below = cnts & ((1ull << sh) - 1ull);
            const int off = (int)(below & 0xffffu) + (int)((below >> 16) & 0xffffu) + (int)((below >> 32) & 0xffffu);
            const int dst = pn.start + off + mine - (int)((base >> sh) & 0xffffu);
            nxt[dst] = p;
            nidn[dst] = (uint16_t)(cpos >> sh);
          } else {
            nxt[i] = p;
            nidn[i] = (uint16_t)cpos;
          }
        }
        __syncthreads();
        { uint16_t* tn = nid; nid = nidn; nidn = tn; }
        { QNode* tl = LA; LA = LB; LB = tl; }
        { unsigned long long* te = EA; EA = EB; EB = te; }
        { uint32_t* tp = cur; cur = nxt; nxt = tp; }
        nL = totalKids + nNoMore;
        nE = nToExpand;
        if (nL >= N || nL == prevSize) finish = true;
        else if (nL + 3 * nE > N) sorted_phase = true;
        QT_ACC(1);
        continue;
      }
      if constexpr (LP) {
        // thread per point.  Class ballots and counts of every 64-position chunk ...
        for (int c = w; c < nch; c += NW) {
          const int i = (c << 6) + lane;
          int chd = 4;
          if (i < n) {
            const QNode nd = LA[nid[i]];
            if (nd.count > 1) {
              const uint32_t p = cur[i];
              chd = (pt_x(p) < nd.x0 + ((nd.x1 - nd.x0 + 1) >> 1) ? 0 : 1) + (pt_y(p) < nd.y0 + ((nd.y1 - nd.y0 + 1) >> 1) ? 0 : 2);
            }
          }
          const unsigned long long b0 = __ballot(chd == 0), b1 = __ballot(chd == 1), b2 = __ballot(chd == 2), b3 = __ballot(chd == 3);
          if (lane == 0) {
            cbal[4 * c] = b0; cbal[4 * c + 1] = b1; cbal[4 * c + 2] = b2; cbal[4 * c + 3] = b3;
            cpre[c] = (unsigned long long)__popcll(b0) | (unsigned long long)__popcll(b1) << 16 |
                      (unsigned long long)__popcll(b2) << 32 | (unsigned long long)__popcll(b3) << 48;
          }
        }
        __syncthreads();
        // ... their exclusive prefix over the chunks (at most 64 of them: one wave) ...
        if (w == 0) {
          const unsigned long long v = lane < nch ? cpre[lane] : 0ull;
          unsigned long long inc = v;
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long u = __shfl_up(inc, o);
            if (lane >= o) inc += u;
          }
          if (lane < nch) cpre[lane] = inc - v;
          if (lane == nch - 1) cpre[nch] = inc;
        }
        __syncthreads();
        // ... give every node its child counts (difference of the prefix at its two ends) and the prefix at its start
        for (int i = t; i < nL; i += T) {
          const QNode nd = LA[i];
          unsigned long long v = 1ull << 42;
          if (nd.count > 1) {
            const unsigned long long s0 = qt_prefix_at(cpre, cbal, nd.start), s1 = qt_prefix_at(cpre, cbal, nd.start + nd.count);
            const unsigned long long d = s1 - s0;  // no field borrows: every class count is monotone in the position
            kids[i] = make_int4((int)(uint32_t)d, (int)(uint32_t)(d >> 32), (int)(uint32_t)s0, (int)(uint32_t)(s0 >> 32));
            const int c0 = (int)(d & 0xffffu), c1 = (int)((d >> 16) & 0xffffu), c2 = (int)((d >> 32) & 0xffffu), c3 = (int)(d >> 48);
            const unsigned long long k = (c0 > 0) + (c1 > 0) + (c2 > 0) + (c3 > 0);
            const unsigned long long q = (c0 > 1) + (c1 > 1) + (c2 > 1) + (c3 > 1);
            v = k | (q << 21);
          }
          scan[i] = v;
        }
        __syncthreads();
      } else {
      for (int i = w; i < nL; i += NW) {
        const QNode nd = LA[i];
        if (nd.count > 1) {
          const int4 c = wave_split(nd, cur, nxt, true);
          if (lane == 0) kids[i] = c;
        } else if (lane == 0) {
          nxt[nd.start] = cur[nd.start];
        }
      }
      __syncthreads();
      for (int i = t; i < nL; i += T) {
        unsigned long long v;
        if (LA[i].count > 1) {
          const int4 c = kids[i];
          const unsigned long long k = (c.x > 0) + (c.y > 0) + (c.z > 0) + (c.w > 0);
          const unsigned long long q = (c.x > 1) + (c.y > 1) + (c.z > 1) + (c.w > 1);
          v = k | (q << 21);
        } else {
          v = 1ull << 42;
        }
        scan[i] = v;
      }
      __syncthreads();
      }
      const unsigned long long tot = block_excl_scan(scan, nL, wt);
      const int totalKids = (int)(tot & kM21), nToExpand = (int)((tot >> 21) & kM21), nNoMore = (int)(tot >> 42);
      for (int i = t; i < nL; i += T) {
        const unsigned long long pre = scan[i];
        const int kpre = (int)(pre & kM21), qpre = (int)((pre >> 21) & kM21), spre = (int)(pre >> 42);
        const QNode nd = LA[i];
        unsigned long long cpos = 0;  // LP: list positions of the four children (16 bits each), or of the node itself
        if (nd.count > 1) {
          const int4 c = kids[i];
          int cnt[4];
          if constexpr (LP) { cnt[0] = c.x & 0xffff; cnt[1] = (int)((uint32_t)c.x >> 16); cnt[2] = c.y & 0xffff; cnt[3] = (int)((uint32_t)c.y >> 16); }
          else { cnt[0] = c.x; cnt[1] = c.y; cnt[2] = c.z; cnt[3] = c.w; }
          const int k = (cnt[0] > 0) + (cnt[1] > 0) + (cnt[2] > 0) + (cnt[3] > 0);
          int ci = 0, qi = 0, st = nd.start;
#pragma unroll
          for (int ch = 0; ch < 4; ch++) {
            if (cnt[ch] > 0) {
              const int pos = totalKids - (kpre + k) + (k - 1 - ci);
              const QNode cn = child_node(nd, ch, st, cnt[ch]);
              LB[pos] = cn;
              cpos |= (unsigned long long)pos << (16 * ch);
              if (cnt[ch] > 1) { EB[qpre + qi] = expand_elem(cn, pos); qi++; }
              ci++;
            }
            st += cnt[ch];
          }
        } else {
          LB[totalKids + spre] = nd;
          cpos = (unsigned long long)(totalKids + spre);
        }
        if constexpr (LP) scan[i] = cpos;  // this thread was the only reader of scan[i]
      }
      __syncthreads();
      if constexpr (LP) {
        // every point moves to its child's segment (stable: rank among the node's points of the same class) and learns the
        // child's list position
        for (int i = t; i < n; i += T) {
          const int id = nid[i];
          const QNode nd = LA[id];
          const uint32_t p = cur[i];
          const unsigned long long cpos = scan[id];
          if (nd.count > 1) {
            const int chd = (pt_x(p) < nd.x0 + ((nd.x1 - nd.x0 + 1) >> 1) ? 0 : 1) + (pt_y(p) < nd.y0 + ((nd.y1 - nd.y0 + 1) >> 1) ? 0 : 2);
            const int sh = chd << 4;
            const int4 kp = kids[id];
            const unsigned long long cnts = (unsigned long long)(uint32_t)kp.x | (unsigned long long)(uint32_t)kp.y << 32;
            const unsigned long long base = (unsigned long long)(uint32_t)kp.z | (unsigned long long)(uint32_t)kp.w << 32;
            const int c = i >> 6;
            const int mine = (int)((cpre[c] >> sh) & 0xffffu) + __popcll(cbal[4 * c + chd] & ((1ull << (i & 63)) - 1ull));
            const unsigned long long below = cnts & ((1ull << sh) - 1ull);
            const int off = (int)(below & 0xffffu) + (int)((below >> 16) & 0xffffu) + (int)((below >> 32) & 0xffffu);
            const int dst = nd.start + off + mine - (int)((base >> sh) & 0xffffu);
            nxt[dst] = p;
            nidn[dst] = (uint16_t)(cpos >> sh);
          } else {
            nxt[i] = p;
            nidn[i] = (uint16_t)cpos;
          }
        }
        __syncthreads();
        { uint16_t* tn = nid; nid = nidn; nidn = tn; }
      }
      { QNode* tl = LA; LA = LB; LB = tl; }
      { unsigned long long* te = EA; EA = EB; EB = te; }
      { uint32_t* tp = cur; cur = nxt; nxt = tp; }
      nL = totalKids + nNoMore;
      nE = nToExpand;
      if (nL >= N || nL == prevSize) finish = true;
      else if (nL + 3 * nE > N) sorted_phase = true;
      QT_ACC(1);
    } else {
      // ======== sorted expansion (src/ORBextractor.cc:692-753)
      const int prevSize = nL;
      const int m = nE;
      block_gnu_sort(EA, m, EB, (uint32_t*)flag, scan, scan + (scan_cap >> 1), (uint16_t*)kids, sh_cnt);
      QT_ACC(2);
      int* big = (int*)EB;
      if (m <= T && nL <= T) {   // block-uniform
        // ---- the usual case, six barriers: thread t owns the node at sorted position m - 1 - t (the order in which the reference
        // walks them) and list position t; counts, scan values and prefixes stay in registers
        const int j = m - 1 - t;
        QNode nd;
        nd.count = 0;
        int4 c = make_int4(0, 0, 0, 0);
        bool isbig = false;
        if (t == 0) sh_jstar = 0;
        if (t < nL) flag[t] = 0;
        if (t < m) {
          nd = LA[(uint32_t)EA[j]];
          isbig = nd.count > kQtThreadNode;
          if (isbig) big[atomicAdd(&sh_cnt[0], 1)] = j;   // the sort left sh_cnt[0] = 0
          else c = thread_split_count(nd, cur);
        }
        int nbig = 0;
        if (__syncthreads_or(isbig)) {   // large nodes (rare in this phase) take a wave each
          nbig = sh_cnt[0];
          for (int b = w; b < nbig; b += NW) {
            const int jb = big[b];
            const int4 cb = wave_split(LA[(uint32_t)EA[jb]], cur, nxt, false);
            if (lane == 0) kids[jb] = cb;
          }
          __syncthreads();
          if (isbig) c = kids[j];
        }
        const int kc = (c.x > 0) + (c.y > 0) + (c.z > 0) + (c.w > 0);
        const int qc = (c.x > 1) + (c.y > 1) + (c.z > 1) + (c.w > 1);
        unsigned long long tot1, tot2, tot3;
        const unsigned long long pre1 = block_scan1(t < m ? (unsigned long long)(kc - 1) : 0ull, wt, spar, tot1);
        spar ^= 1;
        if (t < m) {  // the reference stops as soon as the list holds N nodes (:746-751): the first position whose split gets there
          const int before = nL + (int)pre1, after = before + kc - 1;
          if (after >= N && before < N) sh_jstar = j;
        }
        __syncthreads();
        const int jstar = sh_jstar;
        const bool proc = t < m && j >= jstar;
        if (proc) {
          flag[(uint32_t)EA[j]] = j + 1;
          if (!isbig) thread_split_scatter(nd, c, cur, nxt);
        }
        for (int b = w; b < nbig; b += NW) {
          const int jb = big[b];
          if (jb < jstar) continue;  // wave-uniform
          const QNode bn = LA[(uint32_t)EA[jb]];
          wave_split(bn, cur, nxt, true);
          wave_lds_sync();
          for (int e = lane; e < bn.count; e += 64) cur[bn.start + e] = nxt[bn.start + e];
        }
        const unsigned long long pre2 = block_scan1(proc ? ((unsigned long long)kc | (unsigned long long)qc << 21) : 0ull, wt, spar, tot2);
        spar ^= 1;
        const int front = (int)(tot2 & kM21), qtot = (int)((tot2 >> 21) & kM21);
        if (proc) {
          // the scan ran in processing order (j descending): children enter the list in front of everything created before
          // them, their expandable ones join the next round's vector in creation order
          const int basep = front - ((int)(pre2 & kM21) + kc), eoff = (int)((pre2 >> 21) & kM21);
          const int cnt[4] = {c.x, c.y, c.z, c.w};
          int ci = 0, qi = 0, st = nd.start;
#pragma unroll
          for (int ch = 0; ch < 4; ch++) {
            if (cnt[ch] > 0) {
              const int pos = basep + (kc - 1 - ci);
              const QNode cn = child_node(nd, ch, st, cnt[ch]);
              LB[pos] = cn;
              if (cnt[ch] > 1) { EB[eoff + qi] = expand_elem(cn, pos); qi++; }
              ci++;
            }
            st += cnt[ch];
          }
        }
        const bool keep = t < nL && !flag[t];
        const unsigned long long pre3 = block_scan1(keep ? 1ull : 0ull, wt, spar, tot3);
        spar ^= 1;
        if (keep) LB[front + (int)pre3] = LA[t];
        __syncthreads();
        { QNode* tl = LA; LA = LB; LB = tl; }
        { unsigned long long* te = EA; EA = EB; EB = te; }
        nL = front + (int)tot3;
        nE = qtot;
        if (nL >= N || nL == prevSize) finish = true;
        QT_ACC(3);
        continue;
      }
      // child counts of every expandable node: a thread walks a small node's points, the few large ones are listed (in EB, free
      // until the children are created) and take a wave each
      if (t == 0) sh_cnt[0] = 0;
      __syncthreads();
      for (int j = t; j < m; j += T) {
        const QNode nd = LA[(uint32_t)EA[j]];
        if (nd.count <= kQtThreadNode) kids[j] = thread_split_count(nd, cur);
        else big[atomicAdd(&sh_cnt[0], 1)] = j;
      }
      for (int i = t; i < nL; i += T) flag[i] = 0;
      __syncthreads();
      const int nbig = sh_cnt[0];
      for (int b = w; b < nbig; b += NW) {
        const int j = big[b];
        const int4 c = wave_split(LA[(uint32_t)EA[j]], cur, nxt, false);
        if (lane == 0) kids[j] = c;
      }
      __syncthreads();
      // jstar = the sorted position at which the reference stops splitting (it walks j = m-1 down and stops as soon as
      // the list would hold >= N nodes, :746-751): the inclusive prefix over r = m-1-j of (children - 1) is monotone,
      // so the stop is the first r whose prefix reaches N - nL.
      for (int r = t; r < m; r += T) {
        const int4 c = kids[m - 1 - r];
        scan[r] = (unsigned long long)((c.x > 0) + (c.y > 0) + (c.z > 0) + (c.w > 0) - 1);
      }
      if (t == 0) sh_jstar = 0;
      __syncthreads();
      block_excl_scan(scan, m, wt);
      for (int r = t; r < m; r += T) {
        const int4 c = kids[m - 1 - r];
        const int before = nL + (int)scan[r];
        const int after = before + (c.x > 0) + (c.y > 0) + (c.z > 0) + (c.w > 0) - 1;
        if (after >= N && before < N) sh_jstar = m - 1 - r;
      }
      __syncthreads();
      const int jstar = sh_jstar;
      const int mp = m - jstar;  // processed nodes: sorted positions jstar..m-1 (largest first in time)
      for (int j = jstar + t; j < m; j += T) flag[(uint32_t)EA[j]] = j + 1;
      __syncthreads();
      // only the processed nodes move points: partition into the scratch buffer, then copy the segment back in place
      // (the untouched nodes, the large majority in this phase, keep their points where they are: no buffer swap)
      for (int jj = t; jj < mp; jj += T) {
        const QNode nd = LA[(uint32_t)EA[jstar + jj]];
        if (nd.count <= kQtThreadNode) thread_split_scatter(nd, kids[jstar + jj], cur, nxt);
      }
      for (int b = w; b < nbig; b += NW) {
        const int j = big[b];
        if (j < jstar) continue;  // wave-uniform
        const QNode nd = LA[(uint32_t)EA[j]];
        wave_split(nd, cur, nxt, true);
        wave_lds_sync();
        for (int e = lane; e < nd.count; e += 64) cur[nd.start + e] = nxt[nd.start + e];
      }
      for (int jj = t; jj < mp; jj += T) {
        const int4 c = kids[jstar + jj];
        const unsigned long long k = (c.x > 0) + (c.y > 0) + (c.z > 0) + (c.w > 0);
        const unsigned long long q = (c.x > 1) + (c.y > 1) + (c.z > 1) + (c.w > 1);
        scan[jj] = k | (q << 21);
      }
      __syncthreads();
      const unsigned long long tot = block_excl_scan(scan, mp, wt);
      const int front = (int)(tot & kM21), qtot = (int)((tot >> 21) & kM21);
      for (int jj = t; jj < mp; jj += T) {
        const int j = jstar + jj;
        const unsigned long long pre = scan[jj];
        const int basep = (int)(pre & kM21), qpre = (int)((pre >> 21) & kM21);
        const QNode nd = LA[(uint32_t)EA[j]];
        const int4 c = kids[j];
        const int cnt[4] = {c.x, c.y, c.z, c.w};
        const int k = (c.x > 0) + (c.y > 0) + (c.z > 0) + (c.w > 0);
        const int q = (c.x > 1) + (c.y > 1) + (c.z > 1) + (c.w > 1);
        const int eoff = qtot - (qpre + q);  // creation order runs j = m-1 down to jstar
        int ci = 0, qi = 0, st = nd.start;
#pragma unroll
        for (int ch = 0; ch < 4; ch++) {
          if (cnt[ch] > 0) {
            const int pos = basep + (k - 1 - ci);
            const QNode cn = child_node(nd, ch, st, cnt[ch]);
            LB[pos] = cn;
            if (cnt[ch] > 1) { EB[eoff + qi] = expand_elem(cn, pos); qi++; }
            ci++;
          }
          st += cnt[ch];
        }
      }
      __syncthreads();
      for (int i = t; i < nL; i += T) scan[i] = flag[i] ? 0ull : 1ull;
      __syncthreads();
      const int keepTot = (int)block_excl_scan(scan, nL, wt);
      for (int i = t; i < nL; i += T)
        if (!flag[i]) LB[front + (int)scan[i]] = LA[i];
      __syncthreads();
      { QNode* tl = LA; LA = LB; LB = tl; }
      { unsigned long long* te = EA; EA = EB; EB = te; }
      nL = front + keepTot;
      nE = qtot;
      if (nL >= N || nL == prevSize) finish = true;
      QT_ACC(3);
    }
  }
  // ---- best point of every node, first maximum wins (src/ORBextractor.cc:757-776)
  uint32_t* out = lvl_kp + (long long)frame * g->kp_total + lv.kp_off;
  const int nout = min(nL, lv.kp_cap);
  for (int i = t; i < nout; i += T) {
    const QNode nd = LA[i];
    uint32_t best = cur[nd.start];
    for (int e = 1; e < nd.count; e++) {
      const uint32_t p = cur[nd.start + e];
      if (pt_s(p) > pt_s(best)) best = p;
    }
    out[i] = pack_pt(pt_x(best) + kBorder, pt_y(best) + kBorder, pt_s(best));
  }
  if (t == 0) lvl_n[frame * g->nlevels + level] = nL <= lv.kp_cap ? nL : -nL;  // negative = capacity overflow
  QT_ACC(4);
}

// The part of k_quadtree that touches the node arrays, instantiated once with the arrays in LDS (`nb` = the dynamic LDS
// block) and once with them in HBM (`nb` = this workgroup's slice of the context's node scratch): levels whose quota
// does not fit one CU's LDS (mpIniORBextractor of a 2000-feature configuration: 10 000 features, level-0 quota 2172)
// take the slow HBM instantiation instead of being refused.
template <bool GN>
__device__ __forceinline__ void quadtree_main(const DeviceGeom* __restrict__ g, const CellGeom* __restrict__ cells,
                                              const uint32_t* __restrict__ cand, const int32_t* __restrict__ cell_cnt,
                                              uint32_t* __restrict__ pts, uint32_t* __restrict__ lvl_kp, int32_t* __restrict__ lvl_n,
                                              int node_cap, int scan_cap, int pts_cap, int level_base, uint8_t* nb, uint32_t* lpts,
                                              unsigned long long* wt, int* sh_cnt, int* sh_jstar) {
  unsigned long long* scan = (unsigned long long*)(nb + (size_t)node_cap * (2 * sizeof(QNode) + 2 * 8));
  const int T = blockDim.x, t = threadIdx.x;
  // level_base bit 8: level-major launch (grid = frames x levels: all workgroups of the largest level start first — longest first)
  const int level = (level_base & 0xff) + ((level_base & 0x100) ? (int)blockIdx.y : (int)blockIdx.x), frame = (level_base & 0x100) ? (int)blockIdx.x : (int)blockIdx.y;
  const DeviceLevel& lv = g->lv[level];
  uint32_t* gcur = pts + ((long long)frame * 2 + 0) * g->cand_total + lv.cand_off;
  uint32_t* gnxt = pts + ((long long)frame * 2 + 1) * g->cand_total + lv.cand_off;
  const int32_t* ccnt = cell_cnt + (long long)frame * g->ncells_total + lv.cell_begin;
  const uint32_t* fcand = cand + (long long)frame * g->cand_total;
  for (int c = t; c < lv.ncells; c += T) scan[c] = (unsigned long long)ccnt[c];
  __syncthreads();
  const int n = (int)block_excl_scan(scan, lv.ncells, wt);
  if (n == 0) {
    if (t == 0) lvl_n[frame * g->nlevels + level] = 0;
    return;
  }
  if (!GN && n <= pts_cap)
    quadtree_body<true>(g, cells, fcand, gcur, gnxt, lpts, lpts + pts_cap, lvl_kp, lvl_n, node_cap, scan_cap, nb, n, wt, sh_cnt, sh_jstar, level_base, pts_cap);
  else if (!GN && pts_cap >= 64 && n <= 2 * pts_cap && n <= 4096 && !(level_base & 0x400))   // block-uniform
    quadtree_body<true>(g, cells, fcand, gcur, gnxt, (uint32_t*)nullptr, lpts, lvl_kp, lvl_n, node_cap, scan_cap, nb, n, wt, sh_cnt, sh_jstar, level_base, pts_cap);
  else
    quadtree_body<false>(g, cells, fcand, gcur, gnxt, lpts, lpts + pts_cap, lvl_kp, lvl_n, node_cap, scan_cap, nb, n, wt, sh_cnt, sh_jstar, level_base, pts_cap);
}

// (no register cap: capping at 128 VGPRs — 4 waves per SIMD — was measured at +-1 % in a batch and spills to scratch; HISTORY.md, round 5)
__global__ __launch_bounds__(512, 1) void k_quadtree(const DeviceGeom* __restrict__ g, const CellGeom* __restrict__ cells,
                                                  const uint32_t* __restrict__ cand, const int32_t* __restrict__ cell_cnt,
                                                  uint32_t* __restrict__ pts, uint32_t* __restrict__ lvl_kp,
                                                  int32_t* __restrict__ lvl_n, int node_cap, int scan_cap, int pts_cap, int level_base,
                                                  uint8_t* __restrict__ gnodes, long long gnode_stride) {
  extern __shared__ __align__(16) uint8_t smem[];
  __shared__ unsigned long long wt[16];   // block_excl_scan uses the first 8, block_scan1 alternates between the halves
  __shared__ int sh_cnt[kMaxRoots];
  __shared__ int sh_jstar;
  // LDS carve-up (see quadtree_body): LA, LB, EA, EB, scan, kids, flag, then the two LDS point buffers
  if (gnodes == nullptr) {  // block-uniform (kernel argument)
    uint32_t* lpts = (uint32_t*)(smem + (size_t)node_cap * (2 * sizeof(QNode) + 2 * 8 + sizeof(int4) + 4) + (size_t)scan_cap * 8);
    quadtree_main<false>(g, cells, cand, cell_cnt, pts, lvl_kp, lvl_n, node_cap, scan_cap, pts_cap, level_base, smem, lpts, wt, sh_cnt, &sh_jstar);
  } else {
    uint8_t* nb = gnodes + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * gnode_stride;
    quadtree_main<true>(g, cells, cand, cell_cnt, pts, lvl_kp, lvl_n, node_cap, scan_cap, 0, level_base, nb, (uint32_t*)nullptr, wt, sh_cnt, &sh_jstar);
  }
}

#ifdef ORBX_DEBUG_ABI
// Test hook: block_gnu_sort on caller data (n <= 2048), one workgroup.
__global__ __launch_bounds__(512) void k_debug_gnu_sort(unsigned long long* __restrict__ data, int n) {
  __shared__ unsigned long long sv[2048], stmp[2048], sq[2 * (2048 / 16 + 2)];
  __shared__ uint32_t sseg[2048];
  __shared__ uint16_t sidx[4096];
  __shared__ int scnt;
  for (int i = threadIdx.x; i < n; i += blockDim.x) sv[i] = data[i];
  __syncthreads();
  block_gnu_sort(sv, n, stmp, sseg, sq, sq + (2048 / 16 + 2), sidx, &scnt);
  for (int i = threadIdx.x; i < n; i += blockDim.x) data[i] = sv[i];
}

#endif  // ORBX_DEBUG_ABI

// ------------------------------------------------------------------------------------------------
// K3b: output slots (src/ORBextractor.cc:1122,1143-1164)
// ------------------------------------------------------------------------------------------------
template <bool GS>
__device__ __forceinline__ void assemble_main(const DeviceGeom* __restrict__ g, const uint32_t* __restrict__ lvl_kp,
                                              const int32_t* __restrict__ lvl_n, uint2* __restrict__ kp_list,
                                              int32_t* __restrict__ counts, int lap0, int lap1, unsigned long long* scan,
                                              unsigned long long* wt, int* loff, int* s_overflow, const int frame,
                                              int32_t* __restrict__ mirror_counts = nullptr) {
  const int t = threadIdx.x, T = blockDim.x;
  if (t == 0) {
    int acc = 0, ovf = 0;
    for (int l = 0; l < g->nlevels; l++) {
      const int nl = lvl_n[frame * g->nlevels + l];   // negative = the quadtree overflowed this level's capacity
      ovf |= nl < 0;
      loff[l] = acc;
      acc += max(nl, 0);
    }
    loff[g->nlevels] = acc;
    *s_overflow = ovf;
  }
  __syncthreads();
  const int total = min(loff[g->nlevels], g->out_cap);
  const uint32_t* kp = lvl_kp + (long long)frame * g->kp_total;
  for (int i = t; i < total; i += T) {
    int l = 0;
    while (i >= loff[l + 1]) l++;
    const uint32_t p = kp[g->lv[l].kp_off + (i - loff[l])];
    float x = (float)pt_x(p);
    if (l != 0) x = __fmul_rn(x, g->lv[l].scale);
    scan[i] = (x >= (float)lap0 && x <= (float)lap1) ? 1ull : 0ull;
  }
  __syncthreads();
  const int nst = (int)block_excl_scan(scan, total, wt);
  for (int i = t; i < total; i += T) {
    int l = 0;
    while (i >= loff[l + 1]) l++;
    const uint32_t p = kp[g->lv[l].kp_off + (i - loff[l])];
    float x = (float)pt_x(p);
    if (l != 0) x = __fmul_rn(x, g->lv[l].scale);
    const bool st = x >= (float)lap0 && x <= (float)lap1;
    const int pre = (int)scan[i];
    const int slot = st ? total - 1 - pre : i - pre;
    kp_list[(long long)frame * g->out_cap + i] = make_uint2(p, (uint32_t)l | ((uint32_t)slot << 8));
  }
  if (t == 0) {
    counts[frame * 2] = *s_overflow ? -1 : total; counts[frame * 2 + 1] = total - nst;
    if (mirror_counts) { mirror_counts[frame * 2] = *s_overflow ? -1 : total; mirror_counts[frame * 2 + 1] = total - nst; }
  }
}

// gscan: per-frame scan scratch in HBM for capacities whose scan array does not fit the LDS (nullptr: LDS)
__global__ __launch_bounds__(256) void k_assemble(const DeviceGeom* __restrict__ g, const uint32_t* __restrict__ lvl_kp,
                                                  const int32_t* __restrict__ lvl_n, uint2* __restrict__ kp_list,
                                                  int32_t* __restrict__ counts, int lap0, int lap1,
                                                  unsigned long long* __restrict__ gscan) {
  extern __shared__ __align__(16) uint8_t smem[];
  __shared__ unsigned long long wt[8];
  __shared__ int loff[kMaxLevels + 1];
  __shared__ int s_overflow;
  if (gscan == nullptr) assemble_main<false>(g, lvl_kp, lvl_n, kp_list, counts, lap0, lap1, (unsigned long long*)smem, wt, loff, &s_overflow, (int)blockIdx.x);
  else assemble_main<true>(g, lvl_kp, lvl_n, kp_list, counts, lap0, lap1, gscan + (long long)blockIdx.x * g->out_cap, wt, loff, &s_overflow, (int)blockIdx.x);
}

// k_quadtree of ALL levels of a small batch with k_assemble as its tail: the last level-workgroup of a frame to finish (a counter per
// frame, zero between launches) assembles the frame.  One launch less in the single-frame graph, where every launch costs about 5 us
// whatever it does.  The other workgroups' keypoint lists reach the assembling one through an agent-scope release / acquire pair —
// once per workgroup (eight per frame), which is affordable.  LDS-resident node arrays only (the host takes the separate launches
// otherwise); `smem` is free again when quadtree_main returns and holds the scan array of the assembly.
struct QtaArgs {
  const DeviceGeom* g; const CellGeom* cells; const uint32_t* cand; const int32_t* cell_cnt; uint32_t* pts; uint32_t* lvl_kp; int32_t* lvl_n;
  int node_cap, scan_cap, pts_cap;
  // the tail's own arguments: read from the kernel-argument segment AFTER the quadtree (see below)
  uint2* kp_list; int32_t* counts; int lap0, lap1; int* fin; int32_t* mirror_counts;
};
__global__ __launch_bounds__(512, 1) void k_quadtree_assemble(const QtaArgs a) {
  extern __shared__ __align__(16) uint8_t smem[];
  __shared__ unsigned long long wt[16];
  __shared__ int sh_cnt[kMaxRoots];
  __shared__ int sh_jstar;
  __shared__ int loff[kMaxLevels + 1];
  __shared__ int s_overflow, s_last;
  uint32_t* lpts = (uint32_t*)(smem + (size_t)a.node_cap * (2 * sizeof(QNode) + 2 * 8 + sizeof(int4) + 4) + (size_t)a.scan_cap * 8);
  quadtree_main<false>(a.g, a.cells, a.cand, a.cell_cnt, a.pts, a.lvl_kp, a.lvl_n, a.node_cap, a.scan_cap, a.pts_cap, 0, smem, lpts, wt, sh_cnt, &sh_jstar);
  // The quadtree runs at the edge of the scalar register file: arguments that stay live across it for the tail's sake spill (36
  // bytes of private segment, and a kernel with a private segment pays a scratch set-up on every queue that first runs it).  The
  // tail therefore reads its arguments from the kernel-argument segment through a pointer the compiler cannot see through.
  const QtaArgs* ap = (const QtaArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(ap));
  const int frame = (int)blockIdx.y;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave's list entries have left before the workgroup reports
  __syncthreads();
  if (threadIdx.x == 0) {
    int* fin = ap->fin;
    const int done = __hip_atomic_fetch_add(fin + frame, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    s_last = done == (int)gridDim.x - 1;
    if (s_last) __hip_atomic_store(fin + frame, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
  }
  __syncthreads();
  if (!s_last) return;   // block-uniform
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  assemble_main<false>(ap->g, ap->lvl_kp, ap->lvl_n, ap->kp_list, ap->counts, ap->lap0, ap->lap1, (unsigned long long*)smem, wt, loff, &s_overflow, frame,
                       ap->mirror_counts);
}

// ------------------------------------------------------------------------------------------------
// K4a: cv::GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) of every pyramid level (src/ORBextractor.cc:1132-1133)
// in 8.8 fixed point.  The kernel is VALU-bound, so both passes run on packed dot products:
//   * 64x58 output tile per workgroup; raw tile of 64 rows x 72 bytes (3-px halo, origin at x0-4 so rows are dword
//     aligned; LDS pitch 96 keeps the two row pairs of a 32-lane group on disjoint banks), staged with dword loads;
//     reflect-101 is a row remap for y and a few patched bytes per row for x;
//   * horizontal pass: v_dot4_u32_u8 on byte-aligned windows (v_alignbyte), one lane = 2 rows x 4 px, results
//     stored as row-pair interleaved u16 (row 2p in the low half, row 2p+1 in the high half of a dword);
//   * vertical pass: v_dot2_u32_u16 on those pairs (4 instructions per pixel, exact 32-bit accumulation,
//     single rounding (acc + 2^15) >> 16), one lane = 2 rows x 4 px, one dword store per row.
// ------------------------------------------------------------------------------------------------
// hw: the 7 horizontal weights at the four byte alignments (hrow4); radd: rounding constant folded into the column pass (2^15 on the
// default path, 0 when `flags` asks for the general rounding below).
// flags == 0: the default arithmetic (OpenCV >= 4.5.1: weights sum to 256, (acc + 2^15) >> 16, no saturation needed).  Otherwise the
// arithmetic of the other OpenCV releases (include/orbx.h "gauss_kernel" / "gauss_round" / "gauss_tail"): bit 0 = set whenever the general
// path is taken (results saturate to 255: the 257 kernel reaches 256), bits 1-2 = rounding of the body columns (0 half up, 1 exact ties to
// even, 2 floor); tail_mask = V - 1: the last (w mod V) columns of every row round half up (a SIMD column pass's scalar tail), 0 = no tail
struct BlurConsts { uint32_t hw[10]; uint32_t we[4], wo[4]; uint32_t radd, flags, tail_mask; };

__device__ __forceinline__ int reflect101(int p, int n) {
  while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
  return p;
}

constexpr int kBT_W = 64, kBT_H = 58, kBT_RR = 64, kBT_RP = 96, kBT_RB = 72;  // raw tile: 64 rows, 72 bytes used of 96

template <int OFF>
__device__ __forceinline__ uint32_t bytes4(uint32_t d0, uint32_t d1, uint32_t d2) {
  // 4 bytes starting at byte OFF (0..7) of the 12-byte little-endian window d0,d1,d2
  if (OFF == 0) return d0;
  if (OFF < 4) return __builtin_amdgcn_alignbyte(d1, d0, OFF);
  if (OFF == 4) return d1;
  return __builtin_amdgcn_alignbyte(d2, d1, OFF - 4);
}

typedef unsigned short v2u16_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t udot2(uint32_t a, uint32_t b, uint32_t c) {
  v2u16_t va, vb;
  __builtin_memcpy(&va, &a, 4);
  __builtin_memcpy(&vb, &b, 4);
  return __builtin_amdgcn_udot2(va, vb, c, false);
}

// horizontal 7-tap sums of 4 adjacent pixels: output x (tile coords 4j..4j+3) reads raw bytes x+1 .. x+7 of the 12-byte window
// d0 d1 d2.  The window is not shifted to the weights (six v_alignbyte per row): the WEIGHTS come pre-shifted to the four alignments,
// zero where a dword holds no tap — ten v_dot4 per row and no byte shuffling (exact u32 sums, same result).
__device__ __forceinline__ void hrow4(const uint32_t* rw, const BlurConsts& bc, uint32_t (&a)[4]) {
  const uint32_t d0 = rw[0], d1 = rw[1], d2 = rw[2];
  a[0] = __builtin_amdgcn_udot4(d0, bc.hw[0], __builtin_amdgcn_udot4(d1, bc.hw[1], 0u, false), false);
  a[1] = __builtin_amdgcn_udot4(d0, bc.hw[2], __builtin_amdgcn_udot4(d1, bc.hw[3], __builtin_amdgcn_udot4(d2, bc.hw[4], 0u, false), false), false);
  a[2] = __builtin_amdgcn_udot4(d0, bc.hw[5], __builtin_amdgcn_udot4(d1, bc.hw[6], __builtin_amdgcn_udot4(d2, bc.hw[7], 0u, false), false), false);
  a[3] = __builtin_amdgcn_udot4(d1, bc.hw[8], __builtin_amdgcn_udot4(d2, bc.hw[9], 0u, false), false);
}

// one blur tile (logical item L: frame-major, then the tiles of all levels); 256 threads
__device__ __forceinline__ void blur_tile(const int L, const DeviceGeom* __restrict__ g, const uint8_t* __restrict__ imgs,
                                          long long img_row_stride, long long img_frame_stride,
                                          const uint8_t* __restrict__ pyr, long long pyr_frame_bytes,
                                          uint8_t* __restrict__ blur, long long blur_frame_bytes, const BlurConsts& bc) {
  __shared__ __align__(16) uint8_t raw[kBT_RR * kBT_RP];
  __shared__ __align__(16) uint32_t hp[(kBT_RR / 2) * kBT_W];
  const int t = threadIdx.x;
  const int frame = fast_div(L, g->m_btiles), bt = L - frame * g->btiles_total;
  int l = 0;
#pragma unroll
  for (int i = 1; i < kMaxLevels; i++) l += bt >= g->btile_begin_all[i] ? 1 : 0;  // one scalar load, no dependent chain
  const DeviceLevel& lv = g->lv[l];
  const int tile = bt - lv.btile_begin;
  const int ty = fast_div(tile, lv.m_btiles_x), tx = tile - ty * lv.btiles_x;
  const int x0 = tx * kBT_W, y0 = ty * kBT_H;
  const uint8_t* img;
  int pitch;  // < 2^23 (checked on the host)
  if (l == 0) { img = imgs + (long long)frame * img_frame_stride; pitch = (int)img_row_stride; }
  else { img = pyr + (long long)frame * pyr_frame_bytes + lv.plane_off; pitch = lv.pitch; }
  const int w = lv.w, h = lv.h;
  // raw[r][c] = level(reflect(y0-3+r), reflect(x0-4+c)), c = 0..71
  const bool al = ((pitch & 3) == 0) && ((((unsigned long long)img) & 3) == 0) && (w >= 8);
  if (al) {
    // dwords that start inside [0, w) are loaded (a row is readable up to its 4-byte rounded width), the others and the
    // reflected columns are patched below
    // thread = (column c of 18 dwords, row phase of 14): everything that depends on the column only is computed once,
    // the thread then walks down its column 14 rows at a time (252 of the 256 threads take part)
    if (y0 >= 3 && y0 - 3 + kBT_RR <= h) {   // block-uniform: an interior tile, every source row exists
      // LDS-DMA loads: thread = (row phase of 10, dword column of the 24-dword LDS row): the LDS dword index of (row rph + 10 k, column c) is
      // t + 240 k — lane-linear.  The six dwords per row that hold no pixel and the dwords left / right of the level are not loaded (never
      // read with a non-zero weight, or patched below).
      if (t < 10 * (kBT_RP / 4)) {
        const int rph = (int)(((uint32_t)t * 2731u) >> 16), c = t - rph * (kBT_RP / 4);  // t / 24
        const int sx = x0 - 4 + 4 * c;
        const bool inx = c < kBT_RB / 4 && sx >= 0 && sx < w;
        const uint8_t* colp = img + (int)(__mul24(y0 - 3 + rph, pitch) + sx);
        const int step = __mul24(10, pitch);
#pragma unroll
        for (int k = 0; k < (kBT_RR + 9) / 10; k++) {
          uint32_t* dst = (uint32_t*)raw + 240 * k + (t & ~63);   // wave-uniform; the hardware adds lane * 4
          if (inx && rph + 10 * k < kBT_RR)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(colp + k * step),
                                             (__attribute__((address_space(3))) void*)dst, 4, 0, 0);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (t < 14 * (kBT_RB / 4)) {
      const int rph = (int)(((uint32_t)t * 3641u) >> 16), c = t - rph * (kBT_RB / 4);  // t / 18
      const int sx = x0 - 4 + 4 * c;
      const bool inx = sx >= 0 && sx < w;
      uint32_t* dstp = (uint32_t*)raw + rph * (kBT_RP / 4) + c;
      if (y0 >= 3 && y0 - 3 + kBT_RR <= h) {   // block-uniform: every source row exists (all tiles but the first and last rows of tiles)
        const uint8_t* colp = img + (uint32_t)(__mul24(y0 - 3 + rph, pitch) + sx);
        const uint32_t step = (uint32_t)__mul24(14, pitch);
#pragma unroll
        for (int k = 0; k < (kBT_RR + 13) / 14; k++) {
          if (rph + 14 * k < kBT_RR) {
            uint32_t v = 0;
            if (inx) v = *(const uint32_t*)(colp + k * step);
            dstp[k * 14 * (kBT_RP / 4)] = v;
          }
        }
      } else {
#pragma unroll
        for (int k = 0; k < (kBT_RR + 13) / 14; k++) {
          const int r = rph + 14 * k;
          if (r < kBT_RR) {
            // reflect-101 row: one reflection covers every row a stored output reads (h >= 67); rows further below the
            // level (only in the last tile row, never stored) just need a valid address
            int sy = y0 - 3 + r;
            sy = sy < 0 ? -sy : (sy >= h ? 2 * h - 2 - sy : sy);
            sy = max(sy, 0);
            uint32_t v = 0;
            if (inx) v = *(const uint32_t*)(img + (uint32_t)(__mul24(sy, pitch) + sx));
            dstp[k * 14 * (kBT_RP / 4)] = v;
          }
        }
      }
    }
    const bool left = x0 == 0, right = x0 + kBT_W + 3 > w;  // block-uniform
    if (left || right) {
      __syncthreads();
      // reflect-101 columns: raw col 4+x for x in {-3..-1} <- x' = -x; for x in {w .. w+2} <- x' = 2w-2-x (w >= 8)
      for (int i = t; i < kBT_RR * 8; i += 256) {
        const int r = i >> 3, k = i & 7;
        if (k >= 6) continue;
        uint8_t* row = raw + r * kBT_RP;
        if (k < 3) { if (left) row[4 - (k + 1)] = row[4 + (k + 1)]; }
        else if (right) {
          const int x = w + (k - 3);            // level column to synthesise
          const int c = x - x0 + 4;
          if (c < kBT_RB) row[c] = row[2 * w - 2 - x - x0 + 4];
        }
      }
    }
  } else {
#pragma unroll 1  // cold path: keep it out of the register budget
    for (int i = t; i < kBT_RR * kBT_RB; i += 256) {
      const int r = i / kBT_RB, c = i - r * kBT_RB;
      raw[r * kBT_RP + c] = img[(long long)reflect101(y0 - 3 + r, h) * pitch + reflect101(x0 - 4 + c, w)];
    }
  }
  __syncthreads();
  // horizontal: item = (row pair rp, group j): rows 2rp, 2rp+1, output columns 4j..4j+3
  for (int i = t; i < (kBT_RR / 2) * (kBT_W / 4); i += 256) {
    const int rp = i >> 4, j = i & 15;
    uint32_t a[4], b[4];
    hrow4((const uint32_t*)(raw + (2 * rp) * kBT_RP) + j, bc, a);
    hrow4((const uint32_t*)(raw + (2 * rp + 1) * kBT_RP) + j, bc, b);
    uint4 o;
    o.x = a[0] | (b[0] << 16); o.y = a[1] | (b[1] << 16); o.z = a[2] | (b[2] << 16); o.w = a[3] | (b[3] << 16);
    *(uint4*)(hp + rp * kBT_W + 4 * j) = o;
  }
  __syncthreads();
  // vertical: item = (output row pair op, group j): tile rows 2op, 2op+1 read h rows 2op .. 2op+7 = pairs op .. op+3
  uint8_t* bl = blur + (long long)frame * blur_frame_bytes + lv.bplane_off;
  for (int i = t; i < (kBT_H / 2) * (kBT_W / 4); i += 256) {
    const int op = i >> 4, j = i & 15;
    const int y = y0 + 2 * op, x = x0 + 4 * j;
    if (y >= h || x >= w) continue;
    uint4 P[4];
#pragma unroll
    for (int q = 0; q < 4; q++) P[q] = *(const uint4*)(hp + (op + q) * kBT_W + 4 * j);
    uint32_t e[4], o[4];
    // the four dot2 chains of a column, started from `init` (the rounding constant: folded into the accumulation)
    auto column_sums = [&](const uint32_t (&init)[4]) {
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const uint32_t p0 = c == 0 ? P[0].x : c == 1 ? P[0].y : c == 2 ? P[0].z : P[0].w;
        const uint32_t p1 = c == 0 ? P[1].x : c == 1 ? P[1].y : c == 2 ? P[1].z : P[1].w;
        const uint32_t p2 = c == 0 ? P[2].x : c == 1 ? P[2].y : c == 2 ? P[2].z : P[2].w;
        const uint32_t p3 = c == 0 ? P[3].x : c == 1 ? P[3].y : c == 2 ? P[3].z : P[3].w;
        e[c] = udot2(p3, bc.we[3], udot2(p2, bc.we[2], udot2(p1, bc.we[1], udot2(p0, bc.we[0], init[c]))));
        o[c] = udot2(p3, bc.wo[3], udot2(p2, bc.wo[2], udot2(p1, bc.wo[1], udot2(p0, bc.wo[0], init[c]))));
      }
    };
    if (!bc.flags) {   // uniform (kernel argument): the default arithmetic, radd = 2^15 in a scalar register
      const uint32_t in4[4] = {bc.radd, bc.radd, bc.radd, bc.radd};
      column_sums(in4);
    } else {
      // The other OpenCV releases' arithmetic (round 5: k_blur7 0.219 -> 0.181 ms per 256 frames under "opencv-4.4", 0.170 by default; it had been a
      // per-pixel chain of selects — profiles/blur_variants_r5.txt).  The rounding
      // constant still rides in the accumulation — 2^15 (half up, and the base of ties-to-even), 0 (floor) — chosen per COLUMN only where a floor body
      // meets a half-up tail; an exact tie is "low half == 0 after the 2^15" and rounding it to even clears bit 16; saturation (the 257 kernel
      // reaches 256) is one min on the 32-bit sum; the pixel is byte 2, packed like the default path's.
      const int body = w - (w & (int)bc.tail_mask);   // columns [body, w) round half up (a SIMD column pass's scalar tail)
      const uint32_t mode = (bc.flags >> 1) & 3u;     // uniform: 0 half up, 1 exact ties to even, 2 floor
      uint32_t in4[4];
#pragma unroll
      for (int c = 0; c < 4; c++) in4[c] = (mode == 2u && x + c < body) ? 0u : 32768u;
      column_sums(in4);
      if (mode == 1u) {
#pragma unroll
        for (int c = 0; c < 4; c++) {
          if (x + c < body) {
            if ((e[c] & 0xffffu) == 0u) e[c] &= ~0x10000u;
            if ((o[c] & 0xffffu) == 0u) o[c] &= ~0x10000u;
          }
        }
      }
#pragma unroll
      for (int c = 0; c < 4; c++) { e[c] = min(e[c], 0x00ffffffu); o[c] = min(o[c], 0x00ffffffu); }
    }
    // byte 2 of each 32-bit sum is the rounded pixel ((acc + 2^15) >> 16 <= 255): three byte permutes pack four of them
    const uint32_t pe = __builtin_amdgcn_perm(__builtin_amdgcn_perm(e[3], e[2], 0x0c0c0602u), __builtin_amdgcn_perm(e[1], e[0], 0x0c0c0602u), 0x05040100u);
    const uint32_t po = __builtin_amdgcn_perm(__builtin_amdgcn_perm(o[3], o[2], 0x0c0c0602u), __builtin_amdgcn_perm(o[1], o[0], 0x0c0c0602u), 0x05040100u);
    const uint32_t o0 = (uint32_t)(__mul24(y, lv.pitch) + x);
    *(uint32_t*)(bl + o0) = pe;
    if (y + 1 < h) *(uint32_t*)(bl + o0 + (uint32_t)lv.pitch) = po;
  }
}

__global__ __launch_bounds__(256) void k_blur7(const DeviceGeom* __restrict__ g, const uint8_t* __restrict__ imgs,
                                               long long img_row_stride, long long img_frame_stride,
                                               const uint8_t* __restrict__ pyr, long long pyr_frame_bytes,
                                               uint8_t* __restrict__ blur, long long blur_frame_bytes, BlurConsts bc, int nitems) {
  const int L = xcd_logical_block(nitems);
  if (L < 0) return;  // block-uniform
  blur_tile(L, g, imgs, img_row_stride, img_frame_stride, pyr, pyr_frame_bytes, blur, blur_frame_bytes, bc);
}

// FAST cells and blur tiles of a SMALL batch (the single-frame operator() path) in one launch: both only read the finished pyramid,
// and inside the replayed graph every launch costs about 5 us of device time whatever it does.  Blocks [0, nfast) are cells (256
// threads each), the rest blur tiles; results are those of the two separate launches (the same bodies on the same items).
template <int PITCH, bool PK>
__global__ __launch_bounds__(256) void k_fast_blur(const DeviceGeom* __restrict__ g, const CellGeom* __restrict__ cells,
                                                   const uint8_t* __restrict__ imgs, long long img_row_stride, long long img_frame_stride,
                                                   const uint8_t* __restrict__ pyr, long long pyr_frame_bytes, uint32_t* __restrict__ cand,
                                                   int32_t* __restrict__ cell_cnt, int ini_th, int min_th, int tile_rows, int nfast, int ncells,
                                                   uint32_t m_ncells, int stage_dma, int list_cap, int nwords, uint8_t* __restrict__ blur,
                                                   long long blur_frame_bytes, BlurConsts bc) {
  extern __shared__ __align__(16) uint8_t smem[];
  const int b = (int)blockIdx.x;
  if (b < nfast)   // block-uniform
    fast_cell<256, PITCH, PK>(smem, b, g, cells, imgs, img_row_stride, img_frame_stride, pyr, pyr_frame_bytes, cand, cell_cnt, ini_th, min_th, tile_rows, 0,
                              ncells, m_ncells, stage_dma, list_cap, nwords);
  else
    blur_tile(b - nfast, g, imgs, img_row_stride, img_frame_stride, pyr, pyr_frame_bytes, blur, blur_frame_bytes, bc);
}

// ------------------------------------------------------------------------------------------------
// K4b: orientation + steered BRIEF.  A wave serves K keypoints one after the other (the cos/sin of all K are computed once,
// one per lane).  (1) One 8-byte record per keypoint (written by k_assemble) replaces the level search; (2) the 31x31
// circular patch of the UNBLURRED level is read once, as 31 rows x 10 aligned dwords straight into registers — the integer
// moments are v_dot4 sums with byte masks, reduced over the wave by DPP; (3) the rotated test pattern is a packed dword per
// lane, loaded before the angle is known; (4) the 37x37 window of the BLURRED level the 512 taps fall into is staged in
// the wave's LDS slice with 6 coalesced dword loads and the taps are LDS byte reads (LDSP; without it they are 8 scattered
// byte gathers per lane, which made the kernel texture-addresser bound: TA 80 %, 0.222 ms -> 0.198 ms with the window in
// LDS); one wave ballot = 8 descriptor bytes, one 8-byte store per lane 0..3.
// ------------------------------------------------------------------------------------------------
struct DescConsts { int umax[16]; };

// cv::fastAtan2, degrees; correctly rounded division.  FMA = false: separate IEEE mul / add (the repository's canonical
// form: OpenCV's baseline build).  FMA = true: the contractions a compiler makes when OpenCV's file is built with -mfma (its AVX2 dispatch
// copy, GCC / clang default -ffp-contract): the three Horner steps fused, and 90 - poly * c as one fnma ("atan_fma" option, include/orbx.h).
template <bool FMA>
__device__ __forceinline__ float fast_atan2_deg_t(float y, float x) {
  const float scale = (float)(180.0 / 3.14159265358979323846);
  const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale;
  const float p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
  const float ax = fabsf(x), ay = fabsf(y);
  const float eps = (float)2.2204460492503131e-16;
  float a, c, c2;
  if (ax >= ay) {
    c = __fdiv_rn(ay, __fadd_rn(ax, eps));
    c2 = __fmul_rn(c, c);
    if (FMA) a = __fmul_rn(__fmaf_rn(__fmaf_rn(__fmaf_rn(p7, c2, p5), c2, p3), c2, p1), c);
    else a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
  } else {
    c = __fdiv_rn(ax, __fadd_rn(ay, eps));
    c2 = __fmul_rn(c, c);
    if (FMA) a = __fmaf_rn(-__fmaf_rn(__fmaf_rn(__fmaf_rn(p7, c2, p5), c2, p3), c2, p1), c, 90.f);
    else a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
  }
  if (x < 0) a = __fsub_rn(180.f, a);
  if (y < 0) a = __fsub_rn(360.f, a);
  return a;
}
__device__ __forceinline__ float fast_atan2_deg(float y, float x, int fma = 0) {
  return fma ? fast_atan2_deg_t<true>(y, x) : fast_atan2_deg_t<false>(y, x);
}

// One rotated test point of the steered BRIEF (src/ORBextractor.cc:118-120): cvRound(x*b + y*a), cvRound(x*a - y*b).  fma == 0: separate
// IEEE operations (canonical); fma != 0 (uniform): the contraction GCC / clang make when the reference is built as its
// CMakeLists.txt asks (-O3 -march=native): the FIRST product is the fused one — fma(x, b, y*a), fma(x, a, -(y*b)) ("brief_fma" option).
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t __attribute__((aligned(1))) u32_una;   // a dword at any byte address (global loads take it in one instruction)
struct __attribute__((packed, aligned(1))) U4una { uint32_t x, y, z, w; };   // sixteen bytes at any byte address, one global_load_dwordx4
constexpr int kDescSlicePitch = 48;   // bytes per row of k_describe's LDS slices (three 16-byte LDS-DMA transfers)
constexpr uint32_t kRndBits = 0x4B400000u;   // 1.5 * 2^23 as a float: x + it leaves cvRound(x) in the low mantissa bits (|x| < 2^22), ties to even
// Both coordinates of one rotated test point at once (v_pk_mul_f32 / v_pk_add_f32: two IEEE operations per instruction, each rounded on its
// own): {x b + y a, x a - y b} with AB = {b, a}, ANB = {a, -b} (the negation is exact: x a + (y (-b)) is x a - y b bit for bit).  The results
// stay floats + the rounding constant: the caller's address arithmetic takes the integer out of the mantissa.
template <bool FMA>
__device__ __forceinline__ void rot_tap(float x, float y, f32x2 AB, f32x2 ANB, int& ryb, int& rxb) {
  const f32x2 X = {x, x}, Y = {y, y};
  const f32x2 yy = Y * ANB;
  f32x2 r;
  if (FMA) r = __builtin_elementwise_fma(X, AB, yy);
  else r = X * AB + yy;
  r = r + (f32x2){12582912.f, 12582912.f};
  ryb = __float_as_int(r.x); rxb = __float_as_int(r.y);
}
// the eight taps of one lane (four tests) read from `base` with row pitch `pitch` (LDS window or the blurred plane); the loads sit inside
// the caller's uniform branch on the variant, so the two arithmetic forms are never both evaluated.  ryb = kRndBits + ry: v_mul_i24 sees
// its low 24 bits, 0x400000 + ry, so ryb * pitch + rxb = ry * pitch + rx + (pitch << 22) + kRndBits (mod 2^32): one uniform correction.
template <bool FMA, int SP>   // SP > 0: an LDS slice of that row pitch; 0: the blurred plane, pitch at run time
__device__ __forceinline__ void brief_taps(const uint8_t* __restrict__ base, int ctr, int pitch, const char4 (&pat)[4], float a, float b,
                                           int (&t0)[4], int (&t1)[4]) {
  const f32x2 AB = {b, a}, ANB = {a, -b};
  const uint32_t cu = (uint32_t)ctr - (((uint32_t)(SP ? SP : pitch) << 22) + kRndBits);   // uniform; the sums below are exact mod 2^32
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const float x0 = (float)pat[q].x, y0 = (float)pat[q].y, x1 = (float)pat[q].z, y1 = (float)pat[q].w;
    int ry0, rx0, ry1, rx1;
    rot_tap<FMA>(x0, y0, AB, ANB, ry0, rx0);
    rot_tap<FMA>(x1, y1, AB, ANB, ry1, rx1);
    int o0 = __mul24(ry0, SP ? SP : pitch) + rx0, o1 = __mul24(ry1, SP ? SP : pitch) + rx1;   // one v_mad_i32_i24 each
    asm("" : "+v"(o0), "+v"(o1));   // kept apart from the uniform part: the compiler would re-associate the sum into three additions
    if (SP) {   // an LDS slice: 32-bit addresses, the uniform part is one scalar sum (opaque: the compiler would otherwise express the
      // slice's own base as this sum minus the constant in every other access of the caller)
      int cus = (int)cu;
      asm("" : "+s"(cus));
      const uint8_t* __restrict__ bu = base + cus;
      t0[q] = bu[o0]; t1[q] = bu[o1];
    } else {
      t0[q] = base[(uint32_t)o0 + cu];
      t1[q] = base[(uint32_t)o1 + cu];
    }
  }
}

// Sum over the wave by DPP (no LDS traffic); the total lands in lane 63 (rows 2,3 of the last step).
__device__ __forceinline__ int wave_sum_lane63(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);  // row_half_mirror
  v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);  // row_mirror: every lane = its row's sum
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast15 into rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast31 into rows 2, 3
  return v;
}

// K keypoints per wave: the cos/sin/atan2 evaluation (the largest VALU block, identical in all 64 lanes for one
// keypoint) is done once for K keypoints held in lanes 0..K-1.
// DIRECT (single frame, lapping area trivially all-in or all-out): no assembly pass — the wave finds its keypoints in the quadtree's
// per-level output itself (eight counts to scan) and the output slot is the closed form of src/ORBextractor.cc:1153-1162 for that case
// (everything in the lapping area: slot = N-1-i, return value 0; nothing in it: slot = i, return value N); the first workgroup writes
// the two counts.  direct_mode: 1 = all-in, 2 = all-out.
template <int K, bool LDSP = false, bool DIRECT = false>
__global__ __launch_bounds__(256) void k_describe(const DeviceGeom* __restrict__ g, const uint8_t* __restrict__ imgs,
                                                  long long img_row_stride, long long img_frame_stride,
                                                  const uint8_t* __restrict__ pyr, long long pyr_frame_bytes,
                                                  const uint8_t* __restrict__ blur, long long blur_frame_bytes,
                                                  const uint2* __restrict__ kp_list, const int32_t* __restrict__ counts,
                                                  orbx_keypoint* __restrict__ out_kps, uint8_t* __restrict__ out_desc,
                                                  DescConsts dc, int groups_per_frame, int nitems, uint32_t m_gpf,
                                                  orbx_keypoint* __restrict__ mirror_kps, uint8_t* __restrict__ mirror_desc,
                                                  const uint32_t* __restrict__ lvl_kp = nullptr, const int32_t* __restrict__ lvl_n = nullptr,
                                                  int direct_mode = 0, int32_t* __restrict__ counts_out = nullptr,
                                                  int32_t* __restrict__ mirror_counts = nullptr, int atan_fma = 0, int brief_fma = 0) {
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: the wave's LDS slice is an SGPR base
  const int L = xcd_logical_block(nitems);
  if (L < 0) return;
  const int frame = fast_div(L, m_gpf);
  const int g0 = ((L - frame * groups_per_frame) * 4 + w) * K;  // first keypoint (level-major index) of this wave
  int total;
  int loffr = 0;   // DIRECT: lane l < nlevels holds the first level-major index of level l, lane nlevels the total
  if constexpr (DIRECT) {
    int nl = 0;
    if (lane < g->nlevels) nl = lvl_n[frame * g->nlevels + lane];   // negative = the quadtree overflowed this level's capacity
    const bool ovf = __ballot(nl < 0) != 0ull;
    int inc = max(nl, 0);
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) { const int u = __shfl_up(inc, o); if (lane >= o) inc += u; }   // nlevels <= 16
    loffr = inc - max(nl, 0);
    total = min(__shfl(inc, g->nlevels - 1), g->out_cap);
    if (lane == g->nlevels) loffr = total;
    if (L - frame * groups_per_frame == 0 && w == 0 && lane == 0) {
      const int c0 = ovf ? -1 : total, c1 = direct_mode == 1 ? 0 : total;
      counts_out[frame * 2] = c0; counts_out[frame * 2 + 1] = c1;
      if (mirror_counts) { mirror_counts[frame * 2] = c0; mirror_counts[frame * 2 + 1] = c1; }
    }
  } else {
    total = counts[frame * 2];
  }
  // the block's first wave always has work (or the whole block is past the frame's keypoints): later waves without keypoints
  // still take part in the two workgroup barriers below
  if (((L - frame * groups_per_frame) * 4) * K >= total) return;  // block-uniform
  const int nk = max(0, min(K, total - g0));
  uint2 rec = make_uint2(0u, 0u);
  if constexpr (DIRECT) {
    const int i = g0 + lane;   // level-major index of this lane's keypoint
    int l = 0;
#pragma unroll 1
    for (int q = 1; q < g->nlevels; q++) l += i >= __shfl(loffr, q);   // all lanes take part in the lane reads
    const int first = __shfl(loffr, l);
    if (lane < nk) {
      const uint32_t p = lvl_kp[(long long)frame * g->kp_total + g->lv[l].kp_off + (i - first)];
      const int slot = direct_mode == 1 ? total - 1 - i : i;
      rec = make_uint2(p, (uint32_t)l | ((uint32_t)slot << 8));
    }
  } else {
    if (lane < nk) rec = kp_list[(long long)frame * g->out_cap + g0 + lane];
  }
  // rotated-pattern operands: independent of the keypoints, issued first
  char4 pat[4];
#pragma unroll
  for (int q = 0; q < 4; q++) pat[q] = ((const char4*)c_pattern)[q * 64 + lane];
  // umax[0..15] (each <= 15) packed 4 bits apiece
  unsigned long long umpk = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) umpk |= (unsigned long long)(dc.umax[i] & 15) << (4 * i);
  // ---- intensity centroid of every keypoint (src/ORBextractor.cc:76-103): exact int32 moments, any summation order.  The patch is read as
  // 32 rows x 2 x 16 bytes from column kx - 16 — ONE unaligned 16-byte load per lane (the kernel is bound by its vector-memory instructions), so the window's phase never enters and a lane's byte weights are constants
  // of the wave.  pixel ^ 0x80 is I - 128 as a signed byte; the weights u (v) on the bytes inside the circle are signed bytes, 0 outside; the
  // patch is symmetric (sum u = sum v = 0 over it), so sum u (I - 128) IS m10: two v_dot4_i32_i8 per dword and nothing else.
  int wu[4], wv[4];
  {
    const int r = lane >> 1, v = r - kHalfPatch;   // one 16-byte load per lane: row r of 32, columns kx - 16 + 16 (lane & 1) .. + 15
    const int um = r <= 2 * kHalfPatch ? (int)((umpk >> (4 * (v < 0 ? -v : v))) & 15ull) : -1;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int u0 = 16 * (lane & 1) - 16 + 4 * i;
      uint32_t a = 0, b = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int u = u0 + j;
        if ((u < 0 ? -u : u) <= um) { a |= (uint32_t)(u & 0xff) << (8 * j); b |= (uint32_t)(v & 0xff) << (8 * j); }
      }
      wu[i] = (int)a; wv[i] = (int)b;
    }
  }
  // Every load of the wave's K keypoints is issued before anything waits: (LDSP) the 37 x 37 windows of the blurred level into the wave's K LDS slices by LDS-DMA — they depend on the position only —
  // then the orientation patches into registers: loads return in order, so the one wait for the patches covers the windows too.  The kernel is bound by load latency, not by instructions: one keypoint at a time (load, wait,
  // compute) left a wave with one patch in flight.  Slots k >= nk repeat the wave's last keypoint (straight-line code, exact wait counts).
  __shared__ __align__(16) uint8_t s_patch[LDSP ? 4 * K : 1][37 * kDescSlicePitch + 16];   // 111 transfers of 16 bytes; slices stay 16-byte aligned
  int my_m01 = 0, my_m10 = 0;
  __builtin_amdgcn_sched_barrier(0);   // the loads below go out together, behind the constants above and before the first use
  if (nk > 0) {   // wave-uniform
    if constexpr (LDSP) {
      // The window starts at the dword below kx - 18; a slice row is 48 bytes = three 16-byte LDS-DMA transfers (gfx950's
      // global_load_lds_dwordx4), 37 rows = 111 transfers = TWO instructions of the wave (seven with dword transfers): transfer s = lane + 64 j
      // is row s / 3, bytes 16 (s % 3) .. + 15, and lands at byte 16 s of the slice — lane-linear, as the instruction writes.  Up to 30 bytes
      // past the window's last column are read along: inside the plane's pitch or the next row (the window's last row is not the plane's).
#pragma unroll
      for (int k = 0; k < K; k++) {
        const int kk = min(k, nk - 1);
        const uint32_t p = __builtin_amdgcn_readlane(rec.x, kk);
        const int l = (int)(__builtin_amdgcn_readlane(rec.y, kk) & 0xffu);
        const DeviceLevel& lv = g->lv[l];
        const int kx = pt_x(p), ky = pt_y(p), bp = lv.pitch;
        const uint8_t* bplane = blur + (long long)frame * blur_frame_bytes + lv.bplane_off;  // uniform
        uint8_t* sp = s_patch[w * K + k];
        const uint32_t wbase = (uint32_t)(__mul24(ky - 18, bp) + (kx - 18 - ((kx - 18) & 3)));
#pragma unroll
        for (int j = 0; j < 2; j++) {
          const int sidx = lane + 64 * j, row = (int)(((uint32_t)sidx * 21846u) >> 16), c16 = sidx - 3 * row;   // sidx / 3
          if (sidx < 37 * 3)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bplane + (wbase + (uint32_t)(__mul24(row, bp) + 16 * c16))),
                                             (__attribute__((address_space(3))) void*)(sp + 1024 * j), 16, 0, 0);
        }
      }
    }
    U4una dwm[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
      const int kk = min(k, nk - 1);
      const uint32_t p = __builtin_amdgcn_readlane(rec.x, kk);
      const int l = (int)(__builtin_amdgcn_readlane(rec.y, kk) & 0xffu);
      const DeviceLevel& lv = g->lv[l];
      const int kx = pt_x(p), ky = pt_y(p);
      const uint8_t* img;
      int pitch;  // < 2^23 (checked on the host)
      if (l == 0) { img = imgs + (long long)frame * img_frame_stride; pitch = (int)img_row_stride; }
      else { img = pyr + (long long)frame * pyr_frame_bytes + lv.plane_off; pitch = lv.pitch; }
      // rows ky-15 .. ky+16, columns kx-16 .. kx+15: inside the plane (keypoints keep 19 px from every border); row ky+16 and column kx-16
      // carry zero weights, and so does every byte outside the circle
      const uint8_t* win = img + (uint32_t)(__mul24(ky - kHalfPatch, pitch) + (kx - 16));
      dwm[k] = *(const U4una*)(win + (__umul24((uint32_t)(lane >> 1), (uint32_t)pitch) + 16u * (uint32_t)(lane & 1)));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < K; k++) {
      int m10 = 0, m01 = 0;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int px = (int)((i == 0 ? dwm[k].x : i == 1 ? dwm[k].y : i == 2 ? dwm[k].z : dwm[k].w) ^ 0x80808080u);
        m10 = __builtin_amdgcn_sdot4(px, wu[i], m10, false);
        m01 = __builtin_amdgcn_sdot4(px, wv[i], m01, false);
      }
      m10 = __builtin_amdgcn_readlane(wave_sum_lane63(m10), 63);
      m01 = __builtin_amdgcn_readlane(wave_sum_lane63(m01), 63);
      if (lane == k) { my_m10 = m10; my_m01 = m01; }
    }
  }
  // ---- angle, cos, sin: ONE pass of the (long, glibc-exact) trig code per workgroup — wave 0 computes them for the 4 K
  // keypoints of all four waves, one per lane, instead of every wave running the pass for its own K lanes
  __shared__ int s_mom[4 * K][2];
  __shared__ float s_trig[4 * K][3];
  if (lane < nk) { s_mom[w * K + lane][0] = my_m01; s_mom[w * K + lane][1] = my_m10; }
  __syncthreads();
  if (w == 0 && lane < 4 * K) {   // entries of waves without keypoints hold garbage; nobody reads their results
    const float ang = fast_atan2_deg((float)s_mom[lane][0], (float)s_mom[lane][1], atan_fma);
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    const float rad = __fmul_rn(ang, factorPI);
    s_trig[lane][0] = ang; s_trig[lane][1] = orbx_glibc::cosf_exact(rad); s_trig[lane][2] = orbx_glibc::sinf_exact(rad);
  }
  __syncthreads();
  float my_a = 0.f, my_b = 0.f;
  if (lane < nk) {
    my_a = s_trig[w * K + lane][1]; my_b = s_trig[w * K + lane][2];
    // the keypoint record (src/ORBextractor.cc:1128-1140), one keypoint per lane
    const int l = (int)(rec.y & 0xffu), slot = (int)(rec.y >> 8);
    const DeviceLevel& lv = g->lv[l];
    orbx_keypoint kp;
    float fx = (float)pt_x(rec.x), fy = (float)pt_y(rec.x);
    if (l != 0) { fx = __fmul_rn(fx, lv.scale); fy = __fmul_rn(fy, lv.scale); }
    kp.x = fx; kp.y = fy; kp.size = (float)lv.scaled_patch; kp.angle = s_trig[w * K + lane][0]; kp.response = (float)pt_s(rec.x);
    kp.octave = l; kp.class_id = -1;
    out_kps[(long long)frame * g->out_cap + slot] = kp;
    if (mirror_kps) mirror_kps[(long long)frame * g->out_cap + slot] = kp;
  }
  // ---- steered BRIEF (src/ORBextractor.cc:107-146) on the blurred level
  if constexpr (LDSP) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the windows have landed (long ago)
    wave_lds_sync();
  }
#pragma unroll 1
  for (int k = 0; k < nk; k++) {
    const uint32_t p = __builtin_amdgcn_readlane(rec.x, k);
    const uint32_t ry = __builtin_amdgcn_readlane(rec.y, k);
    const int l = (int)(ry & 0xffu), slot = (int)(ry >> 8);
    const float a = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_a), k));
    const float b = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_b), k));
    const DeviceLevel& lv = g->lv[l];
    const int kx = pt_x(p), ky = pt_y(p);
    const uint8_t* bplane = blur + (long long)frame * blur_frame_bytes + lv.bplane_off;  // uniform
    const int bp = lv.pitch;
    const int ctr = __mul24(ky, bp) + kx;  // taps stay >= 1 px inside the plane: offsets are non-negative
    int t0[4], t1[4];
    if constexpr (LDSP) {
      // the 512 taps from the wave's LDS slice of this keypoint (staged above) instead of 8 scattered byte gathers per lane
      const uint8_t* sp = s_patch[w * K + k];
      const int lctr = 18 * kDescSlicePitch + 18 + ((kx - 18) & 3);
      if (brief_fma) brief_taps<true, kDescSlicePitch>(sp, lctr, kDescSlicePitch, pat, a, b, t0, t1);   // uniform (kernel argument)
      else brief_taps<false, kDescSlicePitch>(sp, lctr, kDescSlicePitch, pat, a, b, t0, t1);
    } else {
      if (brief_fma) brief_taps<true, 0>(bplane, ctr, bp, pat, a, b, t0, t1);
      else brief_taps<false, 0>(bplane, ctr, bp, pat, a, b, t0, t1);
    }
    // test 64 q + lane is bit `lane` of the row's q-th 64-bit word: the four ballots land in lanes 0..7 as the row's eight dwords
    // (v_writelane has no builtin in this compiler.  One statement for the eight of them, behind five wait states: the compiler's hazard
    // recogniser does not look into it, and a v_writelane issued right behind the v_cmp that wrote its SGPR operand reads the old value —
    // measured, the ISA manual lists the wait only for the lane-select operand.)
    const unsigned long long b0 = __ballot(t0[0] < t1[0]), b1 = __ballot(t0[1] < t1[1]), b2 = __ballot(t0[2] < t1[2]), b3 = __ballot(t0[3] < t1[3]);
    uint32_t word = 0;
    asm("s_nop 4\n\tv_writelane_b32 %0, %1, 0\n\tv_writelane_b32 %0, %2, 1\n\tv_writelane_b32 %0, %3, 2\n\tv_writelane_b32 %0, %4, 3\n\t"
        "v_writelane_b32 %0, %5, 4\n\tv_writelane_b32 %0, %6, 5\n\tv_writelane_b32 %0, %7, 6\n\tv_writelane_b32 %0, %8, 7"
        : "+v"(word)
        : "s"((uint32_t)b0), "s"((uint32_t)(b0 >> 32)), "s"((uint32_t)b1), "s"((uint32_t)(b1 >> 32)), "s"((uint32_t)b2), "s"((uint32_t)(b2 >> 32)),
          "s"((uint32_t)b3), "s"((uint32_t)(b3 >> 32)));
    if (lane < 8) *(uint32_t*)(out_desc + ((long long)frame * g->out_cap + slot) * 32 + lane * 4) = word;
    // the single-frame graph: the same rows straight into the caller-visible pinned block (no download node; the HBM copy stays for the
    // searches that take the rows from there)
    if (mirror_desc && lane < 8) *(uint32_t*)(mirror_desc + ((long long)frame * g->out_cap + slot) * 32 + lane * 4) = word;
  }
}

// ------------------------------------------------------------------------------------------------
// K4, batches under the default blur arithmetic: the 7x7 Gaussian INSIDE the descriptor kernel.  The blurred levels have one reader — the 512
// taps around each keypoint — and k_describe is bound by its loads, k_blur7 by its instructions: a wave that stages the RAW 43 x 48 window of a
// keypoint (rows ky-21 .. ky+21, columns kx-24 .. kx+23) holds the orientation patch in it already, blurs the 37 x 37 window it needs with the
// arithmetic of blur_tile (hrow4 / v_dot2 column sums, the same constants: the bytes are those k_blur7 writes), and never touches a blurred
// plane: no k_blur7 launch, no 2 x 1 MB per frame of blurred levels through HBM, 2.0 KB of loads per keypoint instead of 2.8.
// Layout of a slice: raw[r][c] = level(reflect(ky - 21 + r), kx - 24 + c), pitch 48.  Row pairs of horizontal sums: hp[rp][o], o = output
// column, centre raw column o + 4 (x = kx - 20 + o).  Blurred bytes: B[b][o] over the raw slice, pitch 40: y = ky - 18 + b, x = kx - 20 + o.
// Windows at the left / right border: the transfers run a few bytes into the neighbouring row (never read with a weight) and the one or two
// reflect-101 columns are patched in the slice; only where that neighbouring row would be outside the plane (two corners) is the window
// staged byte by byte with reflect-101 in both directions.  Everything after the staging is the same.
// ------------------------------------------------------------------------------------------------
constexpr int kFB_ROWS = 43, kFB_PITCH = 48, kFB_X0 = 24, kFB_Y0 = 21, kFB_HP_ROWS = 22, kFB_HP_COLS = 40, kFB_BPITCH = 40;
#include "fb_items.inc"

template <int K, int R>
__global__ __launch_bounds__(256) void k_describe_blur(const DeviceGeom* __restrict__ g, const uint8_t* __restrict__ imgs,
                                                       long long img_row_stride, long long img_frame_stride,
                                                       const uint8_t* __restrict__ pyr, long long pyr_frame_bytes,
                                                       const uint2* __restrict__ kp_list, const int32_t* __restrict__ counts,
                                                       orbx_keypoint* __restrict__ out_kps, uint8_t* __restrict__ out_desc,
                                                       BlurConsts bc, int groups_per_frame, int nitems, uint32_t m_gpf,
                                                       int atan_fma, int brief_fma) {
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int L = xcd_logical_block(nitems);
  if (L < 0) return;
  const int frame = fast_div(L, m_gpf);
  const int total = counts[frame * 2];
  // a workgroup serves R consecutive groups of 4 K keypoints (R = 1 in the product: with a loop around the rounds the compiler keeps 138 registers
  // where the straight-line kernel keeps 81, and the residency lost costs more than the wave's 100 instructions of set-up: measured)
  const int first = (L - frame * groups_per_frame) * R * 4 * K;   // first keypoint (level-major index) of this workgroup
  if (first >= total) return;  // block-uniform
  char4 pat[4];
#pragma unroll
  for (int q = 0; q < 4; q++) pat[q] = ((const char4*)c_pattern)[q * 64 + lane];
  // moment weights (fb_items.inc, as in k_describe): lane = (patch row r of 32, half); the patch's sixteen bytes sit at raw row 6 + r, column 8 + 16 half
  int wu[4], wv[4];
  {
    const uint4 a = ((const uint4*)c_fb_w)[2 * lane], b = ((const uint4*)c_fb_w)[2 * lane + 1];
    wu[0] = (int)a.x; wu[1] = (int)a.y; wu[2] = (int)a.z; wu[3] = (int)a.w;
    wv[0] = (int)b.x; wv[1] = (int)b.y; wv[2] = (int)b.z; wv[3] = (int)b.w;
  }
  __shared__ __align__(16) uint8_t s_raw[4 * K][kFB_ROWS * kFB_PITCH + 16];
  __shared__ __align__(16) uint32_t s_hp[4][kFB_HP_ROWS * kFB_HP_COLS];
  // the items of the two blur passes this lane works on (fb_items.inc: only what the rotated pattern can read — three wave-trips each), as byte
  // offsets into the slice / the row-pair buffer (a generated table): constants of the wave, so a trip's addressing is one addition
  int h_r0[3], h_r1[3], h_hp[3], v_hp[3], v_b[3];
  {
    uint32_t o[16];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint4 q = ((const uint4*)c_fb_off)[4 * lane + i];
      o[4 * i] = q.x; o[4 * i + 1] = q.y; o[4 * i + 2] = q.z; o[4 * i + 3] = q.w;
    }
#pragma unroll
    for (int t = 0; t < 3; t++) { h_r0[t] = (int)o[5 * t]; h_r1[t] = (int)o[5 * t + 1]; h_hp[t] = (int)o[5 * t + 2]; v_hp[t] = (int)o[5 * t + 3]; v_b[t] = (int)o[5 * t + 4]; }
  }
  __shared__ int s_mom[4 * K][2];
  __shared__ float s_trig[4 * K][3];
  uint32_t* hp = s_hp[w];
#pragma unroll 1
  for (int rnd = 0; rnd < R; rnd++) {
  const int gfirst = first + rnd * 4 * K;                       // this round's first keypoint of the workgroup
  if (gfirst >= total) break;  // block-uniform
  const int g0 = gfirst + w * K;                                 // first keypoint of this wave
  const int nk = max(0, min(K, total - g0));
  uint2 rec = make_uint2(0u, 0u);
  if (lane < nk) rec = kp_list[(long long)frame * g->out_cap + g0 + lane];
  if (rnd) __syncthreads();   // the previous round's readers of s_trig / the slices are done
  int my_m01 = 0, my_m10 = 0;
  __builtin_amdgcn_sched_barrier(0);
  if (nk > 0) {   // wave-uniform
    // ---- stage the K raw windows: 129 sixteen-byte LDS-DMA transfers each (transfer s = lane + 64 j: row s / 3, bytes 16 (s % 3) ..)
    bool fast[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
      const int kk = min(k, nk - 1);
      const uint32_t p = __builtin_amdgcn_readlane(rec.x, kk);
      const int l = (int)(__builtin_amdgcn_readlane(rec.y, kk) & 0xffu);
      const DeviceLevel& lv = g->lv[l];
      const int kx = pt_x(p), ky = pt_y(p), lw = lv.w, lh = lv.h;
      const uint8_t* img;
      int pitch;  // < 2^23 (checked on the host)
      if (l == 0) { img = imgs + (long long)frame * img_frame_stride; pitch = (int)img_row_stride; }
      else { img = pyr + (long long)frame * pyr_frame_bytes + lv.plane_off; pitch = lv.pitch; }
      // uniform.  The transfers may start up to 5 bytes before the row (kx >= 19) and end up to 4 bytes past its pixels: the neighbouring row of
      // the same plane — unless that row is the plane's first (last) one: only those two corners are staged byte by byte.  Columns that need
      // reflect-101 (kx < 21, kx > lw - 22) are patched in the slice below; the other out-of-row bytes are never read with a weight.
      fast[k] = !((kx < kFB_X0 && ky <= kFB_Y0) || (kx + (kFB_PITCH - kFB_X0) > lw && ky + kFB_Y0 >= lh - 1));
      if (fast[k]) {
#pragma unroll
        for (int j = 0; j < 3; j++) {
          const int sidx = min(lane + 64 * j, kFB_ROWS * 3 - 1), row = (int)(((uint32_t)sidx * 21846u) >> 16), c16 = sidx - 3 * row;   // sidx / 3
          int y = ky - kFB_Y0 + row;
          y = y < 0 ? -y : (y >= lh ? 2 * lh - 2 - y : y);   // reflect-101 (one reflection: keypoints keep 19 rows from the border)
          if (lane + 64 * j < kFB_ROWS * 3)   // LDS-DMA takes a source at any byte address: the slice needs no phase
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(img + (uint32_t)(__mul24(y, pitch) + (kx - kFB_X0) + 16 * c16)),
                                             (__attribute__((address_space(3))) void*)(s_raw[w * K + k] + 1024 * j), 16, 0, 0);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < K; k++) {
      uint8_t* sp = s_raw[w * K + k];
      if (!fast[k]) {
        const int kk = min(k, nk - 1);
        const uint32_t p = __builtin_amdgcn_readlane(rec.x, kk);
        const int l = (int)(__builtin_amdgcn_readlane(rec.y, kk) & 0xffu);
        const DeviceLevel& lv = g->lv[l];
        const int kx = pt_x(p), ky = pt_y(p), lw = lv.w, lh = lv.h;
        const uint8_t* img;
        int pitch;
        if (l == 0) { img = imgs + (long long)frame * img_frame_stride; pitch = (int)img_row_stride; }
        else { img = pyr + (long long)frame * pyr_frame_bytes + lv.plane_off; pitch = lv.pitch; }
#pragma unroll 1
        for (int i = lane; i < kFB_ROWS * kFB_PITCH; i += 64) {
          const int row = (int)(((uint32_t)i * 1366u) >> 16), c = i - kFB_PITCH * row;   // i / 48 for i < 2064
          int y = ky - kFB_Y0 + row, x = kx - kFB_X0 + c;
          y = y < 0 ? -y : (y >= lh ? 2 * lh - 2 - y : y);
          x = x < 0 ? -x : (x >= lw ? 2 * lw - 2 - x : x);
          sp[i] = img[(uint32_t)(__mul24(y, pitch) + x)];
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wave_lds_sync();
    // reflect-101 columns of the windows that reach over the left / right border (two pixels at most: kx >= 19, kx <= lw - 20); lane = row.
    // The blur reads them after the workgroup barriers below; the moments' patch (columns 8 .. 39) does not contain them.
#pragma unroll
    for (int k = 0; k < K; k++) {
      const int kk = min(k, nk - 1);
      const uint32_t p = __builtin_amdgcn_readlane(rec.x, kk);
      const int l = (int)(__builtin_amdgcn_readlane(rec.y, kk) & 0xffu);
      const int kx = pt_x(p), lw = g->lv[l].w;
      uint8_t* sp = s_raw[w * K + k];
      if (fast[k] && (kx < kFB_Y0 || kx + kFB_Y0 >= lw) && lane < kFB_ROWS) {   // (the first condition is uniform)
        uint8_t* row = sp + lane * kFB_PITCH;
        if (kx < kFB_Y0) {   // x = kx - 24 + c < 0 for c < 24 - kx; needed from c = 3: the mirror of x is -x, column 2 (24 - kx) - c
          const int c0 = kFB_X0 - kx;
          for (int c = 3; c < c0; c++) row[c] = row[2 * c0 - c];
        } else {             // x >= lw for c > lw - 1 - kx + 24; needed up to c = 45: the mirror of x is 2 lw - 2 - x
          const int c1 = lw - 1 - kx + kFB_X0;   // the column of x = lw - 1
          for (int c = c1 + 1; c <= 45; c++) row[c] = row[2 * c1 - c];
        }
      }
    }
    // ---- moments from the raw slices (src/ORBextractor.cc:76-103)
#pragma unroll
    for (int k = 0; k < K; k++) {
      const uint8_t* sp = s_raw[w * K + k];
      const uint2 a = *(const uint2*)(sp + (6 + (lane >> 1)) * kFB_PITCH + 8 + 16 * (lane & 1));
      const uint2 b = *(const uint2*)(sp + (6 + (lane >> 1)) * kFB_PITCH + 16 + 16 * (lane & 1));
      const uint32_t dwv[4] = {a.x, a.y, b.x, b.y};
      int m10 = 0, m01 = 0;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int px = (int)(dwv[i] ^ 0x80808080u);
        m10 = __builtin_amdgcn_sdot4(px, wu[i], m10, false);
        m01 = __builtin_amdgcn_sdot4(px, wv[i], m01, false);
      }
      m10 = __builtin_amdgcn_readlane(wave_sum_lane63(m10), 63);
      m01 = __builtin_amdgcn_readlane(wave_sum_lane63(m01), 63);
      if (lane == k) { my_m10 = m10; my_m01 = m01; }
    }
  }
  if (lane < nk) { s_mom[w * K + lane][0] = my_m01; s_mom[w * K + lane][1] = my_m10; }
  __syncthreads();
  if (w == 0 && lane < 4 * K) {
    const float ang = fast_atan2_deg((float)s_mom[lane][0], (float)s_mom[lane][1], atan_fma);
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    const float rad = __fmul_rn(ang, factorPI);
    s_trig[lane][0] = ang; s_trig[lane][1] = orbx_glibc::cosf_exact(rad); s_trig[lane][2] = orbx_glibc::sinf_exact(rad);
  }
  __syncthreads();
  float my_a = 0.f, my_b = 0.f;
  if (lane < nk) {
    my_a = s_trig[w * K + lane][1]; my_b = s_trig[w * K + lane][2];
    const int l = (int)(rec.y & 0xffu), slot = (int)(rec.y >> 8);
    const DeviceLevel& lv = g->lv[l];
    orbx_keypoint kp;
    float fx = (float)pt_x(rec.x), fy = (float)pt_y(rec.x);
    if (l != 0) { fx = __fmul_rn(fx, lv.scale); fy = __fmul_rn(fy, lv.scale); }
    kp.x = fx; kp.y = fy; kp.size = (float)lv.scaled_patch; kp.angle = s_trig[w * K + lane][0]; kp.response = (float)pt_s(rec.x);
    kp.octave = l; kp.class_id = -1;
    out_kps[(long long)frame * g->out_cap + slot] = kp;
  }
#pragma unroll 1
  for (int k = 0; k < nk; k++) {
    const int slot = (int)(__builtin_amdgcn_readlane(rec.y, k) >> 8);
    const float a = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_a), k));
    const float b = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_b), k));
    uint8_t* sp = s_raw[w * K + k];
    // ---- horizontal sums, a row pair and four columns per item (blur_tile's form)
#pragma unroll
    for (int t = 0; t < 3; t++) {
      uint32_t ha[4], hb[4];
      hrow4((const uint32_t*)(sp + h_r0[t]), bc, ha);
      hrow4((const uint32_t*)(sp + h_r1[t]), bc, hb);
      uint4 o;   // the low halves of the two sums side by side: one v_perm each
      o.x = __builtin_amdgcn_perm(hb[0], ha[0], 0x05040100u); o.y = __builtin_amdgcn_perm(hb[1], ha[1], 0x05040100u);
      o.z = __builtin_amdgcn_perm(hb[2], ha[2], 0x05040100u); o.w = __builtin_amdgcn_perm(hb[3], ha[3], 0x05040100u);
      if (t < 2 || lane + 128 < kFB_NH) *(uint4*)((uint8_t*)hp + h_hp[t]) = o;
    }
    wave_lds_sync();
    // ---- column sums -> blurred bytes over the raw slice (two rows x four columns per item)
#pragma unroll
    for (int t = 0; t < 3; t++) {
      uint4 P[4];
#pragma unroll
      for (int q = 0; q < 4; q++) P[q] = *(const uint4*)((const uint8_t*)hp + v_hp[t] + q * (kFB_HP_COLS * 4));
      uint32_t e[4], o[4];
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const uint32_t p0 = c == 0 ? P[0].x : c == 1 ? P[0].y : c == 2 ? P[0].z : P[0].w;
        const uint32_t p1 = c == 0 ? P[1].x : c == 1 ? P[1].y : c == 2 ? P[1].z : P[1].w;
        const uint32_t p2 = c == 0 ? P[2].x : c == 1 ? P[2].y : c == 2 ? P[2].z : P[2].w;
        const uint32_t p3 = c == 0 ? P[3].x : c == 1 ? P[3].y : c == 2 ? P[3].z : P[3].w;
        e[c] = udot2(p3, bc.we[3], udot2(p2, bc.we[2], udot2(p1, bc.we[1], udot2(p0, bc.we[0], bc.radd))));
        o[c] = udot2(p3, bc.wo[3], udot2(p2, bc.wo[2], udot2(p1, bc.wo[1], udot2(p0, bc.wo[0], bc.radd))));
      }
      const uint32_t pe = __builtin_amdgcn_perm(__builtin_amdgcn_perm(e[3], e[2], 0x0c0c0602u), __builtin_amdgcn_perm(e[1], e[0], 0x0c0c0602u), 0x05040100u);
      const uint32_t po = __builtin_amdgcn_perm(__builtin_amdgcn_perm(o[3], o[2], 0x0c0c0602u), __builtin_amdgcn_perm(o[1], o[0], 0x0c0c0602u), 0x05040100u);
      if (t < 2 || lane + 128 < kFB_NV) {
        *(uint32_t*)(sp + v_b[t]) = pe;
        *(uint32_t*)(sp + v_b[t] + kFB_BPITCH) = po;
      }
    }
    wave_lds_sync();
    // ---- steered BRIEF on the blurred bytes: (ky + ry, kx + rx) is B[ry + 18][rx + 20]
    int t0[4], t1[4];
    const int lctr = 18 * kFB_BPITCH + 20;
    // the pattern stays four packed registers across the kernel's loops: its 16 floats are converted per keypoint (the compiler would keep
    // them — as 32 registers of pairs — and the kernel lives on its residency)
    char4 patk[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      uint32_t pq;
      __builtin_memcpy(&pq, &pat[q], 4);
      asm volatile("" : "+v"(pq));
      __builtin_memcpy(&patk[q], &pq, 4);
    }
    if (brief_fma) brief_taps<true, kFB_BPITCH>(sp, lctr, kFB_BPITCH, patk, a, b, t0, t1);   // uniform (kernel argument)
    else brief_taps<false, kFB_BPITCH>(sp, lctr, kFB_BPITCH, patk, a, b, t0, t1);
    const unsigned long long b0 = __ballot(t0[0] < t1[0]), b1 = __ballot(t0[1] < t1[1]), b2 = __ballot(t0[2] < t1[2]), b3 = __ballot(t0[3] < t1[3]);
    uint32_t word = 0;
    asm("s_nop 4\n\tv_writelane_b32 %0, %1, 0\n\tv_writelane_b32 %0, %2, 1\n\tv_writelane_b32 %0, %3, 2\n\tv_writelane_b32 %0, %4, 3\n\t"
        "v_writelane_b32 %0, %5, 4\n\tv_writelane_b32 %0, %6, 5\n\tv_writelane_b32 %0, %7, 6\n\tv_writelane_b32 %0, %8, 7"
        : "+v"(word)
        : "s"((uint32_t)b0), "s"((uint32_t)(b0 >> 32)), "s"((uint32_t)b1), "s"((uint32_t)(b1 >> 32)), "s"((uint32_t)b2), "s"((uint32_t)(b2 >> 32)),
          "s"((uint32_t)b3), "s"((uint32_t)(b3 >> 32)));
    if (lane < 8) *(uint32_t*)(out_desc + ((long long)frame * g->out_cap + slot) * 32 + lane * 4) = word;
  }
  }   // rounds
}

#ifdef ORBX_DEBUG_ABI
// Debug/test kernel: the two float paths of K4 in isolation (fastAtan2, then glibc-exact cosf/sinf of
// angle*factorPI) so tests can sweep far more arguments than real frames produce.
__global__ void k_debug_trig(const float* __restrict__ y, const float* __restrict__ x, int n, int angle_is_input,
                             float* __restrict__ angle, float* __restrict__ a, float* __restrict__ b, int atan_fma) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float ang = angle_is_input ? y[i] : fast_atan2_deg(y[i], x[i], atan_fma);
  const float factorPI = (float)(3.14159265358979323846 / 180.f);
  const float r = __fmul_rn(ang, factorPI);
  angle[i] = ang;
  a[i] = orbx_glibc::cosf_exact(r);
  b[i] = orbx_glibc::sinf_exact(r);
}

// Exhaustive check of the device cos/sin path: an order-independent 64-bit digest of (cos, sin)(angle * factorPI) over
// the float bit patterns [first, first + count) — the oracle computes the same digest with the host glibc.
__global__ __launch_bounds__(256) void k_debug_trig_hash(uint32_t first, uint32_t count, unsigned long long* __restrict__ out) {
  const float factorPI = (float)(3.14159265358979323846 / 180.f);
  unsigned long long h = 0;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += (unsigned long long)gridDim.x * blockDim.x) {
    const uint32_t bits = first + (uint32_t)i;
    const float r = __fmul_rn(__uint_as_float(bits), factorPI);
    const uint32_t ab = __float_as_uint(orbx_glibc::cosf_exact(r)), bb = __float_as_uint(orbx_glibc::sinf_exact(r));
    h += ((unsigned long long)ab * 0x9E3779B97F4A7C15ull) ^ ((unsigned long long)bb * 0xC2B2AE3D27D4EB4Full + bits);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) h += __shfl_xor(h, o);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, h);
}

// The same for fastAtan2: digest over `count` pseudo-random integer moment pairs derived from the index by a fixed
// integer mix (identical on the oracle side).
__device__ __forceinline__ uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__global__ __launch_bounds__(256) void k_debug_atan_hash(uint32_t seed, uint32_t count, unsigned long long* __restrict__ out, int atan_fma) {
  unsigned long long h = 0;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += (unsigned long long)gridDim.x * blockDim.x) {
    const uint32_t a = mix32(seed + 2u * (uint32_t)i), b = mix32(seed + 2u * (uint32_t)i + 1u);
    const int m01 = (int)(a % 6000001u) - 3000000, m10 = (b & 15u) == 0 ? 0 : (int)(b % 6000001u) - 3000000;
    const uint32_t bits = __float_as_uint(fast_atan2_deg((float)m01, (float)m10, atan_fma));
    h += ((unsigned long long)bits * 0x9E3779B97F4A7C15ull) ^ (unsigned long long)(uint32_t)i;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) h += __shfl_xor(h, o);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, h);
}

// The same for the rotated test pattern of the steered BRIEF: digest of (ry, rx) of all 512 pattern points over the float bit patterns
// [first, first + count) of the keypoint angle (degrees) — the oracle computes the same digest on the host (orbo_brief_hash).
__global__ __launch_bounds__(256) void k_debug_brief_hash(uint32_t first, uint32_t count, unsigned long long* __restrict__ out, int brief_fma) {
  const float factorPI = (float)(3.14159265358979323846 / 180.f);
  unsigned long long h = 0;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += (unsigned long long)gridDim.x * blockDim.x) {
    const uint32_t bits = first + (uint32_t)i;
    const float r = __fmul_rn(__uint_as_float(bits), factorPI);
    const float a = orbx_glibc::cosf_exact(r), b = orbx_glibc::sinf_exact(r);
    unsigned long long hk = 0;
#pragma unroll 4
    for (int idx = 0; idx < 512; idx++) {
      const float px = (float)c_pattern[idx * 2], py = (float)c_pattern[idx * 2 + 1];
      int ry, rx;
      const f32x2 AB = {b, a}, ANB = {a, -b};
      if (brief_fma) rot_tap<true>(px, py, AB, ANB, ry, rx); else rot_tap<false>(px, py, AB, ANB, ry, rx);
      ry -= (int)kRndBits; rx -= (int)kRndBits;   // the product path leaves the constant in its address arithmetic
      hk += (unsigned long long)(uint32_t)(ry * 64 + rx + 4096) * (0x9E3779B97F4A7C15ull + 2ull * (unsigned long long)idx);
    }
    h += hk ^ (unsigned long long)bits;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) h += __shfl_xor(h, o);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, h);
}

#endif  // ORBX_DEBUG_ABI

// cv::resize's INTER_AREA shortcut for an exact 2 x 2 downscale (the rounded mean of each block): image ingestion only
// (src/System.cc:441-446); the pyramid's scale factor never hits it.
__global__ __launch_bounds__(256) void k_box2(const uint8_t* __restrict__ src, int src_pitch, uint8_t* __restrict__ dst, int dst_pitch, int dw,
                                              int dh) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= dw || y >= dh) return;
  const uint8_t* S = src + (size_t)(2 * y) * src_pitch + 2 * x;
  dst[(size_t)y * dst_pitch + x] = (uint8_t)(((int)S[0] + (int)S[1] + (int)S[src_pitch] + (int)S[src_pitch + 1] + 2) >> 2);
}

// Image ingestion: cv::cvtColor(COLOR_RGB2GRAY / BGR2GRAY / RGBA2GRAY / BGRA2GRAY) of
// src/Tracking.cc:1572-1585 fused behind the upload — OpenCV (>= 4.x) 8-bit path: 15-bit fixed point,
// gray = (R * 9798 + G * 19235 + B * 3735 + 2^14) >> 15.  Each lane converts 4 pixels and stores one dword.
__global__ __launch_bounds__(256) void k_color_to_gray(const uint8_t* __restrict__ src, long long src_frame_stride, int src_pitch,
                                                       int channels, int r_off, int b_off, uint8_t* __restrict__ dst,
                                                       long long dst_frame_stride, int dst_pitch, int rows, int cols) {
  const int x4 = 4 * (blockIdx.x * 64 + (threadIdx.x & 63)), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (y >= rows || x4 >= cols) return;
  const uint8_t* s = src + (long long)blockIdx.z * src_frame_stride + (long long)y * src_pitch;
  uint32_t packed = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    if (x4 + i < cols) {
      const uint8_t* p = s + (long long)(x4 + i) * channels;
      const int g = ((int)p[r_off] * 9798 + (int)p[1] * 19235 + (int)p[b_off] * 3735 + (1 << 14)) >> 15;
      packed |= (uint32_t)g << (8 * i);
    }
  }
  *(uint32_t*)(dst + (long long)blockIdx.z * dst_frame_stride + (long long)y * dst_pitch + x4) = packed;  // pitch is a multiple of 64
}

#ifdef ORBX_DEBUG_ABI
// Calibration kernel for the rocprofv3 FETCH_SIZE / WRITE_SIZE counters (MI355X_MICROARCH.md "HBM": the counters
// are only calibrated for wide streaming reads): copies n bytes with W bytes per lane per access (W = 1, 4, 16),
// i.e. a kernel whose HBM traffic is known exactly, in the access widths the extractor kernels use.
template <typename T>
__global__ __launch_bounds__(256) void k_calib_copy(const T* __restrict__ src, T* __restrict__ dst, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    dst[i] = src[i];
}
#endif  // ORBX_DEBUG_ABI

}  // namespace orbx
