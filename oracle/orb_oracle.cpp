// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of the reference ORB front-end (lturing/ORB_SLAM3_modified,
// src/ORBextractor.cc) together with the OpenCV 4.x primitives that file calls
// (cv::resize INTER_LINEAR 8u, copyMakeBorder, cv::FAST 9/16 + score + NMS,
// cv::GaussianBlur 7x7 sigma 2 fixed point, cv::fastAtan2, cvRound).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
// this library, and only as the checker.  The shipped product (liborbx.so) never
// links, imports or calls it.
//
// PARITY STATUS: pinned to the reference's own code EXCEPT for five OpenCV primitives.  The reference's
// src/ORBextractor.cc is compiled where it lies (oracle/ref_fragments.mk -> oracle/_ref/libref_orbextractor.so) against a
// container-only OpenCV shim (oracle/ref_shims/opencv2) and tests/test_ref_fragments.py requires this restatement to
// equal it bit for bit — constructor tables, level chain, cell loop with the iniTh -> minTh retry, quadtree with the host
// std::sort, IC_Angle, steered BRIEF, operator()'s output order — on every benchmark configuration, degenerate images
// and 120 random shapes / parameters.  What remains **unpinned** ("recalled", no OpenCV source or binary of any version
// exists in the container) are the five algorithms that file calls into OpenCV for — isolated here as orbo_fast, orbo_resize_linear,
// orbo_gaussian_blur7, orbo_fast_atan2 and the reflect-101 border (the shim forwards to exactly these).
//
// WHICH OPENCV each restatement is of (INTEGRATION.md section 6 has the full table):
//   cv::resize INTER_LINEAR 8u   the classic 11-bit fixed-point path, the same in every 3.x / 4.x release (IPP builds of <= 3.4 differ)
//   cv::FAST 9/16 + score + NMS  3.x / 4.x FAST_t<16> + cornerScore<16>
//   cv::copyMakeBorder           reflect-101 index map, every release
//   cv::GaussianBlur 8u 7x7      DEFAULT = OpenCV >= 4.5.1 (error-diffused 8.8 weights {18,34,48,56,...}).  The reference names 4.4.0 and
//                                3.2.0 (CMakeLists.txt:33, README.md:560), whose weights are {18,34,49,55,...} and whose column pass rounds
//                                differently: selectable, orbo_set_gauss_variant / orbo_set_gauss_tail below (product: "gauss_kernel",
//                                "gauss_round", "gauss_tail")
//   cv::fastAtan2                3.x / 4.x polynomial; DEFAULT = separate mul / add, the FMA-contracted form of OpenCV's AVX2 dispatch
//                                copy is orbo_set_atan_fma(1) (product: "atan_fma")
//   and of the reference itself: DEFAULT = src/ORBextractor.cc compiled without FP contraction; its -march=native form (the pattern
//                                rotation of :118-120 fused) is orbo_set_brief_fma(1) (product: "brief_fma"), pinned by the reference's
//                                file compiled with -O3 -mfma (oracle/_ref/libref_orbextractor_fma.so)
// A maintainer checks them against a REAL OpenCV with tools/validate_opencv.cpp; the product's adapter finds the variant of the OpenCV it
// is built with by itself (include/orbx_cv_calibrate.h).  tests/test_oracle_independent.py and tests/test_opencv_variants.py check the
// restatements against independent statements of their definitions.
//
// Float rules: compile with -ffp-contract=off (no FMA fusion),
// cosf/sinf are the host glibc's, division is IEEE.
//
// Each function cites the reference file:line it follows.

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <list>
#include <thread>
#include <utility>
#include <vector>

namespace {

constexpr int kPatchSize = 31;       // src/ORBextractor.cc:71
constexpr int kHalfPatch = 15;       // :72
constexpr int kEdgeThreshold = 19;   // :73

struct KeyPt {            // layout-identical to cv::KeyPoint (7 x 4 B)
  float x, y, size, angle, response;
  int32_t octave, class_id;
};

// rBRIEF pattern, src/ORBextractor.cc:149-407 (data table).
const int8_t kPattern[1024] = {
#include "orb_pattern.inc"
};

// cvRound(float/double): round half to even (lrint / cvtss2si semantics).
inline int cv_round(float v) { return (int)lrintf(v); }
inline int cv_round(double v) { return (int)lrint(v); }

// ---------------------------------------------------------------------------
// cv::resize(src, dst, dsize, 0, 0, INTER_LINEAR), CV_8UC1 — SURVEY §8(c)-R.
// Called by the reference at src/ORBextractor.cc:1183.
// ---------------------------------------------------------------------------
void resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh,
                      int dstride) {
  const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
  const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
  std::vector<int> xofs(dw), yofs(dh);
  std::vector<short> ialpha(dw * 2), ibeta(dh * 2);
  int xmax = dw;
  for (int dx = 0; dx < dw; dx++) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = (int)std::floor(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx + 1 >= sw) {
      xmax = std::min(xmax, dx);
      if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    }
    xofs[dx] = sx;
    float c0 = 1.f - fx, c1 = fx;
    int a0 = cv_round(c0 * 2048.f), a1 = cv_round(c1 * 2048.f);
    ialpha[dx * 2] = (short)std::min(std::max(a0, -32768), 32767);
    ialpha[dx * 2 + 1] = (short)std::min(std::max(a1, -32768), 32767);
  }
  for (int dy = 0; dy < dh; dy++) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = (int)std::floor(fy);
    fy -= sy;
    yofs[dy] = sy;
    float c0 = 1.f - fy, c1 = fy;
    int b0 = cv_round(c0 * 2048.f), b1 = cv_round(c1 * 2048.f);
    ibeta[dy * 2] = (short)std::min(std::max(b0, -32768), 32767);
    ibeta[dy * 2 + 1] = (short)std::min(std::max(b1, -32768), 32767);
  }
  std::vector<int> row0(dw), row1(dw);
  auto hresize = [&](const uint8_t* S, int* D) {
    int dx = 0;
    for (; dx < xmax; dx++) {
      int sx = xofs[dx];
      D[dx] = S[sx] * ialpha[dx * 2] + S[sx + 1] * ialpha[dx * 2 + 1];
    }
    for (; dx < dw; dx++) D[dx] = S[xofs[dx]] * 2048;
  };
  auto clip = [](int x, int a, int b) { return x >= a ? (x < b ? x : b - 1) : a; };
  for (int dy = 0; dy < dh; dy++) {
    int sy0 = clip(yofs[dy], 0, sh), sy1 = clip(yofs[dy] + 1, 0, sh);
    hresize(src + (size_t)sy0 * sstride, row0.data());
    hresize(src + (size_t)sy1 * sstride, row1.data());
    int b0 = ibeta[dy * 2], b1 = ibeta[dy * 2 + 1];
    uint8_t* D = dst + (size_t)dy * dstride;
    for (int x = 0; x < dw; x++)
      D[x] = (uint8_t)((((b0 * (row0[x] >> 4)) >> 16) + ((b1 * (row1[x] >> 4)) >> 16) + 2) >> 2);
  }
}

// ---------------------------------------------------------------------------
// cv::FAST(img, kps, threshold, true) default TYPE_9_16 — SURVEY §8(c)-F.
// Literal row-buffer formulation of OpenCV's FAST_t<16> + cornerScore<16>
// (the HIP path uses the closed single-pass form; this one is the checker).
// Called by the reference at src/ORBextractor.cc:826,845.
// ---------------------------------------------------------------------------
const int kCircle[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                            {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

int corner_score16(const uint8_t* ptr, const int pixel[25], int threshold) {
  const int K = 8, N = K * 3 + 1;
  int v = ptr[0];
  short d[N];
  for (int k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);
  int a0 = threshold;
  for (int k = 0; k < 16; k += 2) {
    int a = std::min((int)d[k + 1], (int)d[k + 2]);
    a = std::min(a, (int)d[k + 3]);
    if (a <= a0) continue;
    a = std::min(a, (int)d[k + 4]);
    a = std::min(a, (int)d[k + 5]);
    a = std::min(a, (int)d[k + 6]);
    a = std::min(a, (int)d[k + 7]);
    a = std::min(a, (int)d[k + 8]);
    a0 = std::max(a0, std::min(a, (int)d[k]));
    a0 = std::max(a0, std::min(a, (int)d[k + 9]));
  }
  int b0 = -a0;
  for (int k = 0; k < 16; k += 2) {
    int b = std::max((int)d[k + 1], (int)d[k + 2]);
    b = std::max(b, (int)d[k + 3]);
    b = std::max(b, (int)d[k + 4]);
    b = std::max(b, (int)d[k + 5]);
    if (b >= b0) continue;
    b = std::max(b, (int)d[k + 6]);
    b = std::max(b, (int)d[k + 7]);
    b = std::max(b, (int)d[k + 8]);
    b0 = std::min(b0, std::max(b, (int)d[k]));
    b0 = std::min(b0, std::max(b, (int)d[k + 9]));
  }
  return -b0 - 1;
}

void fast9_16(const uint8_t* img, int cols, int rows, int stride, int threshold, bool nms,
              std::vector<KeyPt>& out) {
  out.clear();
  const int K = 8, N = 25;
  int pixel[25];
  for (int k = 0; k < 16; k++) pixel[k] = kCircle[k][0] + kCircle[k][1] * stride;
  for (int k = 16; k < 25; k++) pixel[k] = pixel[k - 16];
  threshold = std::min(std::max(threshold, 0), 255);
  uint8_t tab[512];
  for (int i = -255; i <= 255; i++) tab[i + 255] = (uint8_t)(i < -threshold ? 1 : i > threshold ? 2 : 0);
  if (cols <= 0 || rows <= 0) return;
  std::vector<uint8_t> bufmem((size_t)cols * 3, 0);
  std::vector<int> cpmem((size_t)(cols + 1) * 3, 0);
  uint8_t* buf[3] = {bufmem.data(), bufmem.data() + cols, bufmem.data() + 2 * cols};
  int* cpbuf[3] = {cpmem.data(), cpmem.data() + (cols + 1), cpmem.data() + 2 * (cols + 1)};
  for (int i = 3; i < rows - 2; i++) {
    const uint8_t* ptr = img + (size_t)i * stride + 3;
    uint8_t* curr = buf[(i - 3) % 3];
    int* cornerpos = cpbuf[(i - 3) % 3] + 1;
    std::memset(curr, 0, cols);
    int ncorners = 0;
    if (i < rows - 3) {
      for (int j = 3; j < cols - 3; j++, ptr++) {
        int v = ptr[0];
        const uint8_t* t = &tab[0] - v + 255;
        int d = t[ptr[pixel[0]]] | t[ptr[pixel[8]]];
        if (d == 0) continue;
        d &= t[ptr[pixel[2]]] | t[ptr[pixel[10]]];
        d &= t[ptr[pixel[4]]] | t[ptr[pixel[12]]];
        d &= t[ptr[pixel[6]]] | t[ptr[pixel[14]]];
        if (d == 0) continue;
        d &= t[ptr[pixel[1]]] | t[ptr[pixel[9]]];
        d &= t[ptr[pixel[3]]] | t[ptr[pixel[11]]];
        d &= t[ptr[pixel[5]]] | t[ptr[pixel[13]]];
        d &= t[ptr[pixel[7]]] | t[ptr[pixel[15]]];
        if (d & 1) {
          int vt = v - threshold, count = 0;
          for (int k = 0; k < N; k++) {
            int x = ptr[pixel[k]];
            if (x < vt) {
              if (++count > K) {
                cornerpos[ncorners++] = j;
                if (nms) curr[j] = (uint8_t)corner_score16(ptr, pixel, threshold);
                break;
              }
            } else
              count = 0;
          }
        }
        if (d & 2) {
          int vt = v + threshold, count = 0;
          for (int k = 0; k < N; k++) {
            int x = ptr[pixel[k]];
            if (x > vt) {
              if (++count > K) {
                cornerpos[ncorners++] = j;
                if (nms) curr[j] = (uint8_t)corner_score16(ptr, pixel, threshold);
                break;
              }
            } else
              count = 0;
          }
        }
      }
    }
    cornerpos[-1] = ncorners;
    if (i == 3) continue;
    const uint8_t* prev = buf[(i - 4 + 3) % 3];
    const uint8_t* pprev = buf[(i - 5 + 3) % 3];
    cornerpos = cpbuf[(i - 4 + 3) % 3] + 1;
    ncorners = cornerpos[-1];
    for (int k = 0; k < ncorners; k++) {
      int j = cornerpos[k];
      int score = prev[j];
      if (!nms || (score > prev[j + 1] && score > prev[j - 1] && score > pprev[j - 1] && score > pprev[j] &&
                   score > pprev[j + 1] && score > curr[j - 1] && score > curr[j] && score > curr[j + 1])) {
        KeyPt kp{(float)j, (float)(i - 1), 7.f, -1.f, (float)score, 0, -1};
        out.push_back(kp);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// cv::GaussianBlur(Size(7,7), 2, 2, BORDER_REFLECT_101) on contiguous CV_8UC1 —
// SURVEY §8(c)-G (8.8 fixed-point path).  Called at src/ORBextractor.cc:1133.
// ---------------------------------------------------------------------------
inline int reflect101(int p, int n) {
  if (n == 1) return 0;
  while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
  return p;
}

// Which cv::GaussianBlur the blur restates is a process-wide switch of the oracle (orbo_set_gauss_variant; the product's twin is
// orbx_set_option("gauss_kernel" / "gauss_round")), because the 8-bit Gaussian is the one primitive of this path whose BYTES depend on the
// OpenCV release (INTEGRATION.md section 6 has the table):
//   kernel 0  {18,34,48,56,48,34,18}: getGaussianKernelFixedPoint_ED — the 8.8 weights with the rounding error diffused from the outside
//             in, centre = 256 - 2 * sum(others); sum exactly 256.  OpenCV >= 4.5.1 (recalled).
//   kernel 1  {18,34,49,55,49,34,18}: every coefficient rounded on its own (sum 257).  OpenCV 3.4.2 .. 4.5.0's ufixedpoint16 kernel
//             (recalled) and, with the same integers, the 8-bit "smooth symmetrical" integer path of sepFilter2D in OpenCV <= 3.4.1
//             (createSeparableLinearFilter: float kernel x 256 -> CV_32S, bits = 8).
//   round 0   (acc + 2^15) >> 16, saturated to 255: the scalar fixed-point column pass (every 4.x; the scalar tail of 3.x).
//   round 1   round-half-to-even of acc / 2^16, saturated: the SSE2 column pass of OpenCV <= 3.4.1 (SymmColumnVec_32s8u sums the rows in
//             float — exact here — and converts with cvtps2dq); only exact .5 ties differ from round 0.
//   tail V    the last (w mod V) columns of a row are the SIMD loop's scalar tail and round half up whatever `round` says (V = 4 for
//             the SSE2 pass of <= 3.4.1; 8 / 16 / 32 = v_uint16::nlanes of the build's dispatch for 3.4.2 .. 4.5.0); 0 = none.
//   round 2   floor(acc / 2^16), saturated: what the SIMD column pass of 3.4.2 .. 4.5.0 yields when the kernel sums to 257 (its
//             signed-arithmetic bias correction assumes a sum of exactly 256) — recalled, the least certain of the three.
std::atomic<int> g_gauss_kernel{0}, g_gauss_round{0}, g_gauss_tail{0};   // tail V: the last (w mod V) columns round half up

void gaussian_kernel7_fixed(int k[7]) {
  // bit-exact kernel exp(-x^2/(2 sigma^2)), normalised, x256
  const double sigma = 2.0;
  double v[7], sum = 0;
  for (int i = 0; i < 7; i++) {
    double x = i - 3;
    v[i] = std::exp(-0.5 * x * x / (sigma * sigma));
    sum += v[i];
  }
  if (g_gauss_kernel.load() == 1) {   // each coefficient rounded separately
    for (int i = 0; i < 7; i++) k[i] = cv_round(v[i] / sum * 256.0);
    return;
  }
  // error-diffused from the outside in
  double err = 0;
  int s = 0;
  for (int i = 0; i < 3; i++) {
    double adj = v[i] / sum * 256.0 + err;
    int q = cv_round(adj);
    err = adj - q;
    k[i] = k[6 - i] = q;
    s += q;
  }
  k[3] = 256 - 2 * s;
}

void gaussian_blur7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride) {
  int k[7];
  gaussian_kernel7_fixed(k);
  const int rmode = g_gauss_round.load(), tail = g_gauss_tail.load();
  const int body = tail > 1 ? w - (w % tail) : w;
  std::vector<uint16_t> tmp((size_t)w * h);
  for (int y = 0; y < h; y++) {
    const uint8_t* S = src + (size_t)y * sstride;
    for (int x = 0; x < w; x++) {
      uint32_t acc = 0;
      for (int t = 0; t < 7; t++) acc += (uint32_t)k[t] * S[reflect101(x + t - 3, w)];
      tmp[(size_t)y * w + x] = (uint16_t)acc;  // <= 255*257 = 65535, exact in 16 bit
    }
  }
  for (int y = 0; y < h; y++) {
    if (rmode == 0) {   // the default rounding: the plain loop (this is also the timed cpu_baseline path)
      for (int x = 0; x < w; x++) {
        uint32_t acc = 0;
        for (int t = 0; t < 7; t++) acc += (uint32_t)k[t] * tmp[(size_t)reflect101(y + t - 3, h) * w + x];
        dst[(size_t)y * dstride + x] = (uint8_t)std::min((acc + 32768u) >> 16, 255u);
      }
      continue;
    }
    for (int x = 0; x < w; x++) {
      uint32_t acc = 0;
      for (int t = 0; t < 7; t++) acc += (uint32_t)k[t] * tmp[(size_t)reflect101(y + t - 3, h) * w + x];
      uint32_t r;
      const int m = x < body ? rmode : 0;   // the scalar tail of a SIMD column pass rounds half up
      if (m == 2) r = acc >> 16;
      else {
        r = (acc + 32768u) >> 16;
        if (m == 1 && (acc & 0xffffu) == 0x8000u) r &= ~1u;   // tie -> even
      }
      dst[(size_t)y * dstride + x] = (uint8_t)std::min(r, 255u);
    }
  }
}

// ---------------------------------------------------------------------------
// cv::fastAtan2(y, x) in degrees — SURVEY §8(c)-A.  Called at src/ORBextractor.cc:102.
// ---------------------------------------------------------------------------
// g_atan_fma = 1: the same polynomial with the contractions a compiler makes when OpenCV's file is built with -mfma (the AVX2 dispatch
// copy of mathfuncs_core; GCC / clang contract a * b + c by default): three fused Horner steps, and 90 - poly * c as one fnma.
std::atomic<int> g_atan_fma{0};
float fast_atan2(float y, float x) {
  const float p1 = 0.9997878412794807f * (float)(180 / M_PI);
  const float p3 = -0.3258083974640975f * (float)(180 / M_PI);
  const float p5 = 0.1555786518463281f * (float)(180 / M_PI);
  const float p7 = -0.04432655554792128f * (float)(180 / M_PI);
  const bool fma = g_atan_fma.load(std::memory_order_relaxed) != 0;
  float ax = std::fabs(x), ay = std::fabs(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)DBL_EPSILON);
    c2 = c * c;
    if (fma) a = fmaf(fmaf(fmaf(p7, c2, p5), c2, p3), c2, p1) * c;
    else a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + (float)DBL_EPSILON);
    c2 = c * c;
    if (fma) a = fmaf(-fmaf(fmaf(fmaf(p7, c2, p5), c2, p3), c2, p1), c, 90.f);
    else a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

// glibc cosf/sinf through volatile pointers so the compiler can neither fold nor merge them.
float (*volatile p_cosf)(float) = cosf;
float (*volatile p_sinf)(float) = sinf;

struct Image {
  int w = 0, h = 0;
  std::vector<uint8_t> px;  // contiguous, stride == w
  const uint8_t* row(int y) const { return px.data() + (size_t)y * w; }
};

// src/ORBextractor.cc:76-103
float ic_angle(const Image& im, float ptx, float pty, const std::vector<int>& umax) {
  int m_01 = 0, m_10 = 0;
  const int step = im.w;
  const uint8_t* center = im.row(cv_round(pty)) + cv_round(ptx);
  for (int u = -kHalfPatch; u <= kHalfPatch; ++u) m_10 += u * center[u];
  for (int v = 1; v <= kHalfPatch; ++v) {
    int v_sum = 0;
    int d = umax[v];
    for (int u = -d; u <= d; ++u) {
      int val_plus = center[u + v * step], val_minus = center[u - v * step];
      v_sum += (val_plus - val_minus);
      m_10 += u * (val_plus + val_minus);
    }
    m_01 += v * v_sum;
  }
  return fast_atan2((float)m_01, (float)m_10);
}

// g_brief_fma = 1: the rotation of the test pattern as GCC / clang compile src/ORBextractor.cc:118-120 when the reference is built the way
// its CMakeLists.txt:10-13 asks (-O3 -march=native on an FMA machine, default -ffp-contract): x*b + y*a -> fma(x, b, y*a),
// x*a - y*b -> fma(x, a, -(y*b)) (the first product is the fused one).  Pinned by oracle/_ref/libref_orbextractor_fma.so = the reference's
// file compiled here with -O3 -mfma (tests/test_opencv_variants.py).  0 = separate IEEE operations (the canonical form).
std::atomic<int> g_brief_fma{0};
// src/ORBextractor.cc:106-146
void orb_descriptor(const KeyPt& kp, const Image& blurred, uint8_t* desc) {
  const float factorPI = (float)(M_PI / 180.f);
  float angle = (float)kp.angle * factorPI;
  float a = (float)p_cosf(angle), b = (float)p_sinf(angle);
  const int step = blurred.w;
  const uint8_t* center = blurred.row(cv_round(kp.y)) + cv_round(kp.x);
  const int8_t* pat = kPattern;
  const bool bfma = g_brief_fma.load(std::memory_order_relaxed) != 0;
  auto tap = [&](int idx) -> int {
    float px = (float)pat[idx * 2], py = (float)pat[idx * 2 + 1];
    int ry, rx;
    if (bfma) { ry = cv_round(fmaf(px, b, py * a)); rx = cv_round(fmaf(px, a, -(py * b))); }
    else { ry = cv_round(px * b + py * a); rx = cv_round(px * a - py * b); }
    return center[ry * step + rx];
  };
  for (int i = 0; i < 32; ++i, pat += 32) {
    int val = 0;
    for (int j = 0; j < 8; j++) {
      int t0 = tap(2 * j), t1 = tap(2 * j + 1);
      val |= (t0 < t1) << j;
    }
    desc[i] = (uint8_t)val;
  }
}

// src/ORBextractor.cc:480-536 (ExtractorNode) — coordinates kept as ints like cv::Point2i.
struct Node {
  std::vector<KeyPt> keys;
  int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy;
  std::list<Node>::iterator lit;
  bool no_more = false;
  void divide(Node& n1, Node& n2, Node& n3, Node& n4) const {
    const int halfX = (int)std::ceil(static_cast<float>(URx - ULx) / 2);
    const int halfY = (int)std::ceil(static_cast<float>(BRy - ULy) / 2);
    n1.ULx = ULx; n1.ULy = ULy;
    n1.URx = ULx + halfX; n1.URy = ULy;
    n1.BLx = ULx; n1.BLy = ULy + halfY;
    n1.BRx = ULx + halfX; n1.BRy = ULy + halfY;
    n2.ULx = n1.URx; n2.ULy = n1.URy;
    n2.URx = URx; n2.URy = URy;
    n2.BLx = n1.BRx; n2.BLy = n1.BRy;
    n2.BRx = URx; n2.BRy = ULy + halfY;
    n3.ULx = n1.BLx; n3.ULy = n1.BLy;
    n3.URx = n1.BRx; n3.URy = n1.BRy;
    n3.BLx = BLx; n3.BLy = BLy;
    n3.BRx = n1.BRx; n3.BRy = BLy;
    n4.ULx = n3.URx; n4.ULy = n3.URy;
    n4.URx = n2.BRx; n4.URy = n2.BRy;
    n4.BLx = n3.BRx; n4.BLy = n3.BRy;
    n4.BRx = BRx; n4.BRy = BRy;
    for (const KeyPt& kp : keys) {
      if (kp.x < n1.URx) {
        if (kp.y < n1.BRy) n1.keys.push_back(kp);
        else n3.keys.push_back(kp);
      } else if (kp.y < n1.BRy)
        n2.keys.push_back(kp);
      else
        n4.keys.push_back(kp);
    }
    if (n1.keys.size() == 1) n1.no_more = true;
    if (n2.keys.size() == 1) n2.no_more = true;
    if (n3.keys.size() == 1) n3.no_more = true;
    if (n4.keys.size() == 1) n4.no_more = true;
  }
};

// src/ORBextractor.cc:538-553
bool compare_nodes(std::pair<int, Node*>& e1, std::pair<int, Node*>& e2) {
  if (e1.first < e2.first) return true;
  if (e1.first > e2.first) return false;
  return e1.second->ULx < e2.second->ULx;
}

// src/ORBextractor.cc:555-779
std::vector<KeyPt> distribute_octree(const std::vector<KeyPt>& cand, int minX, int maxX, int minY, int maxY, int N,
                                     int* stat_sorted_phase) {
  const int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY));
  const float hX = static_cast<float>(maxX - minX) / nIni;
  std::list<Node> nodes;
  std::vector<Node*> ini(nIni);
  for (int i = 0; i < nIni; i++) {
    Node ni;
    ni.ULx = (int)(hX * static_cast<float>(i)); ni.ULy = 0;
    ni.URx = (int)(hX * static_cast<float>(i + 1)); ni.URy = 0;
    ni.BLx = ni.ULx; ni.BLy = maxY - minY;
    ni.BRx = ni.URx; ni.BRy = maxY - minY;
    nodes.push_back(ni);
    ini[i] = &nodes.back();
  }
  for (const KeyPt& kp : cand) ini[(size_t)(kp.x / hX)]->keys.push_back(kp);
  for (auto lit = nodes.begin(); lit != nodes.end();) {
    if (lit->keys.size() == 1) { lit->no_more = true; ++lit; }
    else if (lit->keys.empty()) lit = nodes.erase(lit);
    else ++lit;
  }
  bool finish = false;
  std::vector<std::pair<int, Node*>> size_ptr;
  auto push_children = [&](Node& c, int* nToExpand) {
    if (c.keys.size() > 0) {
      nodes.push_front(c);
      if (c.keys.size() > 1) {
        if (nToExpand) ++*nToExpand;
        size_ptr.push_back(std::make_pair((int)c.keys.size(), &nodes.front()));
        nodes.front().lit = nodes.begin();
      }
    }
  };
  while (!finish) {
    int prevSize = (int)nodes.size();
    auto lit = nodes.begin();
    int nToExpand = 0;
    size_ptr.clear();
    while (lit != nodes.end()) {
      if (lit->no_more) { ++lit; continue; }
      Node n1, n2, n3, n4;
      lit->divide(n1, n2, n3, n4);
      push_children(n1, &nToExpand);
      push_children(n2, &nToExpand);
      push_children(n3, &nToExpand);
      push_children(n4, &nToExpand);
      lit = nodes.erase(lit);
    }
    if ((int)nodes.size() >= N || (int)nodes.size() == prevSize) {
      finish = true;
    } else if (((int)nodes.size() + nToExpand * 3) > N) {
      if (stat_sorted_phase) ++*stat_sorted_phase;
      while (!finish) {
        prevSize = (int)nodes.size();
        std::vector<std::pair<int, Node*>> prev = size_ptr;
        size_ptr.clear();
        std::sort(prev.begin(), prev.end(), compare_nodes);
        for (int j = (int)prev.size() - 1; j >= 0; j--) {
          Node n1, n2, n3, n4;
          prev[j].second->divide(n1, n2, n3, n4);
          push_children(n1, nullptr);
          push_children(n2, nullptr);
          push_children(n3, nullptr);
          push_children(n4, nullptr);
          nodes.erase(prev[j].second->lit);
          if ((int)nodes.size() >= N) break;
        }
        if ((int)nodes.size() >= N || (int)nodes.size() == prevSize) finish = true;
      }
    }
  }
  std::vector<KeyPt> result;
  for (auto& nd : nodes) {
    const KeyPt* best = &nd.keys[0];
    float maxResponse = best->response;
    for (size_t k = 1; k < nd.keys.size(); k++) {
      if (nd.keys[k].response > maxResponse) {
        best = &nd.keys[k];
        maxResponse = nd.keys[k].response;
      }
    }
    result.push_back(*best);
  }
  return result;
}

struct Extractor {
  int nfeatures, nlevels, iniTh, minTh;
  double scaleFactor;  // include/ORBextractor.h:96 — double member initialised from a float
  std::vector<float> scale, inv_scale, sigma2, inv_sigma2;
  std::vector<int> quota, umax;
  // state of the last extract() (stage dumps)
  std::vector<Image> pyr, blurred;
  std::vector<std::vector<KeyPt>> cand, kps;
  int sorted_phase_count = 0;

  // src/ORBextractor.cc:409-469
  Extractor(int nf, float sf, int nl, int ini, int mn) : nfeatures(nf), nlevels(nl), iniTh(ini), minTh(mn), scaleFactor(sf) {
    scale.resize(nlevels); sigma2.resize(nlevels);
    scale[0] = 1.0f; sigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {
      scale[i] = (float)(scale[i - 1] * scaleFactor);
      sigma2[i] = scale[i] * scale[i];
    }
    inv_scale.resize(nlevels); inv_sigma2.resize(nlevels);
    for (int i = 0; i < nlevels; i++) {
      inv_scale[i] = 1.0f / scale[i];
      inv_sigma2[i] = 1.0f / sigma2[i];
    }
    quota.resize(nlevels);
    float factor = (float)(1.0f / scaleFactor);
    float desired = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int level = 0; level < nlevels - 1; level++) {
      quota[level] = cv_round(desired);
      sum += quota[level];
      desired *= factor;
    }
    quota[nlevels - 1] = std::max(nfeatures - sum, 0);
    umax.resize(kHalfPatch + 1);
    int v, v0, vmax = (int)std::floor(kHalfPatch * std::sqrt(2.f) / 2 + 1);
    int vmin = (int)std::ceil(kHalfPatch * std::sqrt(2.f) / 2);
    const double hp2 = kHalfPatch * kHalfPatch;
    for (v = 0; v <= vmax; ++v) umax[v] = cv_round(std::sqrt(hp2 - v * v));
    for (v = kHalfPatch, v0 = 0; v >= vmin; --v) {
      while (umax[v0] == umax[v0 + 1]) ++v0;
      umax[v] = v0;
      ++v0;
    }
  }

  // src/ORBextractor.cc:1170-1195 (padding bytes are never read by the extractor; levels stored unpadded)
  void compute_pyramid(const uint8_t* img, int rows, int cols, int stride) {
    pyr.assign(nlevels, Image());
    for (int level = 0; level < nlevels; ++level) {
      float s = inv_scale[level];
      int w = cv_round((float)cols * s), h = cv_round((float)rows * s);
      Image& L = pyr[level];
      L.w = w; L.h = h; L.px.resize((size_t)w * h);
      if (level != 0) {
        const Image& P = pyr[level - 1];
        resize_linear_u8(P.px.data(), P.w, P.h, P.w, L.px.data(), w, h, w);
      } else {
        for (int y = 0; y < rows; y++) std::memcpy(L.px.data() + (size_t)y * w, img + (size_t)y * stride, cols);
      }
    }
  }

  // src/ORBextractor.cc:781-896
  void compute_keypoints() {
    cand.assign(nlevels, {});
    kps.assign(nlevels, {});
    const float W = 35;
    for (int level = 0; level < nlevels; ++level) {
      const Image& im = pyr[level];
      const int minBorderX = kEdgeThreshold - 3, minBorderY = minBorderX;
      const int maxBorderX = im.w - kEdgeThreshold + 3, maxBorderY = im.h - kEdgeThreshold + 3;
      std::vector<KeyPt>& toDistribute = cand[level];
      const float width = (float)(maxBorderX - minBorderX), height = (float)(maxBorderY - minBorderY);
      const int nCols = (int)(width / W), nRows = (int)(height / W);
      const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
      std::vector<KeyPt> cell;
      for (int i = 0; i < nRows; i++) {
        const float iniY = (float)(minBorderY + i * hCell);
        float maxY = iniY + hCell + 6;
        if (iniY >= maxBorderY - 3) continue;
        if (maxY > maxBorderY) maxY = (float)maxBorderY;
        for (int j = 0; j < nCols; j++) {
          const float iniX = (float)(minBorderX + j * wCell);
          float maxX = iniX + wCell + 6;
          if (iniX >= maxBorderX - 6) continue;
          if (maxX > maxBorderX) maxX = (float)maxBorderX;
          const int x0 = (int)iniX, x1 = (int)maxX, y0 = (int)iniY, y1 = (int)maxY;
          const uint8_t* sub = im.row(y0) + x0;
          fast9_16(sub, x1 - x0, y1 - y0, im.w, iniTh, true, cell);
          if (cell.empty()) fast9_16(sub, x1 - x0, y1 - y0, im.w, minTh, true, cell);
          for (KeyPt kp : cell) {
            kp.x += j * wCell;
            kp.y += i * hCell;
            toDistribute.push_back(kp);
          }
        }
      }
      std::vector<KeyPt>& keypoints = kps[level];
      keypoints = distribute_octree(toDistribute, minBorderX, maxBorderX, minBorderY, maxBorderY, quota[level],
                                    &sorted_phase_count);
      const int scaledPatchSize = (int)(kPatchSize * scale[level]);
      for (KeyPt& kp : keypoints) {
        kp.x += minBorderX;
        kp.y += minBorderY;
        kp.octave = level;
        kp.size = (float)scaledPatchSize;
      }
    }
    for (int level = 0; level < nlevels; ++level)
      for (KeyPt& kp : kps[level]) kp.angle = ic_angle(pyr[level], kp.x, kp.y, umax);
  }

  // src/ORBextractor.cc:1086-1168
  int extract(const uint8_t* img, int rows, int cols, int stride, int lap0, int lap1, KeyPt* out_kps,
              uint8_t* out_desc, int cap, int* n_out, int* mono_out) {
    *n_out = 0; *mono_out = 0;
    if (!img || rows <= 0 || cols <= 0) return -1;
    sorted_phase_count = 0;
    compute_pyramid(img, rows, cols, stride);
    compute_keypoints();
    int nkeypoints = 0;
    for (int level = 0; level < nlevels; ++level) nkeypoints += (int)kps[level].size();
    *n_out = nkeypoints;
    if (nkeypoints > cap) return -2;
    blurred.assign(nlevels, Image());
    int monoIndex = 0, stereoIndex = nkeypoints - 1;
    for (int level = 0; level < nlevels; ++level) {
      std::vector<KeyPt>& keypoints = kps[level];
      if (keypoints.empty()) continue;
      Image& B = blurred[level];
      B.w = pyr[level].w; B.h = pyr[level].h; B.px.resize(pyr[level].px.size());
      gaussian_blur7(pyr[level].px.data(), B.w, B.h, B.w, B.px.data(), B.w);
      std::vector<uint8_t> desc(keypoints.size() * 32);
      for (size_t i = 0; i < keypoints.size(); i++) orb_descriptor(keypoints[i], B, &desc[i * 32]);
      float s = scale[level];
      int i = 0;
      for (const KeyPt& kp0 : keypoints) {
        KeyPt kp = kp0;  // keep level coordinates in the stage dump
        if (level != 0) { kp.x *= s; kp.y *= s; }
        int idx;
        if (kp.x >= lap0 && kp.x <= lap1) idx = stereoIndex--;
        else idx = monoIndex++;
        out_kps[idx] = kp;
        std::memcpy(out_desc + (size_t)idx * 32, &desc[(size_t)i * 32], 32);
        i++;
      }
    }
    *mono_out = monoIndex;
    return 0;
  }
};

}  // namespace

extern "C" {

void* orbo_create(int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh) {
  return new Extractor(nfeatures, scaleFactor, nlevels, iniTh, minTh);
}
void orbo_destroy(void* h) { delete (Extractor*)h; }

int orbo_extract(void* h, const uint8_t* img, int rows, int cols, int stride, int lap0, int lap1, void* kps,
                 uint8_t* desc, int cap, int* n_out, int* mono_out) {
  return ((Extractor*)h)->extract(img, rows, cols, stride, lap0, lap1, (KeyPt*)kps, desc, cap, n_out, mono_out);
}

void orbo_tables(void* h, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2, int* quota, int* umax16) {
  Extractor* e = (Extractor*)h;
  for (int i = 0; i < e->nlevels; i++) {
    if (scale) scale[i] = e->scale[i];
    if (inv_scale) inv_scale[i] = e->inv_scale[i];
    if (sigma2) sigma2[i] = e->sigma2[i];
    if (inv_sigma2) inv_sigma2[i] = e->inv_sigma2[i];
    if (quota) quota[i] = e->quota[i];
  }
  if (umax16) for (int i = 0; i < 16; i++) umax16[i] = e->umax[i];
}

int orbo_level_size(void* h, int level, int* w, int* hgt) {
  Extractor* e = (Extractor*)h;
  if (level < 0 || level >= (int)e->pyr.size()) return -1;
  *w = e->pyr[level].w; *hgt = e->pyr[level].h;
  return 0;
}
int orbo_level_copy(void* h, int level, int blurred, uint8_t* dst, int dst_stride) {
  Extractor* e = (Extractor*)h;
  const std::vector<Image>& v = blurred ? e->blurred : e->pyr;
  if (level < 0 || level >= (int)v.size() || v[level].px.empty()) return -1;
  for (int y = 0; y < v[level].h; y++) std::memcpy(dst + (size_t)y * dst_stride, v[level].row(y), v[level].w);
  return 0;
}
// stage 0 = FAST candidates handed to the quadtree (cell-shifted, border-relative coords);
// stage 1 = distributed keypoints with orientation (level coords)
int orbo_level_keypoints(void* h, int level, int stage, void* dst, int cap) {
  Extractor* e = (Extractor*)h;
  const auto& v = stage == 0 ? e->cand : e->kps;
  if (level < 0 || level >= (int)v.size()) return -1;
  int n = (int)v[level].size();
  if (dst) std::memcpy(dst, v[level].data(), sizeof(KeyPt) * (size_t)std::min(n, cap));
  return n;
}
int orbo_sorted_phase_count(void* h) { return ((Extractor*)h)->sorted_phase_count; }

// CPU-baseline leg of bench.py: a frame-parallel std::thread pool (SURVEY.md §8(d) "CPU timing plan"), `nthreads` workers with
// one extractor each pulling frames off a shared counter (frame i = frames[i % nframes], src/ORBextractor.cc is
// single-threaded per image) until `seconds` have passed.  Returns the wall time from the common start to the last join;
// *features_out / *frames_out = totals.
double orbo_extract_many(int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh, const uint8_t* frames, int nframes, int rows,
                         int cols, int lap0, int lap1, int nthreads, double seconds, long long* features_out, long long* frames_out) {
  if (nthreads < 1) nthreads = 1;
  std::atomic<long long> next(0), feats(0), done(0);
  std::atomic<int> ready(0);
  std::atomic<bool> go(false);
  std::chrono::steady_clock::time_point t0;
  auto worker = [&]() {
    Extractor ex(nfeatures, scaleFactor, nlevels, iniTh, minTh);
    const int cap = nfeatures + 3 * nlevels + 64;
    std::vector<KeyPt> kps(cap);
    std::vector<uint8_t> desc((size_t)cap * 32);
    int n = 0, mono = 0;
    ex.extract(frames, rows, cols, cols, lap0, lap1, kps.data(), desc.data(), cap, &n, &mono);   // warm-up (allocations, page faults)
    ready.fetch_add(1);
    while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
    long long mine = 0, nf = 0;
    for (;;) {
      const long long i = next.fetch_add(1);
      ex.extract(frames + (size_t)(i % nframes) * rows * cols, rows, cols, cols, lap0, lap1, kps.data(), desc.data(), cap, &n, &mono);
      mine += n; nf++;
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() >= seconds) break;
    }
    feats.fetch_add(mine); done.fetch_add(nf);
  };
  std::vector<std::thread> th;
  for (int t = 0; t < nthreads; t++) th.emplace_back(worker);
  while (ready.load() < nthreads) std::this_thread::yield();
  t0 = std::chrono::steady_clock::now();
  go.store(true, std::memory_order_release);
  for (auto& t : th) t.join();
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  *features_out = feats.load(); *frames_out = done.load();
  return dt;
}

// cv::resize(src, dst, dsize) as System::TrackMonocular calls it on the incoming image (src/System.cc:441-446, INTER_LINEAR by
// default): the fixed-point bilinear path above, except that OpenCV turns an EXACT 2 x 2 downscale into INTER_AREA
// (imgproc/src/resize.cpp: `interpolation == INTER_LINEAR && is_area_fast && iscale_x == 2 && iscale_y == 2`), i.e. the
// rounded mean of each 2 x 2 block.  Recalled OpenCV semantics like the other primitives.
void orbo_cv_resize(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride) {
  if (sw == 2 * dw && sh == 2 * dh) {
    for (int y = 0; y < dh; y++)
      for (int x = 0; x < dw; x++) {
        const uint8_t* S = src + (size_t)(2 * y) * sstride + 2 * x;
        dst[(size_t)y * dstride + x] = (uint8_t)((S[0] + S[1] + S[sstride] + S[sstride + 1] + 2) >> 2);
      }
    return;
  }
  resize_linear_u8(src, sw, sh, sstride, dst, dw, dh, dstride);
}

// isolated primitives
void orbo_resize_linear(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride) {
  resize_linear_u8(src, sw, sh, sstride, dst, dw, dh, dstride);
}
int orbo_fast(const uint8_t* img, int cols, int rows, int stride, int threshold, int nms, void* dst, int cap) {
  std::vector<KeyPt> v;
  fast9_16(img, cols, rows, stride, threshold, nms != 0, v);
  if (dst) std::memcpy(dst, v.data(), sizeof(KeyPt) * std::min((size_t)cap, v.size()));
  return (int)v.size();
}
void orbo_gaussian_blur7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride) {
  gaussian_blur7(src, w, h, sstride, dst, dstride);
}
void orbo_gaussian_kernel7(int* k) { gaussian_kernel7_fixed(k); }
// which OpenCV GaussianBlur the oracle (and everything compiled over oracle/ref_shims) restates; returns 0, or -1 for values out of range
int orbo_set_gauss_variant(int kernel, int round) {
  if (kernel < 0 || kernel > 1 || round < 0 || round > 2) return -1;
  g_gauss_kernel.store(kernel); g_gauss_round.store(round);
  return 0;
}
int orbo_set_gauss_tail(int v) {
  if (!(v == 0 || v == 4 || v == 8 || v == 16 || v == 32 || v == 64)) return -1;
  g_gauss_tail.store(v);
  return 0;
}
int orbo_get_gauss_tail() { return g_gauss_tail.load(); }
int orbo_set_atan_fma(int on) { if (on != 0 && on != 1) return -1; g_atan_fma.store(on); return 0; }
int orbo_get_atan_fma() { return g_atan_fma.load(); }
int orbo_set_brief_fma(int on) { if (on != 0 && on != 1) return -1; g_brief_fma.store(on); return 0; }
// ORBO_VARIANT="kernel,round,tail,atan_fma,brief_fma" in the environment selects the variant at load time: how a test makes the shim of
// oracle/ref_shims behave like another OpenCV build inside a separate executable (tests/test_adapters.py)
static const int g_env_variant = [] {
  const char* e = getenv("ORBO_VARIANT");
  int v[5] = {0, 0, 0, 0, 0};
  if (e && sscanf(e, "%d,%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3], &v[4]) == 5 && v[0] >= 0 && v[0] <= 1 && v[1] >= 0 && v[1] <= 2 &&
      (v[2] == 0 || v[2] == 4 || v[2] == 8 || v[2] == 16 || v[2] == 32 || v[2] == 64) && (v[3] | 1) == 1 && (v[4] | 1) == 1) {
    g_gauss_kernel.store(v[0]); g_gauss_round.store(v[1]); g_gauss_tail.store(v[2]); g_atan_fma.store(v[3]); g_brief_fma.store(v[4]);
    return 1;
  }
  return 0;
}();
int orbo_get_brief_fma() { return g_brief_fma.load(); }
void orbo_get_gauss_variant(int* kernel, int* round) { if (kernel) *kernel = g_gauss_kernel.load(); if (round) *round = g_gauss_round.load(); }
float orbo_fast_atan2(float y, float x) { return fast_atan2(y, x); }
void orbo_cos_sin_deg(float angle_deg, float* a, float* b) {
  const float factorPI = (float)(M_PI / 180.f);
  float r = angle_deg * factorPI;
  *a = p_cosf(r);
  *b = p_sinf(r);
}
// Order-independent 64-bit digest of (cosf, sinf)(angle * factorPI) over the float bit patterns [first, first + count):
// the CPU side of the exhaustive device check (every angle fastAtan2 can return lies in [0, 360]).
uint64_t orbo_trig_hash(uint32_t first, uint32_t count) {
  const float factorPI = (float)(M_PI / 180.f);
  unsigned nt = std::thread::hardware_concurrency();
  if (nt == 0) nt = 1;
  std::vector<uint64_t> part(nt, 0);
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; t++)
    th.emplace_back([&, t] {
      uint64_t h = 0;
      for (uint64_t i = t; i < count; i += nt) {
        const uint32_t bits = first + (uint32_t)i;
        float ang;
        std::memcpy(&ang, &bits, 4);
        const float r = ang * factorPI;
        const float a = p_cosf(r), b = p_sinf(r);
        uint32_t ab, bb;
        std::memcpy(&ab, &a, 4);
        std::memcpy(&bb, &b, 4);
        h += ((uint64_t)ab * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)bb * 0xC2B2AE3D27D4EB4Full + bits);
      }
      part[t] = h;
    });
  for (auto& x : th) x.join();
  uint64_t h = 0;
  for (uint64_t v : part) h += v;
  return h;
}
// Digest of fastAtan2 over `count` pseudo-random integer moment pairs (m01, m10), |m| <= 3e6 (the range IC_Angle can
// produce), derived from the index by a fixed integer mix — the CPU side of orbx_debug_atan_hash.
static inline uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
uint64_t orbo_atan_hash(uint32_t seed, uint32_t count) {
  unsigned nt = std::thread::hardware_concurrency();
  if (nt == 0) nt = 1;
  std::vector<uint64_t> part(nt, 0);
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; t++)
    th.emplace_back([&, t] {
      uint64_t h = 0;
      for (uint64_t i = t; i < count; i += nt) {
        const uint32_t a = mix32(seed + 2u * (uint32_t)i), b = mix32(seed + 2u * (uint32_t)i + 1u);
        const int m01 = (int)(a % 6000001u) - 3000000, m10 = (b & 15u) == 0 ? 0 : (int)(b % 6000001u) - 3000000;
        const float ang = fast_atan2((float)m01, (float)m10);
        uint32_t bits;
        std::memcpy(&bits, &ang, 4);
        h += ((uint64_t)bits * 0x9E3779B97F4A7C15ull) ^ (uint64_t)(uint32_t)i;
      }
      part[t] = h;
    });
  for (auto& x : th) x.join();
  uint64_t h = 0;
  for (uint64_t v : part) h += v;
  return h;
}
// Digest of the rotated test pattern (src/ORBextractor.cc:118-120) over `count` consecutive float bit patterns of the keypoint angle,
// first + i (degrees, as kpt.angle holds it): a, b = cosf / sinf(angle * factorPI), then (ry, rx) of all 512 pattern points — the CPU side
// of orbx_debug_brief_hash.  Honours the brief_fma switch.  *n_diff (optional): how many of the count * 512 points would round
// differently under the other setting of the switch.
uint64_t orbo_brief_hash(uint32_t first, uint32_t count, uint64_t* n_diff) {
  unsigned nt = std::thread::hardware_concurrency();
  if (nt == 0) nt = 1;
  std::vector<uint64_t> part(nt, 0), diff(nt, 0);
  std::vector<std::thread> th;
  const bool bfma = g_brief_fma.load() != 0;
  const float factorPI = (float)(M_PI / 180.f);
  for (unsigned t = 0; t < nt; t++)
    th.emplace_back([&, t] {
      uint64_t h = 0, nd = 0;
      for (uint64_t i = t; i < count; i += nt) {
        const uint32_t bits = first + (uint32_t)i;
        float ang;
        std::memcpy(&ang, &bits, 4);
        const float r = ang * factorPI;
        const float a = p_cosf(r), b = p_sinf(r);
        uint64_t hk = 0;
        for (int idx = 0; idx < 512; idx++) {
          const float px = (float)kPattern[idx * 2], py = (float)kPattern[idx * 2 + 1];
          const int ryf = cv_round(fmaf(px, b, py * a)), rxf = cv_round(fmaf(px, a, -(py * b)));
          const int ryu = cv_round(px * b + py * a), rxu = cv_round(px * a - py * b);
          nd += (ryf != ryu) + (rxf != rxu);
          const int ry = bfma ? ryf : ryu, rx = bfma ? rxf : rxu;
          hk += (uint64_t)(uint32_t)(ry * 64 + rx + 4096) * (0x9E3779B97F4A7C15ull + 2u * (uint64_t)idx);
        }
        h += hk ^ (uint64_t)bits;
      }
      part[t] = h; diff[t] = nd;
    });
  for (auto& x : th) x.join();
  uint64_t h = 0, nd = 0;
  for (unsigned t = 0; t < nt; t++) { h += part[t]; nd += diff[t]; }
  if (n_diff) *n_diff = nd;
  return h;
}
// one rotated pattern point under the current brief_fma switch (tests/support/contract_probe.cpp is compared with this)
void orbo_rot_tap(int x, int y, float a, float b, int* ry, int* rx) {
  const float px = (float)x, py = (float)y;
  if (g_brief_fma.load() != 0) { *ry = cv_round(fmaf(px, b, py * a)); *rx = cv_round(fmaf(px, a, -(py * b))); }
  else { *ry = cv_round(px * b + py * a); *rx = cv_round(px * a - py * b); }
}
// the operand sequence and digest of tests/support/contract_probe.cpp under the current brief_fma switch; *n_diff = operand sets at which
// the two settings of the switch disagree
uint64_t orbo_rot_probe_hash(uint32_t n, uint64_t* n_diff) {
  unsigned nt = std::thread::hardware_concurrency();
  if (nt == 0) nt = 1;
  std::vector<uint64_t> part(nt, 0), diff(nt, 0);
  std::vector<std::thread> th;
  const bool bfma = g_brief_fma.load() != 0;
  for (unsigned t = 0; t < nt; t++)
    th.emplace_back([&, t] {
      uint64_t h = 0, nd = 0;
      for (uint64_t i = t; i < n; i += nt) {
        const uint32_t r0 = mix32(2 * (uint32_t)i + 1), r1 = mix32(2 * (uint32_t)i + 2);
        const float px = (float)((int)(r0 % 27u) - 13), py = (float)((int)((r0 >> 8) % 27u) - 13);
        const float ang = (float)(r1 % 36000001u) * 1e-5f * (float)(M_PI / 180.f);
        const float a = p_cosf(ang), b = p_sinf(ang);
        const int ryf = cv_round(fmaf(px, b, py * a)), rxf = cv_round(fmaf(px, a, -(py * b)));
        const int ryu = cv_round(px * b + py * a), rxu = cv_round(px * a - py * b);
        nd += (ryf != ryu) || (rxf != rxu);
        const int ry = bfma ? ryf : ryu, rx = bfma ? rxf : rxu;
        h += (uint64_t)(uint32_t)(ry * 64 + rx + 4096) * (0x9E3779B97F4A7C15ull + 2ull * (uint64_t)(i & 1023u));
      }
      part[t] = h; diff[t] = nd;
    });
  for (auto& x : th) x.join();
  uint64_t h = 0, nd = 0;
  for (unsigned t = 0; t < nt; t++) { h += part[t]; nd += diff[t]; }
  if (n_diff) *n_diff = nd;
  return h;
}
int orbo_distribute(const void* cand, int n, int minX, int maxX, int minY, int maxY, int N, void* dst, int cap) {
  std::vector<KeyPt> c((const KeyPt*)cand, (const KeyPt*)cand + n);
  std::vector<KeyPt> r = distribute_octree(c, minX, maxX, minY, maxY, N, nullptr);
  if (dst) std::memcpy(dst, r.data(), sizeof(KeyPt) * std::min((size_t)cap, r.size()));
  return (int)r.size();
}
const int8_t* orbo_pattern() { return kPattern; }

}  // extern "C"
