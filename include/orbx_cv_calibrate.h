/* orbx — which OpenCV is the CPU path I replace?  Self-calibration of the drop-in against the OpenCV it is built with.
 *
 * "Bit-exact with the reference CPU path" (BASELINE.json north_star) has three inputs the reference does not fix:
 *   (1) the OpenCV release: cv::GaussianBlur(Size(7,7), 2, 2, BORDER_REFLECT_101) of CV_8UC1 (src/ORBextractor.cc:1133) changed its
 *       fixed-point weights and the rounding of its column pass between releases (CMakeLists.txt:33 asks for 4.4, README.md:560 names
 *       3.2.0 and 4.4.0; orbx's default equals >= 4.5.1);
 *   (2) the OpenCV BUILD: cv::fastAtan2 (src/ORBextractor.cc:102) is one polynomial, but its AVX2 dispatch copy is compiled with FMA
 *       contraction and then differs in the last bit for some arguments;
 *   (3) the reference's own build: CMakeLists.txt:10-13 compiles src/ORBextractor.cc with -O3 -march=native, under which GCC / clang
 *       contract the pattern rotation of :118-120 into FMAs (one rotated point in ten million rounds differently).
 * liborbx.so carries every known variant as a context option (include/orbx.h: "gauss_kernel", "gauss_round", "gauss_tail", "atan_fma",
 * "brief_fma").  This header finds out which one applies, AT RUN TIME, from the very libraries and flags the caller is built with — no
 * version table involved: it runs the real cv::GaussianBlur on a 127 x 72 probe image that separates all variants (noise, plus row
 * bands constructed so that the column pass hits exact .5 ties under either kernel, a saturated band, and a width whose SIMD tails
 * differ for every vector length), the real cv::fastAtan2 on 4096 moment pairs, and an expression of the shape of :118-120 compiled
 * in THIS translation unit on an operand at which contraction matters.  include/ORBextractor.h does this once per process in its
 * constructor and applies the result (ORBX_CV_CALIBRATE=0, or any of ORBX_GAUSS_KERNEL / ORBX_GAUSS_ROUND / ORBX_GAUSS_TAIL / ORBX_ATAN_FMA /
 * ORBX_BRIEF_FMA in the environment, switch that off).  When no variant reproduces the OpenCV at hand (an IPP or OpenCL path, a release
 * unknown to this file) it says so on stderr, with the number of differing bytes, and leaves the defaults.
 * tools/validate_opencv.cpp is the long form: every primitive, natural images, first mismatch.
 *
 * The host code below restates 60 lines of arithmetic for a 9 KB probe image; it is never on the per-frame path.
 */
#ifndef ORBX_CV_CALIBRATE_H
#define ORBX_CV_CALIBRATE_H

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <algorithm>
#include <utility>
#include <vector>

#include "orbx.h"

namespace orbx_cv {

struct Calibration {
  int gauss_kernel = 0, gauss_round = 0, gauss_tail = 0, atan_fma = 0, brief_fma = 0;
  bool gauss_exact = false, atan_exact = false;   // a variant reproduces every byte / bit of the probe
  int gauss_mismatch = 0, atan_mismatch = 0;      // differing bytes / angles of the closest variant otherwise
  int gauss_candidates = 0;                       // variants that reproduce the probe (>= 1 when gauss_exact)
  bool sort_libstdcxx = true;                     // this toolchain's std::sort leaves equal keys in libstdc++'s order (the one liborbx restates)
  int brief_form = 0;                             // build_contraction_form(): 0 unfused, 1 the form brief_fma = 1 reproduces, -1 a form liborbx does not have
  int frame_w = 0, frame_h = 0, frame_mismatch = -1;   // second probe at a real frame size: bytes at which the chosen variant differs (-1: not run)
};

constexpr int kFrameProbeW = 752, kFrameProbeH = 480;   // EuRoC's native size: wide enough for any size-dependent path of cv::GaussianBlur (IPP,
                                                        // OpenCL, parallel_for_ stripes) to show itself; 752 mod V covers the tails of V = 32, 64

constexpr int kProbeW = 127, kProbeH = 72;   // 127 mod V differs for V = 4, 8, 16, 32, 64: every tail length is visible

/* noise + tie bands (sum k_i a_i = 32768 under {18,34,49,55,..}: acc = 257 * 32768; = 32896 under {18,34,48,56,..}: acc = 256 * 32896)
 * + a saturated band (the 257 kernel reaches 256 there) + a black band */
inline void probe_image(std::vector<uint8_t>& img) {
  img.assign((size_t)kProbeW * kProbeH, 0);
  uint32_t s = 0x9E3779B9u;
  for (size_t i = 0; i < img.size(); i++) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; img[i] = (uint8_t)(s >> 11); }
  static const uint8_t t1[7] = {127, 128, 128, 126, 128, 128, 128}, t0[7] = {132, 128, 128, 129, 128, 128, 128};
  for (int r = 0; r < 7; r++) { std::memset(&img[(size_t)(6 + r) * kProbeW], t1[r], kProbeW); std::memset(&img[(size_t)(20 + r) * kProbeW], t0[r], kProbeW); }
  for (int r = 34; r < 44; r++) std::memset(&img[(size_t)r * kProbeW], 255, kProbeW);
  for (int r = 50; r < 58; r++) std::memset(&img[(size_t)r * kProbeW], 0, kProbeW);
}

inline void gauss_weights(int kernel, int k[7]) {
  double v[7], sum = 0;
  for (int i = 0; i < 7; i++) { const double x = i - 3; v[i] = std::exp(-0.5 * x * x / 4.0); sum += v[i]; }
  if (kernel == 1) { for (int i = 0; i < 7; i++) k[i] = (int)std::lrint(v[i] / sum * 256.0); return; }
  double err = 0; int s = 0;
  for (int i = 0; i < 3; i++) { const double adj = v[i] / sum * 256.0 + err; const int q = (int)std::lrint(adj); err = adj - q; k[i] = k[6 - i] = q; s += q; }
  k[3] = 256 - 2 * s;
}

inline int reflect101(int p, int n) { while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p; return p; }

/* exact 32-bit sums of the separable 7 x 7 correlation under reflect-101 (integer arithmetic: no compiler flag can change them) */
inline void gauss_acc(const uint8_t* src, int w, int h, int kernel, std::vector<uint32_t>& acc) {
  int k[7];
  gauss_weights(kernel, k);
  std::vector<uint32_t> hor((size_t)w * h);
  for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {
    uint32_t a = 0;
    for (int t = 0; t < 7; t++) a += (uint32_t)k[t] * src[(size_t)y * w + reflect101(x + t - 3, w)];
    hor[(size_t)y * w + x] = a;
  }
  acc.assign((size_t)w * h, 0);
  for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {
    uint32_t a = 0;
    for (int t = 0; t < 7; t++) a += (uint32_t)k[t] * hor[(size_t)reflect101(y + t - 3, h) * w + x];
    acc[(size_t)y * w + x] = a;
  }
}
inline uint8_t gauss_round_px(uint32_t acc, int mode) {
  uint32_t r = mode == 2 ? acc >> 16 : (acc + 32768u) >> 16;
  if (mode == 1 && (acc & 0xffffu) == 0x8000u) r &= ~1u;
  return (uint8_t)(r > 255u ? 255u : r);
}

/* separate IEEE operations whatever the compiler's contraction setting: every intermediate goes through a volatile */
inline float atan_variant(float y, float x, int fma) {
  const float sc = (float)(180.0 / 3.14159265358979323846);
  const float p1 = 0.9997878412794807f * sc, p3 = -0.3258083974640975f * sc, p5 = 0.1555786518463281f * sc, p7 = -0.04432655554792128f * sc;
  const float ax = std::fabs(x), ay = std::fabs(y), eps = (float)2.2204460492503131e-16;
  volatile float t, c, c2, a;
  if (ax >= ay) {
    t = ax + eps; c = ay / t; c2 = c * c;
    if (fma) { t = std::fma(p7, (float)c2, p5); t = std::fma((float)t, (float)c2, p3); t = std::fma((float)t, (float)c2, p1); a = t * c; }
    else { t = p7 * c2; t = t + p5; t = t * c2; t = t + p3; t = t * c2; t = t + p1; a = t * c; }
  } else {
    t = ay + eps; c = ax / t; c2 = c * c;
    if (fma) { t = std::fma(p7, (float)c2, p5); t = std::fma((float)t, (float)c2, p3); t = std::fma((float)t, (float)c2, p1); a = std::fma(-(float)t, (float)c, 90.f); }
    else { t = p7 * c2; t = t + p5; t = t * c2; t = t + p3; t = t * c2; t = t + p1; t = t * c; a = 90.f - t; }
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

/* (3): does THIS translation unit (= the flags src/ORBextractor.cc would have been compiled with) contract the pattern rotation of
 * src/ORBextractor.cc:118-120, and HOW?  Two expressions, x*b + y*a and x*a - y*b, each with two ways to fuse: the FIRST product into the
 * FMA (form A: fma(x, b, y*a), fma(x, a, -(y*b)) — what GCC and clang emit, and what liborbx's brief_fma = 1 reproduces) or the SECOND
 * (form B: fma(y, a, x*b), fma(-y, b, x*a)).  Operand sets at which exactly one form rounds to another integer than the unfused
 * expression (found by a sweep over 4e8 sets of the pattern's coordinate range; both forms never differ on one set) tell them apart.
 * The probes are `static`: every translation unit that includes this header gets ITS OWN copy, compiled with ITS flags (an `inline`
 * function would be merged by the linker into one copy of unknown flags). */
#if defined(__GNUC__)
#define ORBX_CV_NOINLINE __attribute__((noinline, unused))
#else
#define ORBX_CV_NOINLINE
#endif
static ORBX_CV_NOINLINE int contraction_probe_sum(const int* p, const float* ab) { return (int)std::lrint(p[0] * ab[1] + p[1] * ab[0]); }
static ORBX_CV_NOINLINE int contraction_probe_diff(const int* p, const float* ab) { return (int)std::lrint(p[0] * ab[0] - p[1] * ab[1]); }
/* 0 = not contracted, 1 = form A on both expressions (brief_fma = 1 reproduces it), -1 = anything else (form B, a mix, neither): no
 * option of liborbx reproduces that build */
inline int build_contraction_form() {
  struct Op { int expr; int32_t x, y; uint32_t a, b; int unfused, formA, formB; };
  static const Op kOps[] = {
      {0, -12, 10, 0x3e2ad13du, 0xbf7c69bdu, 14, 13, 14}, {0, 9, 4, 0xbed644d7u, 0x3f688114u, 6, 7, 6}, {0, 12, 3, 0x3f50bff3u, 0xbf142ffdu, -5, -4, -5},
      {0, -8, 10, 0x3f6f411cu, 0x3eb622c6u, 7, 7, 6}, {0, 13, 13, 0xbe5d1242u, 0xbf79f683u, -16, -16, -15}, {0, 13, -10, 0xbf585621u, 0xbf08dfcbu, 2, 2, 1},
      {1, 10, -4, 0x3f528e6cu, 0x3f119bf3u, 11, 10, 11}, {1, -11, 12, 0xbf7e31f2u, 0x3df2c39cu, 9, 10, 9}, {1, -10, 12, 0x3f7e2dc2u, 0xbdf3dbb1u, -8, -9, -8},
      {1, 13, 10, 0xbdf7e3f9u, 0x3f7e1e27u, -11, -11, -12}, {1, -4, -10, 0xbf119beeu, 0x3f528e6fu, 10, 10, 11}, {1, 12, 7, 0x3f755488u, 0x3e924667u, 10, 10, 9}};
  int is_u = 0, is_a = 0, is_b = 0, other = 0;
  for (const Op& o : kOps) {
    volatile int32_t vx = o.x, vy = o.y; volatile uint32_t va = o.a, vb = o.b;   // opaque to constant folding
    const int p[2] = {vx, vy};
    const uint32_t ua = va, ub = vb;
    float ab[2];
    std::memcpy(&ab[0], &ua, 4); std::memcpy(&ab[1], &ub, 4);
    const int got = o.expr == 0 ? contraction_probe_sum(p, ab) : contraction_probe_diff(p, ab);
    // every set separates exactly one form from the other two
    if (o.formA != o.unfused) { if (got == o.formA) is_a++; else if (got == o.unfused) { is_u++; is_b++; } else other++; }
    else { if (got == o.formB) is_b++; else if (got == o.unfused) { is_u++; is_a++; } else other++; }
  }
  const int n = (int)(sizeof(kOps) / sizeof(kOps[0]));
  if (other) return -1;
  if (is_u == n) return 0;            // unfused everywhere (then is_a + is_b == n as well: check it first)
  if (is_a == n) return 1;
  return -1;                          // form B, or the two expressions contracted differently
}
inline int build_contracts_fma() { return build_contraction_form() == 1 ? 1 : 0; }

/* (4): the tie order of std::sort.  DistributeOctTree sorts its (point count, UL.x) pairs with std::sort (src/ORBextractor.cc:700); equal
 * keys are the rule there (stacked nodes share UL.x, small counts repeat), and the order in which they come out is the STANDARD LIBRARY's:
 * it decides the order of the keypoints and, where a tie sits at the node at which the quota is reached, the keypoints themselves.
 * liborbx restates libstdc++'s introsort (csrc/gnu_sort.h).  The probe sorts three keyed id sequences (40 / 100 / 300 elements, 5 - 9
 * distinct keys) with the std::sort of the caller's toolchain — the one src/ORBextractor.cc would be built with — and compares a digest of
 * the permutations with libstdc++'s.  There is no option behind it: another library (libc++, MSVC) means another tie order. */
template <class Sorter>
inline uint64_t sort_probe_digest(Sorter sorter) {
  uint64_t h = 1469598103934665603ull;
  static const int sizes[3] = {40, 100, 300}, nkeys[3] = {5, 7, 9};
  uint32_t s = 0x2545F491u;
  for (int p = 0; p < 3; p++) {
    std::vector<std::pair<int, int> > v((size_t)sizes[p]);
    for (int i = 0; i < sizes[p]; i++) { s = s * 1664525u + 1013904223u; v[(size_t)i] = std::make_pair((int)((s >> 10) % (uint32_t)nkeys[p]), i); }
    sorter(v);
    for (const std::pair<int, int>& e : v) { h ^= (uint64_t)(uint32_t)e.second; h *= 1099511628211ull; }
  }
  return h;
}
constexpr uint64_t kLibstdcxxSortDigest = 0x462bd45918e0b9e3ull;
inline bool std_sort_is_libstdcxx() {
  return sort_probe_digest([](std::vector<std::pair<int, int> >& v) {
           std::sort(v.begin(), v.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.first < b.first; });
         }) == kLibstdcxxSortDigest;
}

/* Blur: void(const uint8_t* src, int w, int h, uint8_t* dst) — the OpenCV at hand; Atan: float(float y, float x) */
template <class Blur, class Atan>
inline Calibration calibrate(Blur blur, Atan at) {
  Calibration c;
  std::vector<uint8_t> img, got((size_t)kProbeW * kProbeH);
  probe_image(img);
  blur(img.data(), kProbeW, kProbeH, got.data());
  int best = 1 << 30;
  static const int tails[6] = {0, 4, 8, 16, 32, 64};
  for (int k = 0; k < 2; k++) {
    std::vector<uint32_t> acc;
    gauss_acc(img.data(), kProbeW, kProbeH, k, acc);
    for (int r = 0; r < 3; r++) for (int ti = 0; ti < (r == 0 ? 1 : 6); ti++) {
      const int V = tails[ti], body = V > 1 ? kProbeW - kProbeW % V : kProbeW;
      int bad = 0;
      for (int y = 0; y < kProbeH; y++) for (int x = 0; x < kProbeW; x++)
        bad += gauss_round_px(acc[(size_t)y * kProbeW + x], x < body ? r : 0) != got[(size_t)y * kProbeW + x];
      if (bad == 0) c.gauss_candidates++;
      if (bad < best) { best = bad; c.gauss_kernel = k; c.gauss_round = r; c.gauss_tail = V; }
    }
  }
  c.gauss_exact = best == 0; c.gauss_mismatch = best;
  if (!c.gauss_exact) c.gauss_kernel = c.gauss_round = c.gauss_tail = 0;
  int bad[2] = {0, 0};
  uint32_t s = 12345u;
  for (int i = 0; i < 4096; i++) {
    s = s * 1664525u + 1013904223u; const int m01 = (int)((s >> 8) % 6000001u) - 3000000;
    s = s * 1664525u + 1013904223u; const int m10 = (i & 15) == 0 ? 0 : (int)((s >> 8) % 6000001u) - 3000000;
    const float g = at((float)m01, (float)m10);
    for (int f = 0; f < 2; f++) { const float e = atan_variant((float)m01, (float)m10, f); bad[f] += std::memcmp(&g, &e, 4) != 0; }
  }
  c.atan_fma = bad[1] < bad[0] ? 1 : 0;
  c.atan_mismatch = bad[c.atan_fma]; c.atan_exact = c.atan_mismatch == 0;
  if (!c.atan_exact) c.atan_fma = 0;
  // second probe: the variant found on the small image must also reproduce a FRAME-sized one (OpenCV paths that switch on the image size —
  // IPP, OpenCL, the stripes of parallel_for_ — do not show on 127 x 72)
  if (c.gauss_exact) {
    const int W = kFrameProbeW, H = kFrameProbeH;
    std::vector<uint8_t> big((size_t)W * H), out((size_t)W * H);
    uint32_t s2 = 0x85EBCA6Bu;
    for (size_t i = 0; i < big.size(); i++) { s2 ^= s2 << 13; s2 ^= s2 >> 17; s2 ^= s2 << 5; big[i] = (uint8_t)(s2 >> 9); }
    for (int y = 200; y < 207; y++) std::memset(&big[(size_t)y * W], (y - 200) == 0 ? 127 : (y - 200) == 3 ? 126 : 128, (size_t)W);   // a tie band
    for (int y = 300; y < 312; y++) std::memset(&big[(size_t)y * W], 255, (size_t)W);                                                   // saturation
    blur(big.data(), W, H, out.data());
    std::vector<uint32_t> acc;
    gauss_acc(big.data(), W, H, c.gauss_kernel, acc);
    const int V = c.gauss_tail, body = V > 1 ? W - W % V : W;
    int bad = 0;
    for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) bad += gauss_round_px(acc[(size_t)y * W + x], x < body ? c.gauss_round : 0) != out[(size_t)y * W + x];
    c.frame_w = W; c.frame_h = H; c.frame_mismatch = bad;
  }
  c.brief_form = build_contraction_form();
  c.brief_fma = c.brief_form == 1 ? 1 : 0;
  c.sort_libstdcxx = std_sort_is_libstdcxx();
  return c;
}

inline bool env_overrides() {
  static const char* names[] = {"ORBX_GAUSS_KERNEL", "ORBX_GAUSS_ROUND", "ORBX_GAUSS_TAIL", "ORBX_ATAN_FMA", "ORBX_BRIEF_FMA"};
  for (const char* n : names) if (std::getenv(n)) return true;
  const char* e = std::getenv("ORBX_CV_CALIBRATE");
  return e && std::atoi(e) == 0;
}

/* applies a calibration to a context; returns ORBX_OK or the first failing orbx_set_option code */
// "No variant reproduces this OpenCV" must not pass as a drop-in: the bits the reference's CPU build would produce over THIS OpenCV are then
// not the bits liborbx computes, and nothing downstream would ever say so.  pinned() is what include/ORBextractor.h's constructor demands;
// ORBX_ALLOW_UNPINNED=1 in the environment turns the refusal back into the stderr report (a maintainer who accepts a documented difference).
inline bool pinned(const Calibration& c) { return c.gauss_exact && c.atan_exact && c.frame_mismatch <= 0 && c.brief_form >= 0 && c.sort_libstdcxx; }
inline bool allow_unpinned() { const char* e = std::getenv("ORBX_ALLOW_UNPINNED"); return e && std::atoi(e) != 0; }
inline std::string why_unpinned(const Calibration& c) {
  std::string w;
  if (!c.gauss_exact) w += "cv::GaussianBlur(7x7, sigma 2, 8u) matches no known variant (closest differs in " + std::to_string(c.gauss_mismatch) + " probe bytes); ";
  if (c.frame_mismatch > 0) w += "cv::GaussianBlur changes its arithmetic with the image size (" + std::to_string(c.frame_mismatch) + " bytes of a frame differ); ";
  if (!c.atan_exact) w += "cv::fastAtan2 matches neither form (" + std::to_string(c.atan_mismatch) + " of 4096 angles differ); ";
  if (c.brief_form < 0) w += "this build contracts the pattern rotation in a form liborbx has no option for; ";
  if (!c.sort_libstdcxx) w += "std::sort of this toolchain does not leave equal keys in libstdc++'s order; ";
  return w;
}

inline int apply(orbx_ctx* ctx, const Calibration& c) {
  if (env_overrides()) return ORBX_OK;
  int rc = orbx_set_option(ctx, "gauss_kernel", c.gauss_kernel);
  if (rc == ORBX_OK) rc = orbx_set_option(ctx, "gauss_round", c.gauss_round);
  if (rc == ORBX_OK) rc = orbx_set_option(ctx, "gauss_tail", c.gauss_tail);
  if (rc == ORBX_OK) rc = orbx_set_option(ctx, "atan_fma", c.atan_fma);
  if (rc == ORBX_OK) rc = orbx_set_option(ctx, "brief_fma", c.brief_fma);
  return rc;
}

inline void report(const Calibration& c, const char* opencv_version, std::FILE* f = stderr) {
  const bool dflt = c.gauss_kernel == 0 && c.gauss_round == 0 && c.gauss_tail == 0 && c.atan_fma == 0 && c.brief_fma == 0;
  if (!c.sort_libstdcxx)
    std::fprintf(f, "[orbx] std::sort of this toolchain does not leave equal keys in libstdc++'s order: DistributeOctTree (src/ORBextractor.cc:700) "
                    "built with it orders tied nodes differently from liborbx — keypoint order (and some keypoints) will differ from a CPU build here\n");
  if (c.brief_form < 0)
    std::fprintf(f, "[orbx] this build contracts the pattern rotation (src/ORBextractor.cc:118-120) in a form liborbx has no option for (not fma(x,b,y*a) / "
                    "fma(x,a,-(y*b))): about one rotated pattern point in ten million will differ from a CPU build with these flags; brief_fma stays 0\n");
  if (c.frame_mismatch > 0)
    std::fprintf(f, "[orbx] cv::GaussianBlur equals variant gauss_kernel=%d gauss_round=%d gauss_tail=%d on the %d x %d probe but differs in %d bytes of a %d x %d "
                    "frame: this OpenCV switches paths with the image size (IPP / OpenCL / threading); descriptors will not be bit-identical — see "
                    "tools/validate_opencv.cpp\n", c.gauss_kernel, c.gauss_round, c.gauss_tail, kProbeW, kProbeH, c.frame_mismatch, c.frame_w, c.frame_h);
  if (c.gauss_exact && c.gauss_candidates > 1)
    std::fprintf(f, "[orbx] %d Gaussian variants reproduce this OpenCV on the probe image (the probe no longer separates them): taking gauss_kernel=%d "
                    "gauss_round=%d gauss_tail=%d, the first\n", c.gauss_candidates, c.gauss_kernel, c.gauss_round, c.gauss_tail);
  if (c.gauss_exact && c.atan_exact && dflt && c.brief_form >= 0 && c.frame_mismatch <= 0 && !std::getenv("ORBX_CV_VERBOSE")) return;
  std::fprintf(f, "[orbx] OpenCV %s: cv::GaussianBlur(7x7, sigma 2, 8u) %s gauss_kernel=%d gauss_round=%d gauss_tail=%d", opencv_version,
               c.gauss_exact ? "== variant" : "matches NO known variant; keeping", c.gauss_kernel, c.gauss_round, c.gauss_tail);
  if (!c.gauss_exact) std::fprintf(f, " (closest variant differs in %d of %d probe bytes: descriptors will not be bit-identical to this OpenCV's; see tools/validate_opencv.cpp)", c.gauss_mismatch, kProbeW * kProbeH);
  std::fprintf(f, "; cv::fastAtan2 %s atan_fma=%d", c.atan_exact ? "==" : "matches neither form; keeping", c.atan_fma);
  if (!c.atan_exact) std::fprintf(f, " (%d of 4096 angles differ)", c.atan_mismatch);
  std::fprintf(f, "; this build %s the pattern rotation: brief_fma=%d; std::sort tie order %s\n", c.brief_fma ? "contracts" : "does not contract", c.brief_fma,
               c.sort_libstdcxx ? "= libstdc++" : "NOT libstdc++");
}

}  // namespace orbx_cv

/* With OpenCV's imgproc in sight: the calibration against it, once per process. */
#if defined(__has_include)
#if __has_include(<opencv2/imgproc/imgproc.hpp>) && __has_include(<opencv2/core/core.hpp>) && !defined(ORBX_FORCE_CV_COMPAT) && !defined(ORBX_NO_CV_CALIBRATION)
#include <opencv2/core/core.hpp>
#include <opencv2/imgproc/imgproc.hpp>
#define ORBX_CV_CALIBRATION 1
namespace orbx_cv {
inline const Calibration& opencv_calibration() {
  static const Calibration cal = [] {
    Calibration c = calibrate(
        [](const uint8_t* src, int w, int h, uint8_t* dst) {
          cv::Mat s(h, w, CV_8UC1, (void*)src, (size_t)w), d;
          cv::Mat work = s.clone();   // as src/ORBextractor.cc:1132-1133: in place on a continuous clone
          cv::GaussianBlur(work, work, cv::Size(7, 7), 2, 2, cv::BORDER_REFLECT_101);
          for (int y = 0; y < h; y++) std::memcpy(dst + (size_t)y * w, work.ptr<unsigned char>(y), (size_t)w);
        },
        [](float y, float x) { return cv::fastAtan2(y, x); });
#ifdef CV_VERSION
    report(c, CV_VERSION);
#else
    report(c, "(version macro absent)");
#endif
    return c;
  }();
  return cal;
}
}  // namespace orbx_cv
#endif
#endif

#endif /* ORBX_CV_CALIBRATE_H */
