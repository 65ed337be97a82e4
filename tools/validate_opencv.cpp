// validate_opencv — for a maintainer who HAS OpenCV: is the OpenCV I build ORB_SLAM3 with the one orbx's results equal?
//
// "Keypoints and descriptors bit-exact against the reference CPU path" rests on the OpenCV primitives src/ORBextractor.cc and src/Frame.cc
// call.  orbx restates them (this repository's container has no OpenCV: oracle/orb_oracle.cpp states the arithmetic, liborbx.so computes
// the same on the GPU) and everything else is checked against the reference's own compiled code.  This program closes the remaining gap
// on the maintainer's machine.  Built against REAL OpenCV (tools/validate_opencv.cmake) it
//   1. prints the OpenCV version and what include/orbx_cv_calibrate.h detects for the build-dependent primitives (cv::GaussianBlur 8u,
//      cv::fastAtan2, FMA contraction of this build) — the options include/ORBextractor.h will apply by itself;
//   2. runs every primitive of the path on a set of images (tools/make_validate_set.py: the natural crops of tests/golden + synthetic
//      frames; without a set, on built-in synthetic images) and compares, bit for bit, with the oracle's restatement
//      (oracle/liborb_oracle.so) under the detected variant:
//        cv::resize INTER_LINEAR down the 8-level pyramid (src/ORBextractor.cc:1183), cv::copyMakeBorder REFLECT_101 (:1185-1191),
//        cv::FAST 9/16 with NMS at thresholds 20 and 7 on every level (:826,:845), cv::GaussianBlur (:1133), cv::fastAtan2 (:102),
//        and with -DORBX_VALIDATE_EXTRAS cv::undistortPoints (src/Frame.cc:766) and cv::cvtColor (src/Tracking.cc:1572-1585);
//      for each it prints MATCH, or the first mismatch (image, level, position, both values) and the number of differing elements;
//   3. with --orbx (needs an MI355X and liborbx.so) extracts every image through the C ABI with the detected options and, when built with
//      -DORBX_VALIDATE_REFERENCE (the reference's src/ORBextractor.cc compiled into this program), compares keypoints, descriptors and
//      the return value with the reference's operator() over the maintainer's OpenCV.
// Exit code 0 = everything compared equal.  In this repository's container it is built against the shim of oracle/ref_shims (whose cv::
// functions ARE the oracle's), which proves the harness, not OpenCV: oracle/ref_fragments.mk -> oracle/_ref/validate_opencv,
// tests/test_validate_opencv.py.
//
//   4. with --expect tools/opencv_pin/expected_digests.txt: prints a 64-bit digest of everything the OpenCV at hand computed per primitive on
//      the set and says which NAMED profile (orbx_set_cpu_profile) carries that digest — the table was made by
//      running this program over the oracle under every profile (tools/opencv_pin/make_expected.py), so it needs no oracle at run time to
//      answer "which CPU path is mine"; --print-digests writes the table lines of the current run.
//
//   validate_opencv [--set validate_set.bin] [--orbx] [--nfeatures 1000] [--verbose] [--expect table.txt] [--print-digests]
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#include <opencv2/imgproc/imgproc.hpp>
#ifdef ORBX_VALIDATE_EXTRAS
#include <opencv2/calib3d/calib3d.hpp>
#endif

#include "orbx.h"
#include "orbx_cv_calibrate.h"

#ifdef ORBX_VALIDATE_REFERENCE
#include "ORBextractor.h"   // the REFERENCE's header (its include directory comes first on the include path of this build)
#endif

extern "C" {   // oracle/liborb_oracle.so — the restatement under test
void orbo_resize_linear(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride);
int orbo_fast(const uint8_t* img, int cols, int rows, int stride, int threshold, int nms, void* dst, int cap);
void orbo_gaussian_blur7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride);
float orbo_fast_atan2(float y, float x);
int orbo_set_gauss_variant(int kernel, int round);
int orbo_set_gauss_tail(int v);
int orbo_set_atan_fma(int on);
int orbo_set_brief_fma(int on);
#ifdef ORBX_VALIDATE_EXTRAS
void mo_undistort_points(const float* xy_in, int n, float fx, float fy, float cx, float cy, const float* dist, int ndist, float* xy_out);
#endif
}

namespace {

struct Img { std::string name; int rows = 0, cols = 0; std::vector<uint8_t> px; };

bool load_set(const char* path, std::vector<Img>& out) {
  FILE* f = std::fopen(path, "rb");
  if (!f) return false;
  char magic[8];
  int32_t n = 0;
  bool ok = std::fread(magic, 1, 8, f) == 8 && std::memcmp(magic, "ORBXVS01", 8) == 0 && std::fread(&n, 4, 1, f) == 1 && n >= 0 && n < 100000;
  for (int i = 0; ok && i < n; i++) {
    int32_t hdr[3];   // rows, cols, name length
    Img im;
    ok = std::fread(hdr, 4, 3, f) == 3 && hdr[0] > 0 && hdr[1] > 0 && hdr[2] >= 0 && hdr[2] < 256;
    if (!ok) break;
    im.rows = hdr[0]; im.cols = hdr[1]; im.name.resize(hdr[2]); im.px.resize((size_t)im.rows * im.cols);
    ok = (hdr[2] == 0 || std::fread(&im.name[0], 1, hdr[2], f) == (size_t)hdr[2]) && std::fread(im.px.data(), 1, im.px.size(), f) == im.px.size();
    if (ok) out.push_back(std::move(im));
  }
  std::fclose(f);
  return ok;
}

void builtin_set(std::vector<Img>& out) {   // value noise + rectangles + flat areas + saturated patches, three shapes
  const int shapes[3][2] = {{480, 640}, {480, 752}, {350, 600}};
  uint32_t s = 20260925u;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; };
  for (int k = 0; k < 3; k++) {
    Img im;
    im.name = "builtin" + std::to_string(k); im.rows = shapes[k][0]; im.cols = shapes[k][1]; im.px.resize((size_t)im.rows * im.cols);
    for (int y = 0; y < im.rows; y++) for (int x = 0; x < im.cols; x++) {
      const int base = 96 + (int)(40 * std::sin(x * 0.031 + k) + 40 * std::cos(y * 0.043));
      im.px[(size_t)y * im.cols + x] = (uint8_t)std::min(255, std::max(0, base + (int)(rnd() % 9) - 4));
    }
    for (int r = 0; r < 220; r++) {
      const int w = 8 + rnd() % 72, h = 8 + rnd() % 72, x0 = rnd() % (im.cols - w), y0 = rnd() % (im.rows - h), v = rnd() % 256;
      for (int y = y0; y < y0 + h; y++) std::memset(&im.px[(size_t)y * im.cols + x0], v, w);
    }
    for (int y = im.rows / 2; y < im.rows; y++) for (int x = 0; x < im.cols / 4; x++) im.px[(size_t)y * im.cols + x] = (uint8_t)(120 + rnd() % 3);
    out.push_back(std::move(im));
  }
}

struct Tally { long compared = 0, differing = 0; int fails = 0; };
struct Fnv64 {
  uint64_t h = 1469598103934665603ull;
  void add(const void* p, size_t n) { const unsigned char* b = (const unsigned char*)p; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } }
};
void verdict(const char* what, const Tally& t, const std::string& first) {
  if (t.differing == 0) std::printf("  %-58s MATCH   (%ld elements)\n", what, t.compared);
  else std::printf("  %-58s MISMATCH %ld of %ld elements; first: %s\n", what, t.differing, t.compared, first.c_str());
}

// src/ORBextractor.cc:414-430,1174-1175: level sizes of the 8-level, 1.2 pyramid
void level_sizes(int rows, int cols, int nlevels, float sf, std::vector<int>& w, std::vector<int>& h) {
  std::vector<float> scale(nlevels), inv(nlevels);
  scale[0] = 1.f;
  for (int i = 1; i < nlevels; i++) scale[i] = (float)(scale[i - 1] * (double)sf);
  for (int i = 0; i < nlevels; i++) inv[i] = 1.0f / scale[i];
  w.resize(nlevels); h.resize(nlevels);
  for (int l = 0; l < nlevels; l++) { w[l] = cvRound((float)cols * inv[l]); h[l] = cvRound((float)rows * inv[l]); }
}

}  // namespace

int main(int argc, char** argv) {
  const char* set_path = nullptr;
  const char* expect_path = nullptr;
  bool with_orbx = false, verbose = false, print_digests = false;
  int nfeatures = 1000;
  for (int i = 1; i < argc; i++) {
    const std::string a = argv[i];
    if (a == "--set" && i + 1 < argc) set_path = argv[++i];
    else if (a == "--orbx") with_orbx = true;
    else if (a == "--verbose") verbose = true;
    else if (a == "--expect" && i + 1 < argc) expect_path = argv[++i];
    else if (a == "--print-digests") print_digests = true;
    else if (a == "--nfeatures" && i + 1 < argc) nfeatures = std::atoi(argv[++i]);
    else { std::fprintf(stderr, "usage: validate_opencv [--set validate_set.bin] [--orbx] [--nfeatures N] [--verbose] [--expect table.txt] [--print-digests]\n"); return 2; }
  }
#ifdef CV_VERSION
  std::printf("OpenCV %s\n", CV_VERSION);
#else
  std::printf("OpenCV: no CV_VERSION macro (the container shim of oracle/ref_shims: this run proves the harness only)\n");
#endif
  // ---- 1. what the adapter's calibration sees
  const orbx_cv::Calibration& cal = orbx_cv::opencv_calibration();
  std::printf("calibration: gauss_kernel=%d gauss_round=%d gauss_tail=%d (%s, %d candidate(s)%s)  atan_fma=%d (%s)  brief_fma=%d (this build %s x*b + y*a)\n",
              cal.gauss_kernel, cal.gauss_round, cal.gauss_tail, cal.gauss_exact ? "exact" : "NO VARIANT MATCHES", cal.gauss_candidates,
              cal.gauss_exact ? "" : (", closest differs in " + std::to_string(cal.gauss_mismatch) + " probe bytes").c_str(), cal.atan_fma,
              cal.atan_exact ? "exact" : "NEITHER FORM MATCHES", cal.brief_fma, cal.brief_fma ? "contracts" : "does not contract");
  std::printf("  -> orbx_set_option(ctx, \"gauss_kernel\", %d); (\"gauss_round\", %d); (\"gauss_tail\", %d); (\"atan_fma\", %d); (\"brief_fma\", %d)   [include/ORBextractor.h does this itself]\n",
              cal.gauss_kernel, cal.gauss_round, cal.gauss_tail, cal.atan_fma, cal.brief_fma);
  std::printf("std::sort of this toolchain (tie order of DistributeOctTree's node vector, src/ORBextractor.cc:700): %s\n",
              cal.sort_libstdcxx ? "libstdc++'s — the order liborbx restates" : "NOT libstdc++'s: keypoint order WILL differ from liborbx (no option exists for this)");
  orbo_set_gauss_variant(cal.gauss_kernel, cal.gauss_round); orbo_set_gauss_tail(cal.gauss_tail); orbo_set_atan_fma(cal.atan_fma); orbo_set_brief_fma(cal.brief_fma);
  int failures = (cal.gauss_exact ? 0 : 1) + (cal.atan_exact ? 0 : 1) + (cal.sort_libstdcxx ? 0 : 1);

  std::vector<Img> set;
  if (set_path) { if (!load_set(set_path, set)) { std::fprintf(stderr, "cannot read %s\n", set_path); return 2; } }
  else builtin_set(set);
  std::printf("%zu images (%s)\n", set.size(), set_path ? set_path : "built-in synthetic set; use tools/make_validate_set.py for the natural crops");

  // ---- 2. the primitives
  Fnv64 d_set, d_resize, d_fast20, d_fast7, d_blur, d_atan;   // what the OpenCV at hand computed, primitive by primitive
  for (const Img& im : set) { d_set.add(&im.rows, 4); d_set.add(&im.cols, 4); d_set.add(im.px.data(), im.px.size()); }
  Tally t_resize, t_border, t_fast20, t_fast7, t_blur, t_atan;
  std::string f_resize, f_border, f_fast20, f_fast7, f_blur, f_atan;
  const int nlevels = 8;
  for (const Img& im : set) {
    std::vector<int> lw, lh;
    level_sizes(im.rows, im.cols, nlevels, 1.2f, lw, lh);
    cv::Mat prev_cv(im.rows, im.cols, CV_8UC1, (void*)im.px.data(), (size_t)im.cols);
    prev_cv = prev_cv.clone();
    std::vector<uint8_t> prev_or(im.px);
    for (int l = 0; l < nlevels; l++) {
      if (lw[l] < 40 || lh[l] < 40) break;
      cv::Mat cur_cv;
      std::vector<uint8_t> cur_or;
      if (l == 0) { cur_cv = prev_cv; cur_or = prev_or; }
      else {
        cv::resize(prev_cv, cur_cv, cv::Size(lw[l], lh[l]), 0, 0, cv::INTER_LINEAR);
        for (int y = 0; y < lh[l]; y++) d_resize.add(cur_cv.ptr<unsigned char>(y), (size_t)lw[l]);
        cur_or.resize((size_t)lw[l] * lh[l]);
        orbo_resize_linear(prev_or.data(), lw[l - 1], lh[l - 1], lw[l - 1], cur_or.data(), lw[l], lh[l], lw[l]);
        for (int y = 0; y < lh[l]; y++) for (int x = 0; x < lw[l]; x++) {
          t_resize.compared++;
          const int a = cur_cv.ptr<unsigned char>(y)[x], b = cur_or[(size_t)y * lw[l] + x];
          if (a != b && !t_resize.differing++) f_resize = im.name + " level " + std::to_string(l) + " (" + std::to_string(x) + "," + std::to_string(y) + "): cv " + std::to_string(a) + " oracle " + std::to_string(b);
        }
        // the OpenCV side continues from ITS OWN level (as the reference does); the oracle side from its own: a mismatch propagates, as it would in the product
      }
      // copyMakeBorder, REFLECT_101, 19 px (EDGE_THRESHOLD): idx(-k) = k, idx(n-1+k) = n-1-k
      {
        cv::Mat padded;
        cv::copyMakeBorder(cur_cv, padded, 19, 19, 19, 19, cv::BORDER_REFLECT_101);
        auto refl = [](int i, int n) { while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i; return i; };
        for (int y = 0; y < padded.rows; y++) for (int x = 0; x < padded.cols; x++) {
          t_border.compared++;
          const int a = padded.ptr<unsigned char>(y)[x], b = cur_cv.ptr<unsigned char>(refl(y - 19, cur_cv.rows))[refl(x - 19, cur_cv.cols)];
          if (a != b && !t_border.differing++) f_border = im.name + " level " + std::to_string(l) + " padded (" + std::to_string(x) + "," + std::to_string(y) + ")";
        }
      }
      // FAST at both thresholds: positions, responses, order
      for (int th : {20, 7}) {
        Tally& T = th == 20 ? t_fast20 : t_fast7;
        std::string& F = th == 20 ? f_fast20 : f_fast7;
        std::vector<cv::KeyPoint> kc;
        cv::FAST(cur_cv, kc, th, true);
        for (const cv::KeyPoint& k : kc) { const float r3[3] = {k.pt.x, k.pt.y, k.response}; (th == 20 ? d_fast20 : d_fast7).add(r3, sizeof r3); }
        std::vector<cv::KeyPoint> ko((size_t)lw[l] * lh[l] + 1);
        // the oracle side runs on the OpenCV side's level so that FAST is judged on identical pixels
        std::vector<uint8_t> same((size_t)lw[l] * lh[l]);
        for (int y = 0; y < lh[l]; y++) std::memcpy(&same[(size_t)y * lw[l]], cur_cv.ptr<unsigned char>(y), lw[l]);
        const int no = orbo_fast(same.data(), lw[l], lh[l], lw[l], th, 1, ko.data(), (int)ko.size());
        T.compared += (long)std::max((size_t)no, kc.size());
        if ((size_t)no != kc.size()) { if (!T.differing) F = im.name + " level " + std::to_string(l) + ": cv " + std::to_string(kc.size()) + " keypoints, oracle " + std::to_string(no); T.differing += std::labs((long)no - (long)kc.size()); }
        for (size_t i = 0; i < std::min((size_t)std::max(no, 0), kc.size()); i++)
          if (kc[i].pt.x != ko[i].pt.x || kc[i].pt.y != ko[i].pt.y || kc[i].response != ko[i].response) {
            if (!T.differing++) F = im.name + " level " + std::to_string(l) + " keypoint " + std::to_string(i) + ": cv (" + std::to_string((int)kc[i].pt.x) + "," + std::to_string((int)kc[i].pt.y) + ") r " + std::to_string((int)kc[i].response) + ", oracle (" + std::to_string((int)ko[i].pt.x) + "," + std::to_string((int)ko[i].pt.y) + ") r " + std::to_string((int)ko[i].response);
          }
      }
      // GaussianBlur as the reference calls it: in place on a continuous clone
      {
        cv::Mat work = cur_cv.clone();
        cv::GaussianBlur(work, work, cv::Size(7, 7), 2, 2, cv::BORDER_REFLECT_101);
        for (int y = 0; y < lh[l]; y++) d_blur.add(work.ptr<unsigned char>(y), (size_t)lw[l]);
        std::vector<uint8_t> same((size_t)lw[l] * lh[l]), bo((size_t)lw[l] * lh[l]);
        for (int y = 0; y < lh[l]; y++) std::memcpy(&same[(size_t)y * lw[l]], cur_cv.ptr<unsigned char>(y), lw[l]);
        orbo_gaussian_blur7(same.data(), lw[l], lh[l], lw[l], bo.data(), lw[l]);
        for (int y = 0; y < lh[l]; y++) for (int x = 0; x < lw[l]; x++) {
          t_blur.compared++;
          const int a = work.ptr<unsigned char>(y)[x], b = bo[(size_t)y * lw[l] + x];
          if (a != b && !t_blur.differing++) f_blur = im.name + " level " + std::to_string(l) + " (" + std::to_string(x) + "," + std::to_string(y) + "): cv " + std::to_string(a) + " oracle " + std::to_string(b);
        }
      }
      prev_cv = cur_cv; prev_or.swap(cur_or);
    }
  }
  {
    uint32_t s = 99u;
    for (int i = 0; i < 2000000; i++) {
      s = s * 1664525u + 1013904223u; const int m01 = (int)((s >> 8) % 6000001u) - 3000000;
      s = s * 1664525u + 1013904223u; const int m10 = (i & 31) == 0 ? 0 : (int)((s >> 8) % 6000001u) - 3000000;
      const float a = cv::fastAtan2((float)m01, (float)m10), b = orbo_fast_atan2((float)m01, (float)m10);
      t_atan.compared++;
      d_atan.add(&a, 4);
      if (std::memcmp(&a, &b, 4) != 0 && !t_atan.differing++) f_atan = "fastAtan2(" + std::to_string(m01) + ", " + std::to_string(m10) + "): cv " + std::to_string(a) + " oracle " + std::to_string(b);
    }
  }
  std::printf("primitives, OpenCV at hand vs oracle/orb_oracle.cpp:\n");
  verdict("cv::resize INTER_LINEAR, 8-level pyramid", t_resize, f_resize);
  verdict("cv::copyMakeBorder BORDER_REFLECT_101, 19 px", t_border, f_border);
  verdict("cv::FAST(threshold 20, NMS): positions, responses, order", t_fast20, f_fast20);
  verdict("cv::FAST(threshold 7, NMS)", t_fast7, f_fast7);
  verdict("cv::GaussianBlur(7x7, 2, 2, REFLECT_101) under the calibration", t_blur, f_blur);
  verdict("cv::fastAtan2 under the calibration", t_atan, f_atan);
  failures += (t_resize.differing != 0) + (t_border.differing != 0) + (t_fast20.differing != 0) + (t_fast7.differing != 0) + (t_blur.differing != 0) + (t_atan.differing != 0);
#ifdef ORBX_VALIDATE_EXTRAS
  {
    Tally t_un, t_gray;
    std::string f_un, f_gray;
    const float fx = 458.654f, fy = 457.296f, cx = 367.215f, cy = 248.375f;   // Examples/Monocular/EuRoC.yaml
    const float dist[4] = {-0.28340811f, 0.07395907f, 0.00019359f, 1.76187114e-05f};
    cv::Mat K = cv::Mat::eye(3, 3, CV_32F);
    K.at<float>(0, 0) = fx; K.at<float>(1, 1) = fy; K.at<float>(0, 2) = cx; K.at<float>(1, 2) = cy;
    cv::Mat D(4, 1, CV_32F);
    for (int i = 0; i < 4; i++) D.at<float>(i) = dist[i];
    const int n = 20000;
    cv::Mat pts(n, 2, CV_32F);
    std::vector<float> in(2 * n), out(2 * n);
    uint32_t s = 7u;
    for (int i = 0; i < n; i++) {
      s = s * 1664525u + 1013904223u; in[2 * i] = (float)((s >> 8) % 752000u) * 1e-3f;
      s = s * 1664525u + 1013904223u; in[2 * i + 1] = (float)((s >> 8) % 480000u) * 1e-3f;
      pts.at<float>(i, 0) = in[2 * i]; pts.at<float>(i, 1) = in[2 * i + 1];
    }
    cv::Mat p2 = pts.reshape(2);
    cv::undistortPoints(p2, p2, K, D, cv::Mat(), K);   // src/Frame.cc:764-767
    p2 = p2.reshape(1);
    mo_undistort_points(in.data(), n, fx, fy, cx, cy, dist, 4, out.data());
    for (int i = 0; i < 2 * n; i++) {
      t_un.compared++;
      const float a = p2.at<float>(i / 2, i % 2), b = out[i];
      if (std::memcmp(&a, &b, 4) != 0 && !t_un.differing++) f_un = "point " + std::to_string(i / 2) + ": cv " + std::to_string(a) + " oracle " + std::to_string(b);
    }
    verdict("cv::undistortPoints (EuRoC intrinsics)", t_un, f_un);
    cv::Mat bgr(240, 320, CV_8UC3), gray;
    for (int i = 0; i < 240 * 320 * 3; i++) { s = s * 1664525u + 1013904223u; bgr.data[i] = (uint8_t)(s >> 16); }
    cv::cvtColor(bgr, gray, cv::COLOR_BGR2GRAY);
    for (int i = 0; i < 240 * 320; i++) {
      t_gray.compared++;
      const int B = bgr.data[3 * i], G = bgr.data[3 * i + 1], R = bgr.data[3 * i + 2];
      const int e = (R * 9798 + G * 19235 + B * 3735 + (1 << 14)) >> 15;
      if (e != gray.data[i] && !t_gray.differing++) f_gray = "pixel " + std::to_string(i) + ": cv " + std::to_string((int)gray.data[i]) + " restated " + std::to_string(e);
    }
    verdict("cv::cvtColor(BGR2GRAY)", t_gray, f_gray);
    failures += (t_un.differing != 0) + (t_gray.differing != 0);
  }
#endif

  // ---- 2b. which NAMED profile is this OpenCV?  (digests of what it computed against the table made over the oracle under every profile)
  {
    struct Row { const char* key; const char* what; uint64_t h; } rows[] = {
        {"resize", "cv::resize INTER_LINEAR (every level)", d_resize.h}, {"fast20", "cv::FAST threshold 20", d_fast20.h}, {"fast7", "cv::FAST threshold 7", d_fast7.h},
        {"blur", "cv::GaussianBlur 7x7 sigma 2", d_blur.h}, {"atan", "cv::fastAtan2", d_atan.h}};
    if (print_digests) {
      std::printf("DIGEST set %016llx\n", (unsigned long long)d_set.h);
      for (const Row& r : rows) std::printf("DIGEST %s %016llx\n", r.key, (unsigned long long)r.h);
    }
    if (expect_path) {
      FILE* f = std::fopen(expect_path, "r");
      if (!f) { std::fprintf(stderr, "cannot read %s\n", expect_path); return 2; }
      struct Exp { std::string profile, key; uint64_t h; };
      std::vector<Exp> table;
      char line[512];
      while (std::fgets(line, sizeof line, f)) {
        char prof[128], key[64];
        unsigned long long h = 0;
        if (line[0] == '#' || std::sscanf(line, "%127s %63s %llx", prof, key, &h) != 3) continue;
        table.push_back({prof, key, (uint64_t)h});
      }
      std::fclose(f);
      bool set_ok = false;
      for (const Exp& e : table) if (e.key == "set" && e.h == d_set.h) set_ok = true;
      std::printf("which named profile is this OpenCV (table %s, %zu rows):\n", expect_path, table.size());
      if (!set_ok) { std::printf("  the table was made for ANOTHER image set (set digest %016llx is not in it): regenerate the set with tools/make_validate_set.py <out> --synthetic 1\n", (unsigned long long)d_set.h); failures++; }
      else {
        std::string blur_prof, atan_prof;
        for (const Row& r : rows) {
          std::string who;
          for (const Exp& e : table) if (e.key == r.key && e.h == r.h) who += (who.empty() ? "" : ", ") + e.profile;
          std::printf("  %-40s %016llx  %s\n", r.what, (unsigned long long)r.h, who.empty() ? "IN NO PROFILE OF THE TABLE" : ("= " + who).c_str());
          if (who.empty()) failures++;
          if (std::string(r.key) == "blur") blur_prof = who;
          if (std::string(r.key) == "atan") atan_prof = who;
        }
        if (!blur_prof.empty() && !atan_prof.empty()) {
          const std::string first = blur_prof.substr(0, blur_prof.find(','));
          const int fma_build = (atan_prof.find("atan_fma=1") != std::string::npos ? 2 : 0) | (cal.brief_fma ? 1 : 0);
          std::printf("  -> orbx_set_cpu_profile(ctx, \"%s\", %d)   (bench.py --profile %s --fma-build %d)\n", first.c_str(), fma_build, first.c_str(), fma_build);
        }
      }
    }
  }

  // ---- 3. whole operator() on the GPU
  if (with_orbx) {
    orbx_ctx* ctx = nullptr;
    const int rc = orbx_create(&ctx, nfeatures, 1.2f, 8, 20, 7, -1);
    if (rc != ORBX_OK) { std::printf("orbx: orbx_create failed (%d) — no MI355X here?\n", rc); return 3; }
    if (orbx_cv::apply(ctx, cal) != ORBX_OK) { std::printf("orbx: %s\n", orbx_last_error(ctx)); return 3; }
    const int cap = orbx_keypoint_capacity(ctx);
    std::vector<orbx_keypoint> kps(cap);
    std::vector<uint8_t> desc((size_t)cap * 32);
    Tally t_op;
    std::string f_op;
    long total = 0;
#ifdef ORBX_VALIDATE_REFERENCE
    ORB_SLAM3::ORBextractor ref(nfeatures, 1.2f, 8, 20, 7);
#endif
    for (const Img& im : set) {
      int n = 0, mono = 0;
      if (orbx_extract(ctx, im.px.data(), im.rows, im.cols, (size_t)im.cols, 0, 1000, kps.data(), desc.data(), &n, &mono) != ORBX_OK) { std::printf("orbx: %s\n", orbx_last_error(ctx)); return 3; }
      total += n;
#ifdef ORBX_VALIDATE_REFERENCE
      cv::Mat image(im.rows, im.cols, CV_8UC1, (void*)im.px.data(), (size_t)im.cols), d;
      std::vector<cv::KeyPoint> k;
      std::vector<int> lap = {0, 1000};
      const int rmono = ref(image, cv::Mat(), k, d, lap);
      t_op.compared += (long)std::max((size_t)n, k.size()) * 2;
      if ((int)k.size() != n || rmono != mono) { if (!t_op.differing++) f_op = im.name + ": reference " + std::to_string(k.size()) + " keypoints / returns " + std::to_string(rmono) + ", orbx " + std::to_string(n) + " / " + std::to_string(mono); continue; }
      for (int i = 0; i < n; i++) {
        if (std::memcmp(&k[i], &kps[i], sizeof(orbx_keypoint)) != 0 && !t_op.differing++) f_op = im.name + " keypoint " + std::to_string(i) + " (x " + std::to_string(k[i].pt.x) + " vs " + std::to_string(kps[i].x) + ", angle " + std::to_string(k[i].angle) + " vs " + std::to_string(kps[i].angle) + ")";
        else if (std::memcmp(&k[i], &kps[i], sizeof(orbx_keypoint)) != 0) t_op.differing++;
        if (std::memcmp(d.ptr<unsigned char>(i), &desc[(size_t)i * 32], 32) != 0 && !t_op.differing++) f_op = im.name + " descriptor " + std::to_string(i);
        else if (std::memcmp(d.ptr<unsigned char>(i), &desc[(size_t)i * 32], 32) != 0) t_op.differing++;
      }
#endif
      if (verbose) std::printf("  %s: %d keypoints, returns %d\n", im.name.c_str(), n, mono);
    }
    orbx_destroy(ctx);
#ifdef ORBX_VALIDATE_REFERENCE
    std::printf("operator(): the reference's src/ORBextractor.cc over the OpenCV at hand vs liborbx.so (%ld keypoints):\n", total);
    verdict("keypoints (all 7 fields as bit patterns), descriptors, return value", t_op, f_op);
    failures += t_op.differing != 0;
#else
    std::printf("operator(): liborbx.so extracted %ld keypoints; build with -DORBX_VALIDATE_REFERENCE (tools/validate_opencv.cmake) to compare them with the reference's own operator()\n", total);
#endif
  }
  if (failures) std::printf("RESULT: %d check(s) FAILED — orbx will not be bit-identical to a CPU build over this OpenCV; INTEGRATION.md section 6\n", failures);
  else std::printf("RESULT: ALL MATCH\n");
  return failures ? 1 : 0;
}
