// TEST INFRASTRUCTURE (see opencv2/core/core.hpp beside this file): the imgproc / core ALGORITHMS src/ORBextractor.cc calls
// (:102 fastAtan2, :1133 GaussianBlur, :1183 resize, :1185,:1190 copyMakeBorder), forwarded to the oracle's isolated
// primitives (oracle/orb_oracle.cpp: orbo_*).  They stay "recalled OpenCV semantics"; everything AROUND them in the
// reference-compiled library is the reference's own code.
#pragma once
#include "../core/core.hpp"

extern "C" {
void orbo_resize_linear(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride);
void orbo_gaussian_blur7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride);
float orbo_fast_atan2(float y, float x);
}

namespace cv {

enum { INTER_LINEAR = 1 };
enum { BORDER_REFLECT_101 = 4, BORDER_ISOLATED = 16 };

static inline float fastAtan2(float y, float x) { return orbo_fast_atan2(y, x); }

// cv::resize(src, dst, dsize, 0, 0, INTER_LINEAR): dst.create(dsize) keeps dst when it already has that size — at
// src/ORBextractor.cc:1183 dst is the ROI inside the padded level buffer, which must be written in place
static inline void resize(const Mat& src, Mat& dst, Size dsize, double fx, double fy, int interpolation) {
  assert(fx == 0 && fy == 0 && interpolation == INTER_LINEAR && src.type() == CV_8UC1);
  (void)fx; (void)fy; (void)interpolation;
  dst.create(dsize.height, dsize.width, src.type());
  orbo_resize_linear(src.data, src.cols, src.rows, (int)src.step, dst.data, dst.cols, dst.rows, (int)dst.step);
}

// cv::copyMakeBorder with BORDER_REFLECT_101: idx(-k) = k, idx(n-1+k) = n-1-k.  dst keeps its
// buffer when it already has the padded size (then src may be the ROI in the middle of dst: rows are processed so that
// no source pixel is overwritten before it is read — the interior is copied onto itself).  Without BORDER_ISOLATED OpenCV
// would read real pixels around a source ROI; the extractor passes BORDER_ISOLATED for levels >= 1 and a whole image at
// level 0, so the isolated form is what both calls compute.
static inline void copyMakeBorder(const Mat& src, Mat& dst, int top, int bottom, int left, int right, int borderType) {
  assert((borderType & ~BORDER_ISOLATED) == BORDER_REFLECT_101 && src.type() == CV_8UC1);
  (void)borderType;
  const int w = src.cols, h = src.rows;
  dst.create(h + top + bottom, w + left + right, src.type());
  auto refl = [](int i, int n) { if (n == 1) return 0; while (i < 0 || i >= n) { if (i < 0) i = -i; else i = 2 * (n - 1) - i; } return i; };
  std::vector<unsigned char> tmp((size_t)h * w);
  for (int r = 0; r < h; r++) std::memcpy(&tmp[(size_t)r * w], src.data + (size_t)r * src.step, w);
  for (int r = 0; r < dst.rows; r++) {
    const unsigned char* s = &tmp[(size_t)refl(r - top, h) * w];
    unsigned char* d = dst.data + (size_t)r * dst.step;
    for (int c = 0; c < dst.cols; c++) d[c] = s[refl(c - left, w)];
  }
}

// cv::GaussianBlur(src, dst, Size(7,7), 2, 2, BORDER_REFLECT_101), in place on a continuous clone (:1132-1133)
static inline void GaussianBlur(const Mat& src, Mat& dst, Size ksize, double sx, double sy, int borderType) {
  assert(ksize.width == 7 && ksize.height == 7 && sx == 2 && sy == 2 && borderType == BORDER_REFLECT_101 && src.type() == CV_8UC1);
  (void)ksize; (void)sx; (void)sy; (void)borderType;
  Mat out(src.rows, src.cols, src.type());
  orbo_gaussian_blur7(src.data, src.cols, src.rows, (int)src.step, out.data, (int)out.step);
#ifdef ORBO_SHIM_UNKNOWN_BLUR   // tests only: "an OpenCV whose Gaussian no known variant reproduces" (every 97th byte one off)
  for (int i = 0; i < src.rows * src.cols; i += 97) { unsigned char& b = out.data[(i / src.cols) * (int)out.step + i % src.cols]; b = (unsigned char)(b < 255 ? b + 1 : 254); }
#endif
  out.copyTo(dst);
}

}  // namespace cv
