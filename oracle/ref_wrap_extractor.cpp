// TEST INFRASTRUCTURE: C wrapper around the REFERENCE's own ORBextractor (src/ORBextractor.cc + include/ORBextractor.h,
// compiled from /root/reference by oracle/ref_fragments.mk into oracle/_ref/libref_orbextractor.so against the container shim
// of oracle/ref_shims/opencv2).  Everything in that library is the reference's code — constructor tables, ComputePyramid's
// level chain and padding, the per-cell FAST loop with the iniTh -> minTh retry, DistributeOctTree / DivideNode / compareNodes
// with the host std::sort, IC_Angle, the steered BRIEF, operator()'s output order — except the five OpenCV algorithms it calls
// (FAST, resize, copyMakeBorder, GaussianBlur, fastAtan2), which the shim forwards to the oracle's isolated primitives.
// Used only to validate the oracle's restatement (tests/test_ref_fragments.py); never shipped, never measured as product.
#include <cstdint>
#include <cstring>
#include <vector>

#include "ORBextractor.h"

using ORB_SLAM3::ORBextractor;

extern "C" {

void* ref_ext_create(int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh) {
  return new ORBextractor(nfeatures, scaleFactor, nlevels, iniTh, minTh);
}
void ref_ext_destroy(void* h) { delete (ORBextractor*)h; }

// operator()(image, mask, keypoints, descriptors, vLappingArea); returns its return value, *n_out = keypoints
int ref_ext_extract(void* h, const uint8_t* img, int rows, int cols, int stride, int lap0, int lap1, void* kps, uint8_t* desc, int cap, int* n_out) {
  ORBextractor* e = (ORBextractor*)h;
  cv::Mat image(rows, cols, CV_8UC1, (void*)img, (size_t)stride);
  std::vector<cv::KeyPoint> keys;
  cv::Mat descriptors;
  std::vector<int> lap = {lap0, lap1};
  const int mono = (*e)(image, cv::Mat(), keys, descriptors, lap);
  *n_out = (int)keys.size();
  if ((int)keys.size() > cap) return -100000;
  std::memcpy(kps, keys.data(), keys.size() * sizeof(cv::KeyPoint));
  for (size_t i = 0; i < keys.size(); i++) std::memcpy(desc + i * 32, descriptors.ptr((int)i), 32);
  return mono;
}

void ref_ext_tables(void* h, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2) {
  ORBextractor* e = (ORBextractor*)h;
  const int n = e->GetLevels();
  std::vector<float> a = e->GetScaleFactors(), b = e->GetInverseScaleFactors(), c = e->GetScaleSigmaSquares(), d = e->GetInverseScaleSigmaSquares();
  for (int i = 0; i < n; i++) { scale[i] = a[i]; inv_scale[i] = b[i]; sigma2[i] = c[i]; inv_sigma2[i] = d[i]; }
}

// mvImagePyramid[level] with its EDGE_THRESHOLD padding: dst gets (h + 38) rows of (w + 38) bytes when with_border, else the ROI
int ref_ext_level(void* h, int level, int with_border, uint8_t* dst, int* w, int* hgt) {
  ORBextractor* e = (ORBextractor*)h;
  if (level < 0 || level >= (int)e->mvImagePyramid.size() || e->mvImagePyramid[level].empty()) return -1;
  const cv::Mat& L = e->mvImagePyramid[level];
  *w = L.cols; *hgt = L.rows;
  if (!dst) return 0;
  const int b = with_border ? 19 : 0;
  for (int r = -b; r < L.rows + b; r++) std::memcpy(dst + (size_t)(r + b) * (L.cols + 2 * b), L.data + (ptrdiff_t)r * (ptrdiff_t)L.step - b, L.cols + 2 * b);
  return 0;
}

}  // extern "C"
