// TEST INFRASTRUCTURE (see opencv2/core/core.hpp): cv::FAST as called at src/ORBextractor.cc:826,845, forwarded to the
// oracle's isolated FAST-9/16 + score + 3x3 NMS primitive (orbo_fast, "recalled OpenCV semantics").
#pragma once
#include "../core/core.hpp"

extern "C" int orbo_fast(const uint8_t* img, int cols, int rows, int stride, int threshold, int nms, void* dst, int cap);

namespace cv {

static inline void FAST(const Mat& image, std::vector<KeyPoint>& keypoints, int threshold, bool nonmaxSuppression = true) {
  assert(image.type() == CV_8UC1);
  int cap = image.rows * image.cols + 1;
  keypoints.resize(cap);
  const int n = orbo_fast(image.data, image.cols, image.rows, (int)image.step, threshold, nonmaxSuppression ? 1 : 0, keypoints.data(), cap);
  keypoints.resize(n < 0 ? 0 : n);
}

// only named by the dead ComputeKeyPointsOld (src/ORBextractor.cc:898-1075, its call is commented out at :1101)
struct KeyPointsFilter { static void retainBest(std::vector<KeyPoint>&, int) { std::abort(); } };

}  // namespace cv
