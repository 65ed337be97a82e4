// TEST INFRASTRUCTURE: what src/Frame.cc uses of include/ORBmatcher.h — the two thresholds (src/ORBmatcher.cc:35-36) and
// DescriptorDistance (:2009-2029, the 256-bit Hamming distance; pinned separately against the reference's own code by
// tests/test_ref_fragments.py and the `statics` record of tests/test_matcher_world.py).
#pragma once
#include <cstdint>
#include <cstring>
#include <opencv2/core/core.hpp>
namespace ORB_SLAM3 {
class ORBmatcher {
 public:
  static const int TH_LOW = 50;
  static const int TH_HIGH = 100;
  static const int HISTO_LENGTH = 30;
  static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b) {
    int d = 0;
    for (int i = 0; i < 4; i++) {
      uint64_t x, y;
      std::memcpy(&x, a.ptr<unsigned char>() + 8 * i, 8);
      std::memcpy(&y, b.ptr<unsigned char>() + 8 * i, 8);
      d += __builtin_popcountll(x ^ y);
    }
    return d;
  }
};
}  // namespace ORB_SLAM3
