// TEST INFRASTRUCTURE (-include'd before everything): lets the reference's src/Frame.cc and include/Frame.h compile UNMODIFIED, where
// they lie, without Eigen / g2o / the IMU code.  include/Frame.h pulls "ImuTypes.h", "Converter.h" and "Settings.h" from its own directory
// (a quoted include looks beside the includer first), so those three are replaced by defining their include guards here and supplying the
// few members src/Frame.cc uses; every other dependency is shadowed through the include path (this directory comes first).  Own code.
#pragma once
#include <climits>
#include <cmath>
#include <cstdlib>
#include <iostream>
#include <list>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#include <opencv2/imgproc/imgproc.hpp>
#include <opencv2/frame_world_extras.hpp>

#include "Eigen/Core"
#include "sophus/se3.hpp"

#ifdef FRAME_WORLD_DROPIN
#include <ORBVocabulary.h>   // this repository's (the include path puts its include/ first); same guard as the reference's, which is then skipped
#endif

using namespace std;   // the reference's headers leak this (Thirdparty/DBoW2/DBoW2/FORB.h) and include/Frame.h:105 relies on it

// ---- include/ImuTypes.h: the members src/Frame.cc touches (:439-444, :457-470, :481-491)
#define IMUTYPES_H
namespace ORB_SLAM3 {
namespace IMU {
class Bias {};
class Calib {
 public:
  Sophus::SE3<float> mTbc, mTcb;
  bool mbIsSet = false;
};
class Preintegrated {
 public:
  void SetNewBias(const Bias&) {}
};
}  // namespace IMU
}  // namespace ORB_SLAM3

// ---- include/Converter.h: :57,102,202,1035 toMatrix3f (src/Converter.cc: element-wise copy of a 3x3 CV_32F), :742 toDescriptorVector
#define CONVERTER_H
namespace ORB_SLAM3 {
class Converter {
 public:
  static Eigen::Matrix3f toMatrix3f(const cv::Mat& m) {
    Eigen::Matrix3f r;
    if (!m.empty()) for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r(i, j) = m.at<float>(i, j);
    return r;
  }
  static std::vector<cv::Mat> toDescriptorVector(const cv::Mat& D) {
    std::vector<cv::Mat> v;
    v.reserve(D.rows);
    for (int j = 0; j < D.rows; j++) v.push_back(D.row(j));
    return v;
  }
};
}  // namespace ORB_SLAM3

// ---- include/Settings.h: only included, never used by Frame
#define ORB_SLAM3_SETTINGS_H
