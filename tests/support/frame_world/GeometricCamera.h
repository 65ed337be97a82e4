// TEST INFRASTRUCTURE: include/CameraModels/GeometricCamera.h as src/Frame.cc sees it (project :532,:1198; the Pinhole / KannalaBrandt8
// casts of :291,:345-348,:766,:793,:1059-1060,:1156), with stand-in models — the reference's camera models are geometry outside the hot path.
#pragma once
#include <cmath>
#include <vector>
#include <opencv2/core/core.hpp>
#include "Eigen/Core"
namespace ORB_SLAM3 {
class GeometricCamera {
 public:
  virtual ~GeometricCamera() {}
  virtual Eigen::Vector2f project(const Eigen::Vector3f& v3D) = 0;
};
class Pinhole : public GeometricCamera {
 public:
  float fx = 458.f, fy = 457.f, cx = 367.f, cy = 248.f;
  Eigen::Vector2f project(const Eigen::Vector3f& p) override { return Eigen::Vector2f(fx * p(0) / p(2) + cx, fy * p(1) / p(2) + cy); }
  cv::Mat toK() {   // src/CameraModels/Pinhole.cpp:129-133
    cv::Mat K = cv::Mat::zeros(3, 3, CV_32F);
    K.at<float>(0, 0) = fx; K.at<float>(1, 1) = fy; K.at<float>(0, 2) = cx; K.at<float>(1, 2) = cy; K.at<float>(2, 2) = 1.f;
    return K;
  }
  Eigen::Matrix3f toK_() {
    Eigen::Matrix3f K;
    K(0, 0) = fx; K(0, 1) = 0; K(0, 2) = cx; K(1, 0) = 0; K(1, 1) = fy; K(1, 2) = cy; K(2, 0) = 0; K(2, 1) = 0; K(2, 2) = 1.f;
    return K;
  }
};
class KannalaBrandt8 : public Pinhole {
 public:
  std::vector<int> mvLappingArea = std::vector<int>(2, 0);
  // stand-in for the triangulation of src/CameraModels/KannalaBrandt8.cpp:403-470 (geometry, outside the hot path): a deterministic
  // depth from the two keypoints, so that the bookkeeping of Frame::ComputeStereoFishEyeMatches (:1126-1166) is exercised
  float TriangulateMatches(GeometricCamera*, const cv::KeyPoint& kp1, const cv::KeyPoint& kp2, const Eigen::Matrix3f&, const Eigen::Vector3f&,
                           const float sigmaLevel, const float unc, Eigen::Vector3f& p3D) {
    const float d = kp1.pt.x - kp2.pt.x + 0.25f * (sigmaLevel - unc);
    if (std::fabs(kp1.pt.y - kp2.pt.y) > 40.f || d <= 0.5f) return -1.f;
    const float z = 400.f / d;
    p3D = Eigen::Vector3f((kp1.pt.x - cx) * z / fx, (kp1.pt.y - cy) * z / fy, z);
    return z;
  }
};
}  // namespace ORB_SLAM3
