// TEST INFRASTRUCTURE: the part of include/CameraModels/GeometricCamera.h (:61-86) the matcher path calls, plus two
// concrete stand-in models for the tests (the reference's Pinhole / KannalaBrandt8 live outside the hot path).
#pragma once
#include <cmath>
#include <opencv2/core/core.hpp>
#include "../Eigen/Core"

namespace ORB_SLAM3 {

class GeometricCamera {
 public:
  virtual ~GeometricCamera() {}
  virtual Eigen::Vector2f project(const Eigen::Vector3f& v3D) = 0;
  virtual bool epipolarConstrain(GeometricCamera* otherCamera, const cv::KeyPoint& kp1, const cv::KeyPoint& kp2, const Eigen::Matrix3f& R12,
                                 const Eigen::Vector3f& t12, const float sigmaLevel, const float unc) = 0;
};

// stand-in pinhole: u = fx * x / z + cx.  epipolarConstrain: distance of kp2 to the epipolar line of kp1, like Pinhole.cpp
struct TestPinhole : GeometricCamera {
  float fx = 458.f, fy = 457.f, cx = 367.f, cy = 248.f;
  Eigen::Vector2f project(const Eigen::Vector3f& p) override { return Eigen::Vector2f(fx * p(0) / p(2) + cx, fy * p(1) / p(2) + cy); }
  Eigen::Vector3f ray(const cv::KeyPoint& kp) const { return Eigen::Vector3f((kp.pt.x - cx) / fx, (kp.pt.y - cy) / fy, 1.f); }
  bool epipolarConstrain(GeometricCamera* other, const cv::KeyPoint& kp1, const cv::KeyPoint& kp2, const Eigen::Matrix3f& R12,
                         const Eigen::Vector3f& t12, const float, const float unc) override {
    TestPinhole* o = static_cast<TestPinhole*>(other);
    // l2 = F12^T x1 with F12 = K1^-T [t12]x R12 K2^-1, written on normalised rays
    const Eigen::Vector3f x1 = ray(kp1), x2 = o->ray(kp2);
    const Eigen::Vector3f Rx2 = R12 * x2;
    const Eigen::Vector3f n(t12(1) * Rx2(2) - t12(2) * Rx2(1), t12(2) * Rx2(0) - t12(0) * Rx2(2), t12(0) * Rx2(1) - t12(1) * Rx2(0));
    const float num = x1.dot(n);
    const float den = n(0) * n(0) + n(1) * n(1);
    if (den == 0.f) return false;
    const float dsqr = num * num / den * fx * fx;
    return dsqr < 3.84f * unc;
  }
};

// stand-in wide-angle model (equidistant): r = f * atan2(rho, z); a different project() from the pinhole
struct TestFisheye : TestPinhole {
  Eigen::Vector2f project(const Eigen::Vector3f& p) override {
    const float rho = std::sqrt(p(0) * p(0) + p(1) * p(1));
    if (rho < 1e-9f) return Eigen::Vector2f(cx, cy);
    const float theta = std::atan2(rho, p(2));
    return Eigen::Vector2f(fx * theta * p(0) / rho + cx, fy * theta * p(1) / rho + cy);
  }
};

}  // namespace ORB_SLAM3
