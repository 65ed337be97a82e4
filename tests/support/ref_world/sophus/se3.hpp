// TEST INFRASTRUCTURE: Sophus::SE3<float> with the members the matcher path uses (see Eigen/Core beside this file).
#pragma once
#include "../Eigen/Core"

namespace Sophus {

template <class Scalar> class SE3;
template <> class SE3<float> {
 public:
  SE3() {}
  SE3(const Eigen::Matrix3f& R, const Eigen::Vector3f& t) : R_(R), t_(t) {}
  const Eigen::Matrix3f& rotationMatrix() const { return R_; }
  const Eigen::Vector3f& translation() const { return t_; }
  SE3 inverse() const { const Eigen::Matrix3f Rt = R_.transpose(); return SE3(Rt, -(Rt * t_)); }
  Eigen::Vector3f operator*(const Eigen::Vector3f& p) const { return R_ * p + t_; }
  SE3 operator*(const SE3& o) const { return SE3(R_ * o.R_, R_ * o.t_ + t_); }
 private:
  Eigen::Matrix3f R_;
  Eigen::Vector3f t_;
};
typedef SE3<float> SE3f;

}  // namespace Sophus
