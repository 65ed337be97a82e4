// TEST INFRASTRUCTURE: Sophus::Sim3<float> with the members the matcher path uses (see Eigen/Core beside this file).
#pragma once
#include "se3.hpp"

namespace Sophus {

template <class Scalar> class Sim3;
template <> class Sim3<float> {
 public:
  Sim3() {}
  // test-only constructor (the real class is built from an RxSO3 and a translation; the matcher never constructs one)
  Sim3(float s, const Eigen::Matrix3f& R, const Eigen::Vector3f& t) : s_(s), R_(R), t_(t) {}
  const Eigen::Matrix3f& rotationMatrix() const { return R_; }
  const Eigen::Vector3f& translation() const { return t_; }
  float scale() const { return s_; }
  Sim3 inverse() const {
    const Eigen::Matrix3f Rt = R_.transpose();
    const float si = 1.0f / s_;
    return Sim3(si, Rt, -((Rt * t_) * si));
  }
  Eigen::Vector3f operator*(const Eigen::Vector3f& p) const { return (R_ * p) * s_ + t_; }
 private:
  float s_ = 1.0f;
  Eigen::Matrix3f R_;
  Eigen::Vector3f t_;
};
typedef Sim3<float> Sim3f;

}  // namespace Sophus
