// TEST INFRASTRUCTURE: Sophus::SE3<float> (ref_world's) plus the empty SE3<double> include/Frame.h:371 declares.
#pragma once
#include "../Eigen/Core"
#include "../../ref_world/sophus/se3.hpp"
namespace Sophus {
template <> class SE3<double> {};
}
