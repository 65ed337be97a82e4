// TEST INFRASTRUCTURE: src/Frame.cc only holds KeyFrame pointers (mpReferenceKF, mpLastKeyFrame).
#pragma once
namespace ORB_SLAM3 { class KeyFrame {}; }
