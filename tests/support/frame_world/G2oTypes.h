// TEST INFRASTRUCTURE: src/Frame.cc includes include/G2oTypes.h for the ConstraintPoseImu pointer it copies (:56); nothing else is used.
#pragma once
namespace ORB_SLAM3 { class ConstraintPoseImu {}; }
