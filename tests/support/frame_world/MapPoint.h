// TEST INFRASTRUCTURE: the members of include/MapPoint.h that src/Frame.cc reads and writes (isInFrustum :512-586, ProjectPointDistort
// :588-650, isInFrustumChecks :1168-1242).  PredictScale(dist, Frame*) restates src/MapPoint.cc:531-546.
#pragma once
#include <cmath>
#include "Eigen/Core"
namespace ORB_SLAM3 {
class Frame;
class MapPoint {
 public:
  long unsigned int mnId = 0;
  float mTrackProjX = 0, mTrackProjY = 0, mTrackDepth = 0, mTrackDepthR = 0, mTrackProjXR = 0, mTrackProjYR = 0;
  bool mbTrackInView = false, mbTrackInViewR = false;
  int mnTrackScaleLevel = 0, mnTrackScaleLevelR = 0;
  float mTrackViewCos = 0, mTrackViewCosR = 0;
  Eigen::Vector3f mWorldPos, mNormalVector;
  float mfMinDistance = 0, mfMaxDistance = 0;
  Eigen::Vector3f GetWorldPos() { return mWorldPos; }
  Eigen::Vector3f GetNormal() { return mNormalVector; }
  float GetMinDistanceInvariance() { return 0.8f * mfMinDistance; }
  float GetMaxDistanceInvariance() { return 1.2f * mfMaxDistance; }
  int PredictScale(const float& currentDist, Frame* pF);   // defined in tests/support/frame_world.cpp (it needs the complete Frame)
};
}  // namespace ORB_SLAM3
