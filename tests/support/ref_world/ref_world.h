// TEST INFRASTRUCTURE: one small object model — ORB_SLAM3::Frame / KeyFrame / MapPoint with exactly the members
// src/ORBmatcher.cc touches (include/Frame.h, include/KeyFrame.h, include/MapPoint.h) — shared by
//   * the REFERENCE's src/ORBmatcher.cc, compiled where it lies (oracle/ref_fragments.mk -> oracle/_ref/ref_matcher_world), and
//   * this repository's drop-in ORBmatcher (include/ORBmatcher.h + csrc/ref_adapter/ORBmatcher.cc -> tests/support/matcher_world.bin),
// so that both run on identical object graphs and their results can be compared byte for byte (tests/test_matcher_world.py).
// Own code.  Member names, types and the observable semantics of the mutators (AddObservation, Replace, AddMapPoint,
// ReplaceMapPointMatch, EraseMapPointMatch) follow the reference; locks, serialization, the map / atlas back-pointers and
// everything the matcher never reads are left out.  The include guards are the reference's, so the reference's own
// include/ORBmatcher.h (which includes "Frame.h" etc. from its own directory) picks THIS model up.
#ifndef FRAME_H
#define FRAME_H
#define KEYFRAME_H
#define MAPPOINT_H
#define MAP_H

#include <algorithm>
#include <cmath>
#include <map>
#include <set>
#include <tuple>
#include <vector>

#include <opencv2/core/core.hpp>

#include "CameraModels/GeometricCamera.h"
#include "Eigen/Core"
#include "Thirdparty/DBoW2/DBoW2/FeatureVector.h"
#include "sophus/se3.hpp"
#include "sophus/sim3.hpp"

// the reference's headers leak this (e.g. Thirdparty/DBoW2/DBoW2/FORB.h, include/KeyFrameDatabase.h) and its
// include/ORBmatcher.h:76 depends on it
using namespace std;

#define FRAME_GRID_ROWS 48
#define FRAME_GRID_COLS 64

namespace ORB_SLAM3 {

class KeyFrame;
class Frame;

// include/Map.h: what src/KeyFrameDatabase.cc asks of a map (:74-98 clearMap, :253, :714 IsBad)
class Map {
 public:
  long unsigned int mnId = 0;
  bool mbBad = false;
  bool IsBad() { return mbBad; }
  long unsigned int GetId() { return mnId; }
};

struct KeyFrameIdLess { bool operator()(const KeyFrame* a, const KeyFrame* b) const; };

class MapPoint {
 public:
  long unsigned int mnId = 0;
  // include/MapPoint.h:171-179 — variables used by the tracking
  float mTrackProjX = 0, mTrackProjY = 0, mTrackDepth = 0, mTrackDepthR = 0, mTrackProjXR = 0, mTrackProjYR = 0;
  bool mbTrackInView = false, mbTrackInViewR = false;
  int mnTrackScaleLevel = 0, mnTrackScaleLevelR = 0;
  float mTrackViewCos = 0, mTrackViewCosR = 0;

  Eigen::Vector3f GetWorldPos() { return mWorldPos; }
  Eigen::Vector3f GetNormal() { return mNormalVector; }
  int Observations() { return nObs; }
  bool isBad() { return mbBad; }
  cv::Mat GetDescriptor() { return mDescriptor.clone(); }
  float GetMinDistanceInvariance() { return 0.8f * mfMinDistance; }   // src/MapPoint.cc
  float GetMaxDistanceInvariance() { return 1.2f * mfMaxDistance; }
  inline int PredictScale(const float& currentDist, KeyFrame* pKF);   // src/MapPoint.cc:514-529
  inline int PredictScale(const float& currentDist, Frame* pF);       // :531-546
  std::tuple<int, int> GetIndexInKeyFrame(KeyFrame* pKF) {
    auto it = mObservations.find(pKF);
    return it != mObservations.end() ? it->second : std::tuple<int, int>(-1, -1);
  }
  bool IsInKeyFrame(KeyFrame* pKF) { return mObservations.count(pKF) != 0; }
  inline void AddObservation(KeyFrame* pKF, int idx);                 // src/MapPoint.cc:141-166
  inline void Replace(MapPoint* pMP);                                 // :248-299
  MapPoint* GetReplaced() { return mpReplaced; }
  // stand-in for ComputeDistinctiveDescriptors (:339-412): a deterministic change of the descriptor, so that a drop-in that
  // read descriptors at another moment than the reference would be caught
  void ComputeDistinctiveDescriptors() {
    if (mDescriptor.empty()) return;
    unsigned char* d = mDescriptor.ptr<unsigned char>();
    for (int i = 0; i < 32; i++) d[i] = (unsigned char)(d[i] ^ (unsigned char)(0x5a + 7 * i + 13 * nObs));
  }

  // test-side state (the reference keeps these protected)
  Eigen::Vector3f mWorldPos, mNormalVector;
  cv::Mat mDescriptor;
  float mfMinDistance = 0, mfMaxDistance = 0;
  bool mbBad = false;
  int nObs = 0;
  std::map<KeyFrame*, std::tuple<int, int>, KeyFrameIdLess> mObservations;
  MapPoint* mpReplaced = nullptr;
};

class KeyFrame {
 public:
  long unsigned int mnId = 0;
  int mnGridCols = FRAME_GRID_COLS, mnGridRows = FRAME_GRID_ROWS;
  float mfGridElementWidthInv = 0, mfGridElementHeightInv = 0;
  float fx = 0, fy = 0, cx = 0, cy = 0, invfx = 0, invfy = 0, mbf = 0, mb = 0, mThDepth = 0;
  int N = 0;
  std::vector<cv::KeyPoint> mvKeys, mvKeysUn, mvKeysRight;
  std::vector<float> mvuRight;
  cv::Mat mDescriptors;
  DBoW2::BowVector mBowVec;
  DBoW2::FeatureVector mFeatVec;
  // include/KeyFrame.h:335-346 — variables used by the keyframe database (the reference leaves them uninitialised until the first
  // query; 0 here, and the scenario drivers only print fields a query has written or that start from this 0)
  long unsigned int mnLoopQuery = 0;
  int mnLoopWords = 0;
  float mLoopScore = 0;
  long unsigned int mnRelocQuery = 0;
  int mnRelocWords = 0;
  float mRelocScore = 0;
  long unsigned int mnMergeQuery = 0;
  int mnMergeWords = 0;
  float mMergeScore = 0;
  long unsigned int mnPlaceRecognitionQuery = 0;
  int mnPlaceRecognitionWords = 0;
  float mPlaceRecognitionScore = 0;
  int mnScaleLevels = 0;
  float mfScaleFactor = 0, mfLogScaleFactor = 0;
  std::vector<float> mvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
  int mnMinX = 0, mnMinY = 0, mnMaxX = 0, mnMaxY = 0;   // include/KeyFrame.h:403-406: int, truncated from the Frame's floats
  GeometricCamera *mpCamera = nullptr, *mpCamera2 = nullptr;
  int NLeft = -1, NRight = -1;
  std::vector<std::vector<std::vector<size_t> > > mGrid, mGridRight;

  Sophus::SE3f GetPose() { return mTcw; }
  Sophus::SE3f GetPoseInverse() { return mTcw.inverse(); }
  Eigen::Vector3f GetCameraCenter() { return mTcw.inverse().translation(); }
  Sophus::SE3<float> GetRightPose() { return mTrl * mTcw; }                       // src/KeyFrame.cc
  Sophus::SE3<float> GetRightPoseInverse() { return (mTrl * mTcw).inverse(); }
  Eigen::Vector3f GetRightCameraCenter() { return (mTrl * mTcw).inverse().translation(); }

  void AddMapPoint(MapPoint* pMP, const size_t& idx) { mvpMapPoints[idx] = pMP; }
  void EraseMapPointMatch(const int& idx) { mvpMapPoints[idx] = static_cast<MapPoint*>(NULL); }
  void ReplaceMapPointMatch(const int& idx, MapPoint* pMP) { mvpMapPoints[idx] = pMP; }
  std::set<MapPoint*> GetMapPoints() {   // src/KeyFrame.cc: the good ones
    std::set<MapPoint*> s;
    for (MapPoint* pMP : mvpMapPoints)
      if (pMP && !pMP->isBad()) s.insert(pMP);
    return s;
  }
  std::vector<MapPoint*> GetMapPointMatches() { return mvpMapPoints; }
  MapPoint* GetMapPoint(const size_t& idx) { return mvpMapPoints[idx]; }
  bool IsInImage(const float& x, const float& y) const { return (x >= mnMinX && x < mnMaxX && y >= mnMinY && y < mnMaxY); }   // src/KeyFrame.cc:750-753

  // src/KeyFrame.cc:704-748
  std::vector<size_t> GetFeaturesInArea(const float& x, const float& y, const float& r, const bool bRight = false) const {
    std::vector<size_t> vIndices;
    vIndices.reserve(N);
    float factorX = r;
    float factorY = r;
    const int nMinCellX = std::max(0, (int)floor((x - mnMinX - factorX) * mfGridElementWidthInv));
    if (nMinCellX >= mnGridCols) return vIndices;
    const int nMaxCellX = std::min((int)mnGridCols - 1, (int)ceil((x - mnMinX + factorX) * mfGridElementWidthInv));
    if (nMaxCellX < 0) return vIndices;
    const int nMinCellY = std::max(0, (int)floor((y - mnMinY - factorY) * mfGridElementHeightInv));
    if (nMinCellY >= mnGridRows) return vIndices;
    const int nMaxCellY = std::min((int)mnGridRows - 1, (int)ceil((y - mnMinY + factorY) * mfGridElementHeightInv));
    if (nMaxCellY < 0) return vIndices;
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++) {
      for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
        const std::vector<size_t> vCell = (!bRight) ? mGrid[ix][iy] : mGridRight[ix][iy];
        for (size_t j = 0, jend = vCell.size(); j < jend; j++) {
          const cv::KeyPoint& kpUn = (NLeft == -1) ? mvKeysUn[vCell[j]] : (!bRight) ? mvKeys[vCell[j]] : mvKeysRight[vCell[j]];
          const float distx = kpUn.pt.x - x;
          const float disty = kpUn.pt.y - y;
          if (fabs(distx) < r && fabs(disty) < r) vIndices.push_back(vCell[j]);
        }
      }
    }
    return vIndices;
  }

  // covisibility graph and map as src/KeyFrameDatabase.cc reads them (src/KeyFrame.cc:228-251, :1140)
  std::set<KeyFrame*> GetConnectedKeyFrames() {
    std::set<KeyFrame*> s;
    for (std::map<KeyFrame*, int>::iterator mit = mConnectedKeyFrameWeights.begin(); mit != mConnectedKeyFrameWeights.end(); mit++) s.insert(mit->first);
    return s;
  }
  std::vector<KeyFrame*> GetBestCovisibilityKeyFrames(const int& N) {
    if ((int)mvpOrderedConnectedKeyFrames.size() < N) return mvpOrderedConnectedKeyFrames;
    return std::vector<KeyFrame*>(mvpOrderedConnectedKeyFrames.begin(), mvpOrderedConnectedKeyFrames.begin() + N);
  }
  Map* GetMap() { return mpMap; }
  bool isBad() { return mbBad; }

  // test-side state
  Sophus::SE3f mTcw, mTrl;
  std::vector<MapPoint*> mvpMapPoints;
  std::map<KeyFrame*, int> mConnectedKeyFrameWeights;
  std::vector<KeyFrame*> mvpOrderedConnectedKeyFrames;
  Map* mpMap = nullptr;
  bool mbBad = false;
};

inline bool KeyFrameIdLess::operator()(const KeyFrame* a, const KeyFrame* b) const { return a->mnId < b->mnId; }

class Frame {
 public:
  long unsigned int mnId = 0;   // include/Frame.h:271-272: unique per frame (nNextId++ in the constructors), shared by copies
  float mbf = 0, mb = 0;
  int N = 0;
  std::vector<cv::KeyPoint> mvKeys, mvKeysRight, mvKeysUn;
  std::vector<MapPoint*> mvpMapPoints;
  std::vector<float> mvuRight, mvDepth;
  DBoW2::BowVector mBowVec;
  DBoW2::FeatureVector mFeatVec;
  cv::Mat mDescriptors, mDescriptorsRight;
  std::vector<bool> mvbOutlier;
  inline static float mfGridElementWidthInv = 0, mfGridElementHeightInv = 0;
  std::vector<std::size_t> mGrid[FRAME_GRID_COLS][FRAME_GRID_ROWS];
  int mnScaleLevels = 0;
  float mfScaleFactor = 0, mfLogScaleFactor = 0;
  std::vector<float> mvScaleFactors, mvInvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
  inline static float mnMinX = 0, mnMaxX = 0, mnMinY = 0, mnMaxY = 0;
  GeometricCamera *mpCamera = nullptr, *mpCamera2 = nullptr;
  int Nleft = -1, Nright = -1;
  std::vector<int> mvLeftToRightMatch, mvRightToLeftMatch;
  std::vector<std::size_t> mGridRight[FRAME_GRID_COLS][FRAME_GRID_ROWS];

  inline Sophus::SE3<float> GetPose() const { return mTcw; }
  Sophus::SE3f GetRelativePoseTrl() { return mTrl; }

  // src/Frame.cc:725-735
  bool PosInGrid(const cv::KeyPoint& kp, int& posX, int& posY) {
    posX = round((kp.pt.x - mnMinX) * mfGridElementWidthInv);
    posY = round((kp.pt.y - mnMinY) * mfGridElementHeightInv);
    if (posX < 0 || posX >= FRAME_GRID_COLS || posY < 0 || posY >= FRAME_GRID_ROWS) return false;
    return true;
  }
  // src/Frame.cc:385-416
  void AssignFeaturesToGrid() {
    for (int i = 0; i < FRAME_GRID_COLS; i++)
      for (int j = 0; j < FRAME_GRID_ROWS; j++) { mGrid[i][j].clear(); mGridRight[i][j].clear(); }
    for (int i = 0; i < N; i++) {
      const cv::KeyPoint& kp = (Nleft == -1) ? mvKeysUn[i] : (i < Nleft) ? mvKeys[i] : mvKeysRight[i - Nleft];
      int nGridPosX, nGridPosY;
      if (PosInGrid(kp, nGridPosX, nGridPosY)) {
        if (Nleft == -1 || i < Nleft) mGrid[nGridPosX][nGridPosY].push_back(i);
        else mGridRight[nGridPosX][nGridPosY].push_back(i - Nleft);
      }
    }
  }
  // src/Frame.cc:657-723
  std::vector<size_t> GetFeaturesInArea(const float& x, const float& y, const float& r, const int minLevel = -1, const int maxLevel = -1,
                                        const bool bRight = false) const {
    std::vector<size_t> vIndices;
    vIndices.reserve(N);
    float factorX = r;
    float factorY = r;
    const int nMinCellX = std::max(0, (int)floor((x - mnMinX - factorX) * mfGridElementWidthInv));
    if (nMinCellX >= FRAME_GRID_COLS) return vIndices;
    const int nMaxCellX = std::min((int)FRAME_GRID_COLS - 1, (int)ceil((x - mnMinX + factorX) * mfGridElementWidthInv));
    if (nMaxCellX < 0) return vIndices;
    const int nMinCellY = std::max(0, (int)floor((y - mnMinY - factorY) * mfGridElementHeightInv));
    if (nMinCellY >= FRAME_GRID_ROWS) return vIndices;
    const int nMaxCellY = std::min((int)FRAME_GRID_ROWS - 1, (int)ceil((y - mnMinY + factorY) * mfGridElementHeightInv));
    if (nMaxCellY < 0) return vIndices;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++) {
      for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
        const std::vector<size_t> vCell = (!bRight) ? mGrid[ix][iy] : mGridRight[ix][iy];
        if (vCell.empty()) continue;
        for (size_t j = 0, jend = vCell.size(); j < jend; j++) {
          const cv::KeyPoint& kpUn = (Nleft == -1) ? mvKeysUn[vCell[j]] : (!bRight) ? mvKeys[vCell[j]] : mvKeysRight[vCell[j]];
          if (bCheckLevels) {
            if (kpUn.octave < minLevel) continue;
            if (maxLevel >= 0)
              if (kpUn.octave > maxLevel) continue;
          }
          const float distx = kpUn.pt.x - x;
          const float disty = kpUn.pt.y - y;
          if (fabs(distx) < factorX && fabs(disty) < factorY) vIndices.push_back(vCell[j]);
        }
      }
    }
    return vIndices;
  }

  // test-side state
  Sophus::SE3f mTcw, mTrl;
};

inline int MapPoint::PredictScale(const float& currentDist, KeyFrame* pKF) {
  float ratio = mfMaxDistance / currentDist;
  int nScale = ceil(log(ratio) / pKF->mfLogScaleFactor);
  if (nScale < 0) nScale = 0;
  else if (nScale >= pKF->mnScaleLevels) nScale = pKF->mnScaleLevels - 1;
  return nScale;
}
inline int MapPoint::PredictScale(const float& currentDist, Frame* pF) {
  float ratio = mfMaxDistance / currentDist;
  int nScale = ceil(log(ratio) / pF->mfLogScaleFactor);
  if (nScale < 0) nScale = 0;
  else if (nScale >= pF->mnScaleLevels) nScale = pF->mnScaleLevels - 1;
  return nScale;
}
inline void MapPoint::AddObservation(KeyFrame* pKF, int idx) {
  std::tuple<int, int> indexes = mObservations.count(pKF) ? mObservations[pKF] : std::tuple<int, int>(-1, -1);
  if (pKF->NLeft != -1 && idx >= pKF->NLeft) std::get<1>(indexes) = idx;
  else std::get<0>(indexes) = idx;
  mObservations[pKF] = indexes;
  if (!pKF->mpCamera2 && pKF->mvuRight[idx] >= 0) nObs += 2;
  else nObs++;
}
inline void MapPoint::Replace(MapPoint* pMP) {
  if (pMP->mnId == this->mnId) return;
  std::map<KeyFrame*, std::tuple<int, int>, KeyFrameIdLess> obs = mObservations;
  mObservations.clear();
  mbBad = true;
  mpReplaced = pMP;
  for (auto& kv : obs) {
    KeyFrame* pKF = kv.first;
    const int leftIndex = std::get<0>(kv.second), rightIndex = std::get<1>(kv.second);
    if (!pMP->IsInKeyFrame(pKF)) {
      if (leftIndex != -1) { pKF->ReplaceMapPointMatch(leftIndex, pMP); pMP->AddObservation(pKF, leftIndex); }
      if (rightIndex != -1) { pKF->ReplaceMapPointMatch(rightIndex, pMP); pMP->AddObservation(pKF, rightIndex); }
    } else {
      if (leftIndex != -1) pKF->EraseMapPointMatch(leftIndex);
      if (rightIndex != -1) pKF->EraseMapPointMatch(rightIndex);
    }
  }
  pMP->ComputeDistinctiveDescriptors();
}

}  // namespace ORB_SLAM3

#endif  // FRAME_H
