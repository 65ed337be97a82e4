/* orbx adapter — drop-in replacement for the reference's include/ORBmatcher.h (:36-103): the same class, the same twelve
 * public search routines with the same signatures, the public thresholds, EIGEN_MAKE_ALIGNED_OPERATOR_NEW and the protected
 * helpers, so that Tracking.cc, LocalMapping.cc, LoopClosing.cc and CloudPoint.cc compile and link unchanged.  The
 * definitions live in orb_slam3_modified_amd/csrc/ref_adapter/ORBmatcher.cc (it replaces src/ORBmatcher.cc in the
 * reference's build): windows / vocabulary-node candidate lists, every Hamming distance and the
 * order-free arg-min run on the GPU through include/orbx.h; the pose / camera arithmetic and the MapPoint / KeyFrame
 * bookkeeping run through the reference's own objects, in the reference's order.
 *
 * Like the reference's header this one includes the reference's MapPoint.h / KeyFrame.h / Frame.h and sophus/sim3.hpp — it is
 * meant to sit in the reference's include/ directory.  In this repository's tests the same names resolve to the small object
 * model of tests/support/ref_world/.
 */
#ifndef ORBMATCHER_H
#define ORBMATCHER_H

#include <set>
#include <utility>
#include <vector>

#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#include "sophus/sim3.hpp"

#include "MapPoint.h"
#include "KeyFrame.h"
#include "Frame.h"

#include "orbx.h"

namespace ORB_SLAM3 {

class ORBmatcher {
 public:
  ORBmatcher(float nnratio = 0.6, bool checkOri = true);

  // Computes the Hamming distance between two ORB descriptors (src/ORBmatcher.cc:2058-2074)
  static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b);

  // Tracking::SearchLocalPoints (src/ORBmatcher.cc:43-212)
  int SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th = 3, const bool bFarPoints = false,
                         const float thFarPoints = 50.0f);

  // Tracking::TrackWithMotionModel (:1676-1885)
  int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono);

  // Tracking::Relocalization (:1887-2010)
  int SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const std::set<MapPoint*>& sAlreadyFound, const float th, const int ORBdist);

  // LoopClosing, with a similarity transformation (:427-532, :534-646)
  int SearchByProjection(KeyFrame* pKF, Sophus::Sim3<float>& Scw, const std::vector<MapPoint*>& vpPoints, std::vector<MapPoint*>& vpMatched,
                         int th, float ratioHamming = 1.0);
  int SearchByProjection(KeyFrame* pKF, Sophus::Sim3<float>& Scw, const std::vector<MapPoint*>& vpPoints,
                         const std::vector<KeyFrame*>& vpPointsKFs, std::vector<MapPoint*>& vpMatched, std::vector<KeyFrame*>& vpMatchedKF,
                         int th, float ratioHamming = 1.0);

  // matching inside vocabulary nodes (:223-425, :765-905)
  int SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches);
  int SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12);

  // Tracking::MonocularInitialization (:648-763)
  int SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12, int windowSize = 10);

  // LocalMapping::CreateNewMapPoints (:907-1146)
  int SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<std::pair<size_t, size_t> >& vMatchedPairs, const bool bOnlyStereo,
                             const bool bCoarse = false);

  // (:1457-1674)
  int SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12, const Sophus::Sim3f& S12, const float th);

  // LocalMapping::SearchInNeighbors (:1148-1338) and LoopClosing::SearchAndFuse (:1340-1455)
  int Fuse(KeyFrame* pKF, const std::vector<MapPoint*>& vpMapPoints, const float th = 3.0, const bool bRight = false);
  int Fuse(KeyFrame* pKF, Sophus::Sim3f& Scw, const std::vector<MapPoint*>& vpPoints, float th, std::vector<MapPoint*>& vpReplacePoint);

 public:
  static const int TH_LOW;
  static const int TH_HIGH;
  static const int HISTO_LENGTH;
  EIGEN_MAKE_ALIGNED_OPERATOR_NEW

  // ---- not in the reference's header ---------------------------------------------------------------------------------
  // The orbx context the stack-constructed matchers of the reference (`ORBmatcher matcher(0.9,true);`) run on: one per
  // THREAD (Tracking, LocalMapping and LoopClosing match concurrently and a context serves one caller at a time), created
  // on first use on the thread's current HIP device, destroyed with the thread.  Throws std::runtime_error without a GPU.
  static orbx_ctx* DefaultContext();
  // cv::BFMatcher(NORM_HAMMING).knnMatch(query, train, k = 2) as Frame::ComputeStereoFishEyeMatches uses it
  // (src/Frame.cc:43,1144): idx / dist hold 2 entries per query (-1 / 256 when there are fewer than 2 train rows).
  static void KnnMatch2(const cv::Mat& queryDesc, const cv::Mat& trainDesc, std::vector<int>& idx, std::vector<int>& dist);
  // Resident search targets of the calling thread.  A cached target is only used when the 64-bit digest of the
  // object's keypoints / descriptors / mvuRight still matches, so recycled ids (Tracking::Reset(), src/Tracking.cc:3819-3820) are
  // safe without any call; InvalidateTargets() drops the calling thread's cache explicitly (optional hook for Reset paths),
  // RecycledIdsSeen() counts how often a recycled (id, count, address) with different contents was met on this thread.
  static void InvalidateTargets();
  static unsigned long RecycledIdsSeen();

 protected:
  float RadiusByViewingCos(const float& viewCos);
  void ComputeThreeMaxima(std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3);

  float mfNNratio;
  bool mbCheckOrientation;
};

}  // namespace ORB_SLAM3

#endif  // ORBMATCHER_H
