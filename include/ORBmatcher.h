/* orbx adapter — the Frame-independent part of the reference's include/ORBmatcher.h (:36-103): constructor,
 * the public thresholds, static DescriptorDistance, ComputeThreeMaxima, plus the GPU candidate-list searches that
 * replace the inner `for candidates: DescriptorDistance(...)` loops of the 12 Search.../Fuse routines
 * (src/ORBmatcher.cc; per-routine tie and accept rules: SURVEY.md §3.3).
 *
 * SearchForInitialization (the heaviest Hamming workload of monocular tracking, src/ORBmatcher.cc:648-763) and
 * SearchByProjection(Frame&, vector<MapPoint*>&, ...) (the per-frame local-map search, :43-141) and
 * SearchByProjection(CurrentFrame, LastFrame, th, bMono) (the motion-model search, :1676-1885) and
 * SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches) (reference-keyframe tracking / relocalisation, :223-425) and
 * SearchForTriangulation(pKF1, pKF2, vMatchedPairs, bOnlyStereo, bCoarse) (LocalMapping, :907-1146) are provided in full as
 * templates over the reference's Frame / MapPoint.  The other routines take KeyFrame / Sophus types that belong to the
 * reference and are out of this repository's scope; INTEGRATION.md shows the few-line change that routes
 * each routine's candidate loop through NearestInCandidates() below while the geometry and the greedy bookkeeping
 * stay in src/ORBmatcher.cc.
 */
#ifndef ORBMATCHER_H
#define ORBMATCHER_H

#include <cmath>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "orbx.h"
#include "orbx_cv_compat.h"

namespace ORB_SLAM3 {

class ORBmatcher {
 public:
  ORBmatcher(float nnratio = 0.6, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

  // src/ORBmatcher.cc:2058-2074: 256-bit Hamming distance of two 1x32 CV_8U rows
  static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b) {
    return orbx_hamming(a.ptr<unsigned char>(), b.ptr<unsigned char>());
  }

  // src/ORBmatcher.cc:35-37
  static constexpr int TH_LOW = 50;
  static constexpr int TH_HIGH = 100;
  static constexpr int HISTO_LENGTH = 30;

  // Result of one query of a candidate-list search.
  struct Nearest { int bestIdx, bestDist, secondIdx, secondDist; };

  // The shared inner loop of the Search.../Fuse routines on the GPU: query q is compared with the train rows
  // cand[rowPtr[q] .. rowPtr[q+1]); ties resolve to the FIRST candidate (strict `<`, e.g. src/ORBmatcher.cc:103-111)
  // or, with lastWins, to the LAST one (SearchForTriangulation's `<=`, :1017).  allDist (optional) receives every
  // candidate distance for the routines whose greedy bookkeeping must be replayed on the host in query order.
  static std::vector<Nearest> NearestInCandidates(orbx_ctx* ctx, const cv::Mat& queryDesc, const cv::Mat& trainDesc,
                                                  const std::vector<int>& rowPtr, const std::vector<int>& cand,
                                                  bool lastWins = false, std::vector<int>* allDist = nullptr) {
    const int nq = queryDesc.rows, nt = trainDesc.rows;
    if (!queryDesc.isContinuous() || !trainDesc.isContinuous()) throw std::runtime_error("descriptor matrices must be continuous");
    std::vector<int32_t> bi(nq), bd(nq), si(nq), sd(nq);
    if (allDist) allDist->resize(cand.size());
    const int rc = orbx_nn_csr(ctx, queryDesc.data, nq, trainDesc.data, nt, rowPtr.data(), cand.data(), lastWins ? 1 : 0, bi.data(),
                               bd.data(), si.data(), sd.data(), allDist ? allDist->data() : nullptr);
    if (rc != ORBX_OK) throw std::runtime_error(std::string("ORBmatcher: ") + orbx_last_error(ctx));
    std::vector<Nearest> out(nq);
    for (int q = 0; q < nq; q++) out[q] = Nearest{bi[q], bd[q], si[q], sd[q]};
    return out;
  }

  // cv::BFMatcher(NORM_HAMMING).knnMatch(query, train, k = 2) as used by Frame::ComputeStereoFishEyeMatches
  // (src/Frame.cc:43,1144): idx / dist hold 2 entries per query (-1 / 256 when there are fewer than 2 train rows).
  static void KnnMatch2(orbx_ctx* ctx, const cv::Mat& queryDesc, const cv::Mat& trainDesc, std::vector<int>& idx,
                        std::vector<int>& dist) {
    idx.assign((size_t)queryDesc.rows * 2, -1);
    dist.assign((size_t)queryDesc.rows * 2, 256);
    if (queryDesc.rows == 0) return;
    const int rc = orbx_knn2_allpairs(ctx, queryDesc.data, queryDesc.rows, trainDesc.data, trainDesc.rows, idx.data(), dist.data());
    if (rc != ORBX_OK) throw std::runtime_error(std::string("ORBmatcher: ") + orbx_last_error(ctx));
  }

  // The context of the stack-constructed matchers of the reference (`ORBmatcher matcher(0.9, true);`,
  // src/Tracking.cc:2494).  One per THREAD: Tracking, LocalMapping and LoopClosing run matchers concurrently, and an orbx
  // context (stream, scratch arena) serves one caller at a time.  Created on first use on the thread's current HIP device,
  // destroyed when the thread ends.
  static orbx_ctx* DefaultContext() {
    struct Holder {
      orbx_ctx* c = nullptr;
      ~Holder() { if (c) orbx_destroy(c); }
    };
    static thread_local Holder h;
    if (!h.c && orbx_create(&h.c, 1, 1.2f, 1, 20, 7, -1) != ORBX_OK) {
      h.c = nullptr;
      throw std::runtime_error("ORBmatcher: no MI355X / HIP device");
    }
    return h.c;
  }

  // Matching for the map initialisation (monocular), src/ORBmatcher.cc:648-763.  FrameT is the reference's Frame (or
  // anything with mvKeysUn, mDescriptors and the static image bounds mnMinX / mnMinY / mnMaxX / mnMaxY): candidate
  // windows (Frame::GetFeaturesInArea) and all Hamming distances run on the GPU, the greedy assignment is replayed on
  // the host in the reference's order.  Same arguments, same return value, vbPrevMatched updated in place.
  template <class FrameT>
  int SearchForInitialization(FrameT& F1, FrameT& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12,
                              int windowSize = 10) {
    const int n1 = (int)F1.mvKeysUn.size(), n2 = (int)F2.mvKeysUn.size();
    vnMatches12.assign(n1, -1);
    if (n1 == 0) return 0;
    if (!F1.mDescriptors.isContinuous() || (n2 && !F2.mDescriptors.isContinuous()))
      throw std::runtime_error("descriptor matrices must be continuous");
    static_assert(sizeof(cv::KeyPoint) == sizeof(orbx_keypoint) && sizeof(cv::Point2f) == 8, "POD layouts");
    int nmatches = 0;
    const int rc = orbx_search_for_initialization(
        DefaultContext(), (const orbx_keypoint*)F1.mvKeysUn.data(), F1.mDescriptors.data, n1, (const orbx_keypoint*)F2.mvKeysUn.data(),
        n2 ? F2.mDescriptors.data : nullptr, n2, FrameT::mnMinX, FrameT::mnMinY, FrameT::mnMaxX, FrameT::mnMaxY,
        (float*)vbPrevMatched.data(), windowSize, mfNNratio, mbCheckOrientation ? 1 : 0, vnMatches12.data(), &nmatches);
    if (rc != ORBX_OK) throw std::runtime_error(std::string("ORBmatcher::SearchForInitialization: ") + orbx_last_error(DefaultContext()));
    return nmatches;
  }

  // Tracking::SearchLocalPoints' projection search, src/ORBmatcher.cc:43-141 (same arguments, same return value, same
  // F.mvpMapPoints afterwards).  FrameT / MapPointT are the reference's Frame / MapPoint (only the members the routine
  // reads are used).  Windows, candidate gates and every Hamming distance run on the GPU in one call, the greedy part
  // is replayed in the reference's order.  Two-camera rigs (F.Nleft != -1, :143-210) are not covered: they throw.
  template <class FrameT, class MapPointT>
  int SearchByProjection(FrameT& F, const std::vector<MapPointT*>& vpMapPoints, const float th = 3, const bool bFarPoints = false,
                         const float thFarPoints = 50.0f) {
    if (F.Nleft != -1) throw std::runtime_error("ORBmatcher::SearchByProjection: two-camera frames are not routed to the GPU");
    const int n = (int)F.mvKeysUn.size(), nmp = (int)vpMapPoints.size();
    if (n == 0 || nmp == 0) return 0;
    if (!F.mDescriptors.isContinuous()) throw std::runtime_error("descriptor matrix must be continuous");
    std::vector<int32_t> kpObs(n, -1), kpMatch(n, -1), lvl(nmp, 0), obs(nmp, 0);
    for (int i = 0; i < n; i++)
      if (F.mvpMapPoints[i]) kpObs[i] = F.mvpMapPoints[i]->Observations();
    std::vector<unsigned char> inView(nmp, 0), mpDesc((size_t)nmp * 32, 0);
    std::vector<float> px(nmp, 0.f), py(nmp, 0.f), pxr(nmp, 0.f), vc(nmp, 0.f);
    for (int i = 0; i < nmp; i++) {
      MapPointT* pMP = vpMapPoints[i];
      if (!pMP->mbTrackInView) continue;                          // :52-53 (mbTrackInViewR belongs to the two-camera block)
      if (bFarPoints && pMP->mTrackDepth > thFarPoints) continue;  // :55-56
      if (pMP->isBad()) continue;                                  // :58-59
      inView[i] = 1;
      px[i] = pMP->mTrackProjX; py[i] = pMP->mTrackProjY; pxr[i] = pMP->mTrackProjXR; vc[i] = pMP->mTrackViewCos;
      lvl[i] = pMP->mnTrackScaleLevel; obs[i] = pMP->Observations();
      const cv::Mat d = pMP->GetDescriptor();
      std::memcpy(&mpDesc[(size_t)i * 32], d.template ptr<unsigned char>(), 32);
    }
    const bool stereo = !F.mvuRight.empty();
    int nmatches = 0;
    const int rc = orbx_search_by_projection(
        DefaultContext(), (const orbx_keypoint*)F.mvKeysUn.data(), F.mDescriptors.data, stereo ? F.mvuRight.data() : nullptr, kpObs.data(), n,
        FrameT::mnMinX, FrameT::mnMinY, FrameT::mnMaxX, FrameT::mnMaxY, F.mvScaleFactors.data(), (int)F.mvScaleFactors.size(), inView.data(),
        px.data(), py.data(), pxr.data(), vc.data(), lvl.data(), mpDesc.data(), obs.data(), nmp, th, mfNNratio, kpMatch.data(), &nmatches);
    if (rc != ORBX_OK) throw std::runtime_error(std::string("ORBmatcher::SearchByProjection: ") + orbx_last_error(DefaultContext()));
    for (int i = 0; i < n; i++)
      if (kpMatch[i] >= 0) F.mvpMapPoints[i] = vpMapPoints[kpMatch[i]];
    return nmatches;
  }

  // Tracking::TrackWithMotionModel's projection search, src/ORBmatcher.cc:1676-1885 (same arguments, same return value,
  // same CurrentFrame.mvpMapPoints afterwards).  The pose / camera arithmetic (:1686-1718) runs here through the
  // reference's own types (Sophus::SE3f, GeometricCamera — whatever FrameT provides); windows, gates and Hamming
  // distances run on the GPU, the greedy part and the rotation filter are replayed in the reference's order.
  template <class FrameT>
  int SearchByProjection(FrameT& CurrentFrame, const FrameT& LastFrame, const float th, const bool bMono) {
    if (CurrentFrame.Nleft != -1 || LastFrame.Nleft != -1)
      throw std::runtime_error("ORBmatcher::SearchByProjection: two-camera frames are not routed to the GPU");
    const auto Tcw = CurrentFrame.GetPose();
    const auto twc = Tcw.inverse().translation();
    const auto Tlw = LastFrame.GetPose();
    const auto tlc = Tlw * twc;
    const bool bForward = tlc(2) > CurrentFrame.mb && !bMono;
    const bool bBackward = -tlc(2) > CurrentFrame.mb && !bMono;
    const int n = (int)CurrentFrame.mvKeysUn.size(), nlast = LastFrame.N;
    if (n == 0 || nlast == 0) return 0;
    if (!CurrentFrame.mDescriptors.isContinuous()) throw std::runtime_error("descriptor matrix must be continuous");
    std::vector<unsigned char> valid(nlast, 0), lpDesc((size_t)nlast * 32, 0);
    std::vector<float> u(nlast, 0.f), v(nlast, 0.f), invz(nlast, 0.f), ang(nlast, 0.f);
    std::vector<int32_t> oct(nlast, 0), obs(nlast, 0), kpObs(n, -1), kpMatch(n, -1);
    for (int i = 0; i < nlast; i++) {
      auto* pMP = LastFrame.mvpMapPoints[i];
      if (!pMP || LastFrame.mvbOutlier[i]) continue;
      const auto x3Dw = pMP->GetWorldPos();
      const auto x3Dc = Tcw * x3Dw;
      const float invzc = 1.0 / x3Dc(2);
      if (invzc < 0) continue;
      const auto uv = CurrentFrame.mpCamera->project(x3Dc);
      if (uv(0) < FrameT::mnMinX || uv(0) > FrameT::mnMaxX) continue;
      if (uv(1) < FrameT::mnMinY || uv(1) > FrameT::mnMaxY) continue;
      valid[i] = 1; u[i] = uv(0); v[i] = uv(1); invz[i] = invzc;
      oct[i] = LastFrame.mvKeys[i].octave; ang[i] = LastFrame.mvKeysUn[i].angle; obs[i] = pMP->Observations();
      const cv::Mat d = pMP->GetDescriptor();
      std::memcpy(&lpDesc[(size_t)i * 32], d.template ptr<unsigned char>(), 32);
    }
    for (int i = 0; i < n; i++)
      if (CurrentFrame.mvpMapPoints[i]) kpObs[i] = CurrentFrame.mvpMapPoints[i]->Observations();
    const bool stereo = !CurrentFrame.mvuRight.empty();
    int nmatches = 0;
    const int rc = orbx_search_by_projection_last(
        DefaultContext(), (const orbx_keypoint*)CurrentFrame.mvKeysUn.data(), CurrentFrame.mDescriptors.data,
        stereo ? CurrentFrame.mvuRight.data() : nullptr, kpObs.data(), n, FrameT::mnMinX, FrameT::mnMinY, FrameT::mnMaxX, FrameT::mnMaxY,
        CurrentFrame.mvScaleFactors.data(), (int)CurrentFrame.mvScaleFactors.size(), CurrentFrame.mbf, valid.data(), u.data(), v.data(),
        invz.data(), oct.data(), ang.data(), lpDesc.data(), obs.data(), nlast, th, bForward ? 1 : (bBackward ? 2 : 0),
        mbCheckOrientation ? 1 : 0, kpMatch.data(), &nmatches);
    if (rc != ORBX_OK) throw std::runtime_error(std::string("ORBmatcher::SearchByProjection: ") + orbx_last_error(DefaultContext()));
    for (int i = 0; i < n; i++) {
      if (kpMatch[i] >= 0) CurrentFrame.mvpMapPoints[i] = LastFrame.mvpMapPoints[kpMatch[i]];
      else if (kpMatch[i] == -2) CurrentFrame.mvpMapPoints[i] = nullptr;
    }
    return nmatches;
  }

  // Matching by vocabulary node (TrackReferenceKeyFrame, Relocalization), src/ORBmatcher.cc:223-425 — same arguments, same
  // return value, same vpMapPointMatches.  KeyFrameT / FrameT / MapPointT are the reference's types (mFeatVec is a
  // DBoW2::FeatureVector, i.e. an ordered map node -> feature indices).  Single-camera only: a two-camera rig throws.
  template <class KeyFrameT, class FrameT, class MapPointT>
  int SearchByBoW(KeyFrameT* pKF, FrameT& F, std::vector<MapPointT*>& vpMapPointMatches) {
    if (F.Nleft != -1 || pKF->mpCamera2)
      throw std::runtime_error("ORBmatcher::SearchByBoW: two-camera frames are not routed to the GPU");
    const std::vector<MapPointT*> vpMapPointsKF = pKF->GetMapPointMatches();
    vpMapPointMatches = std::vector<MapPointT*>(F.N, static_cast<MapPointT*>(nullptr));
    const int nkf = (int)vpMapPointsKF.size(), nf = F.N;
    if (nkf == 0 || nf == 0) return 0;
    if (!pKF->mDescriptors.isContinuous() || !F.mDescriptors.isContinuous()) throw std::runtime_error("descriptor matrices must be continuous");
    std::vector<unsigned char> valid(nkf, 0);
    std::vector<float> kfAngle(nkf), fAngle(nf);
    for (int i = 0; i < nkf; i++) {
      valid[i] = vpMapPointsKF[i] && !vpMapPointsKF[i]->isBad();
      kfAngle[i] = pKF->mvKeysUn[i].angle;
    }
    for (int i = 0; i < nf; i++) fAngle[i] = F.mvKeys[i].angle;
    auto flatten = [](const auto& fv, std::vector<uint32_t>& node, std::vector<int32_t>& ptr, std::vector<uint32_t>& idx) {
      ptr.push_back(0);
      for (const auto& kv : fv) {
        node.push_back((uint32_t)kv.first);
        for (unsigned f : kv.second) idx.push_back(f);
        ptr.push_back((int32_t)idx.size());
      }
    };
    std::vector<uint32_t> kn, ki, fn, fi;
    std::vector<int32_t> kp, fp;
    flatten(pKF->mFeatVec, kn, kp, ki);
    flatten(F.mFeatVec, fn, fp, fi);
    std::vector<int32_t> match(nf, -1);
    int nmatches = 0;
    const int rc = orbx_search_by_bow(DefaultContext(), pKF->mDescriptors.data, kfAngle.data(), valid.data(), nkf, kn.data(), kp.data(),
                                      ki.data(), (int)kn.size(), F.mDescriptors.data, fAngle.data(), nf, fn.data(), fp.data(), fi.data(),
                                      (int)fn.size(), mfNNratio, mbCheckOrientation ? 1 : 0, match.data(), &nmatches);
    if (rc != ORBX_OK) throw std::runtime_error(std::string("ORBmatcher::SearchByBoW: ") + orbx_last_error(DefaultContext()));
    for (int i = 0; i < nf; i++)
      if (match[i] >= 0) vpMapPointMatches[i] = vpMapPointsKF[match[i]];
    return nmatches;
  }

  // LocalMapping::CreateNewMapPoints' matcher, src/ORBmatcher.cc:907-1146 (same arguments, same return value, same
  // vMatchedPairs) for single-camera keyframes.  Every Hamming distance between features of the same vocabulary node is
  // computed on the GPU in one launch; the epipole test and pCamera1->epipolarConstrain(...) run here through the
  // reference's own camera objects, only for the candidates the distance tests let through, in the reference's order.
  template <class KeyFrameT>
  int SearchForTriangulation(KeyFrameT* pKF1, KeyFrameT* pKF2, std::vector<std::pair<size_t, size_t> >& vMatchedPairs,
                             const bool bOnlyStereo, const bool bCoarse = false) {
    if (pKF1->mpCamera2 || pKF2->mpCamera2)
      throw std::runtime_error("ORBmatcher::SearchForTriangulation: two-camera keyframes are not routed to the GPU");
    const auto T1w = pKF1->GetPose();
    const auto T2w = pKF2->GetPose();
    const auto Tw2 = pKF2->GetPoseInverse();
    const auto Cw = pKF1->GetCameraCenter();
    const auto C2 = T2w * Cw;
    const auto ep = pKF2->mpCamera->project(C2);
    const auto T12 = T1w * Tw2;
    const auto R12 = T12.rotationMatrix();
    const auto t12 = T12.translation();
    auto* pCamera1 = pKF1->mpCamera;
    auto* pCamera2 = pKF2->mpCamera;
    const int n1 = pKF1->N, n2 = pKF2->N;
    vMatchedPairs.clear();
    if (n1 == 0 || n2 == 0) return 0;
    if (!pKF1->mDescriptors.isContinuous() || !pKF2->mDescriptors.isContinuous()) throw std::runtime_error("descriptor matrices must be continuous");
    // rows: features of KF1 without a map point (and stereo ones only, if asked), in the reference's visiting order;
    // candidates: the features of KF2 in the same node that have no map point either (:968-996)
    std::vector<int> q1;
    std::vector<int32_t> rowPtr(1, 0), cand;
    std::vector<unsigned char> qd;
    auto f1it = pKF1->mFeatVec.begin(), f1end = pKF1->mFeatVec.end();
    auto f2it = pKF2->mFeatVec.begin(), f2end = pKF2->mFeatVec.end();
    while (f1it != f1end && f2it != f2end) {
      if (f1it->first == f2it->first) {
        for (size_t i1 = 0; i1 < f1it->second.size(); i1++) {
          const size_t idx1 = f1it->second[i1];
          if (pKF1->GetMapPoint(idx1)) continue;
          const bool bStereo1 = pKF1->mvuRight[idx1] >= 0;
          if (bOnlyStereo && !bStereo1) continue;
          q1.push_back((int)idx1);
          const unsigned char* d = pKF1->mDescriptors.template ptr<unsigned char>((int)idx1);
          qd.insert(qd.end(), d, d + 32);
          for (size_t i2 = 0; i2 < f2it->second.size(); i2++) {
            const size_t idx2 = f2it->second[i2];
            if (pKF2->GetMapPoint(idx2)) continue;
            if (bOnlyStereo && !(pKF2->mvuRight[idx2] >= 0)) continue;
            cand.push_back((int32_t)idx2);
          }
          rowPtr.push_back((int32_t)cand.size());
        }
        ++f1it; ++f2it;
      } else if (f1it->first < f2it->first) {
        f1it = pKF1->mFeatVec.lower_bound(f2it->first);
      } else {
        f2it = pKF2->mFeatVec.lower_bound(f1it->first);
      }
    }
    const int nq = (int)q1.size();
    std::vector<int32_t> dist(cand.size());
    if (nq && !cand.empty()) {
      const int rc = orbx_nn_csr(DefaultContext(), qd.data(), nq, pKF2->mDescriptors.data, n2, rowPtr.data(), cand.data(), 1, nullptr,
                                 nullptr, nullptr, nullptr, dist.data());
      if (rc != ORBX_OK) throw std::runtime_error(std::string("ORBmatcher::SearchForTriangulation: ") + orbx_last_error(DefaultContext()));
    }
    int nmatches = 0;
    std::vector<int> vMatches12(n1, -1);
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    for (int q = 0; q < nq; q++) {
      const size_t idx1 = (size_t)q1[q];
      const bool bStereo1 = pKF1->mvuRight[idx1] >= 0;
      const cv::KeyPoint& kp1 = pKF1->mvKeysUn[idx1];
      int bestDist = TH_LOW, bestIdx2 = -1;
      for (int c = rowPtr[q]; c < rowPtr[q + 1]; c++) {
        const size_t idx2 = (size_t)cand[c];
        const int d = dist[c];
        if (d > TH_LOW || d > bestDist) continue;
        const bool bStereo2 = pKF2->mvuRight[idx2] >= 0;
        const cv::KeyPoint& kp2 = pKF2->mvKeysUn[idx2];
        if (!bStereo1 && !bStereo2) {
          const float distex = ep(0) - kp2.pt.x;
          const float distey = ep(1) - kp2.pt.y;
          if (distex * distex + distey * distey < 100 * pKF2->mvScaleFactors[kp2.octave]) continue;
        }
        if (bCoarse || pCamera1->epipolarConstrain(pCamera2, kp1, kp2, R12, t12, pKF1->mvLevelSigma2[kp1.octave], pKF2->mvLevelSigma2[kp2.octave])) {
          bestIdx2 = (int)idx2;
          bestDist = d;
        }
      }
      if (bestIdx2 >= 0) {
        const cv::KeyPoint& kp2 = pKF2->mvKeysUn[bestIdx2];
        vMatches12[idx1] = bestIdx2;
        nmatches++;
        if (mbCheckOrientation) {
          float rot = kp1.angle - kp2.angle;
          if (rot < 0.0) rot += 360.0f;
          int bin = (int)std::round(rot * factor);
          if (bin == HISTO_LENGTH) bin = 0;
          if (bin >= 0 && bin < HISTO_LENGTH) rotHist[bin].push_back((int)idx1);
        }
      }
    }
    if (mbCheckOrientation) {
      int ind1 = -1, ind2 = -1, ind3 = -1;
      ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
      for (int i = 0; i < HISTO_LENGTH; i++) {
        if (i == ind1 || i == ind2 || i == ind3) continue;
        for (size_t j = 0; j < rotHist[i].size(); j++) { vMatches12[rotHist[i][j]] = -1; nmatches--; }
      }
    }
    vMatchedPairs.reserve(nmatches);
    for (size_t i = 0; i < vMatches12.size(); i++)
      if (vMatches12[i] >= 0) vMatchedPairs.push_back(std::make_pair(i, (size_t)vMatches12[i]));
    return nmatches;
  }

  // src/ORBmatcher.cc:2012-2053 (public here so the host-side replays can use it)
  static void ComputeThreeMaxima(std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3) {
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
      const int s = (int)histo[i].size();
      if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
      else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
      else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
  }

 protected:
  float mfNNratio;
  bool mbCheckOrientation;
};

}  // namespace ORB_SLAM3

#endif  // ORBMATCHER_H
