// TEST / BENCH INFRASTRUCTURE: the synthetic world shared by tests/support/matcher_world.cpp (scenario driver of the 12 matcher
// routines) and tools/streamed_frontend.cpp (the per-frame Tracking sequence timed by bench.py): views of one synthetic stream
// (keypoints, descriptors, feature vectors), map points with the gates' fields, Frames / KeyFrames of tests/support/ref_world/
// filled the way src/Frame.cc / src/KeyFrame.cc fill them.  Compiles against either ORBmatcher.h (the reference's or the drop-in).
#ifndef ORBX_WORLD_SCENE_H
#define ORBX_WORLD_SCENE_H
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

#include "ORBmatcher.h"

using namespace ORB_SLAM3;

namespace {

struct View {
  int n = 0;
  std::vector<cv::KeyPoint> kps;
  cv::Mat desc;
  DBoW2::FeatureVector fv;
};

struct World {
  int rows = 0, cols = 0, nlevels = 0;
  std::vector<float> scale, sigma2, inv_sigma2;
  float scaleFactor = 1.2f, logScaleFactor = 0;
  std::vector<View> views;
};

bool load_world(const char* path, World& w) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  int32_t hdr[4];
  f.read((char*)hdr, sizeof(hdr));
  if (hdr[0] != 0x0b5e55ed) return false;
  w.rows = hdr[1]; w.cols = hdr[2]; w.nlevels = hdr[3];
  w.scale.resize(w.nlevels); w.sigma2.resize(w.nlevels); w.inv_sigma2.resize(w.nlevels);
  f.read((char*)w.scale.data(), 4 * w.nlevels);
  f.read((char*)w.sigma2.data(), 4 * w.nlevels);
  f.read((char*)w.inv_sigma2.data(), 4 * w.nlevels);
  f.read((char*)&w.scaleFactor, 4);
  f.read((char*)&w.logScaleFactor, 4);
  int32_t nviews = 0;
  f.read((char*)&nviews, 4);
  w.views.resize(nviews);
  for (View& v : w.views) {
    int32_t n = 0;
    f.read((char*)&n, 4);
    v.n = n;
    v.kps.resize(n);
    static_assert(sizeof(cv::KeyPoint) == 28, "KeyPoint layout");
    f.read((char*)v.kps.data(), (std::streamsize)n * 28);
    v.desc = cv::Mat(n, 32, CV_8U);
    f.read((char*)v.desc.data, (std::streamsize)n * 32);
    int32_t nfv = 0;
    f.read((char*)&nfv, 4);
    for (int i = 0; i < nfv; i++) {
      uint32_t p[2];
      f.read((char*)p, 8);
      v.fv[p[0]].push_back(p[1]);
    }
  }
  return (bool)f;
}

inline uint32_t H(uint32_t i, uint32_t salt) {   // small integer hash: all "random" choices of the scenarios
  uint32_t x = i * 2654435761u + salt * 40503u + 0x9e3779b9u;
  x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
  return x;
}

Eigen::Matrix3f rot_yx(float ay, float ax) {
  Eigen::Matrix3f Ry, Rx;
  const float cy = std::cos(ay), sy = std::sin(ay), cx = std::cos(ax), sx = std::sin(ax);
  Ry(0, 0) = cy; Ry(0, 2) = sy; Ry(2, 0) = -sy; Ry(2, 2) = cy;
  Rx(1, 1) = cx; Rx(1, 2) = -sx; Rx(2, 1) = sx; Rx(2, 2) = cx;
  return Ry * Rx;
}

struct Cams {
  TestPinhole pinL, pinR;
  TestFisheye fishL, fishR;
};

// Everything a scenario owns; rebuilt from scratch for every scenario so that the scenarios are independent.
struct Scene {
  const World& w;
  Cams cams;
  std::vector<MapPoint> mps;     // map points seen from view 0 (world frame = camera frame of view 0)
  std::vector<MapPoint> mpsB;    // map points created from another view (keyframe-2 side of the two-keyframe routines)
  std::vector<MapPoint> extra;   // points bound to frames before a call ("already there")
  std::vector<KeyFrame*> kfs;
  // KeyFrame::nNextId / Frame::nNextId are process-wide statics in the reference: ids never repeat, also across scenes
  static unsigned long nextKfId, nextFrameId;
  static unsigned long kfIdBase;   // first keyframe id of this scene: results print ids relative to it (a scene may be run many times)

  explicit Scene(const World& w_, bool distorted) : w(w_) {
    kfIdBase = nextKfId;
    for (TestPinhole* c : {(TestPinhole*)&cams.pinL, (TestPinhole*)&cams.pinR, (TestPinhole*)&cams.fishL, (TestPinhole*)&cams.fishR}) {
      c->fx = 458.f; c->fy = 457.f; c->cx = 0.5f * w.cols + 3.5f; c->cy = 0.5f * w.rows - 2.25f;
    }
    // image bounds (src/Frame.cc:153-160): non-integer when the image is undistorted (ComputeImageBounds)
    Frame::mnMinX = distorted ? -12.7f : 0.f;
    Frame::mnMinY = distorted ? -9.4f : 0.f;
    Frame::mnMaxX = distorted ? (float)w.cols + 13.2f : (float)w.cols;
    Frame::mnMaxY = distorted ? (float)w.rows + 8.9f : (float)w.rows;
    Frame::mfGridElementWidthInv = static_cast<float>(FRAME_GRID_COLS) / static_cast<float>(Frame::mnMaxX - Frame::mnMinX);
    Frame::mfGridElementHeightInv = static_cast<float>(FRAME_GRID_ROWS) / static_cast<float>(Frame::mnMaxY - Frame::mnMinY);
  }
  ~Scene() { for (KeyFrame* k : kfs) delete k; }

  // pose of the camera that took view t: the stream shifts its content by (-1.5 t, -0.5 t) px, i.e. a small rotation
  Sophus::SE3f pose(int t, float tx = 0.f, float ty = 0.f, float tz = 0.f) const {
    return Sophus::SE3f(rot_yx(-1.5f * t / 458.f, 0.5f * t / 457.f), Eigen::Vector3f(tx, ty, tz));
  }
  // left-to-right transform of the two-camera rig: the right view is 2 stream steps further plus a small baseline
  Sophus::SE3f trl() const { return Sophus::SE3f(rot_yx(-1.5f * 2 / 458.f, 0.5f * 2 / 457.f), Eigen::Vector3f(-0.02f, 0.001f, 0.003f)); }

  float depth_of(int i, int salt) const { return 2.5f + (float)(H(i, 100 + salt) % 1000) / 250.f; }

  // map points from the keypoints of view `v`, seen by a camera at `Tcw`
  void make_points(std::vector<MapPoint>& out, int v, const Sophus::SE3f& Tcw, int idBase, int salt) {
    const View& V = w.views[v];
    out.assign(V.n, MapPoint());
    const Sophus::SE3f Twc = Tcw.inverse();
    const Eigen::Vector3f Ow = Twc.translation();
    for (int i = 0; i < V.n; i++) {
      MapPoint& m = out[i];
      m.mnId = idBase + i;
      const float z = depth_of(i, salt);
      const Eigen::Vector3f pc((V.kps[i].pt.x - cams.pinL.cx) / cams.pinL.fx * z, (V.kps[i].pt.y - cams.pinL.cy) / cams.pinL.fy * z, z);
      m.mWorldPos = Twc * pc;
      Eigen::Vector3f PO = m.mWorldPos - Ow;
      const float dist = PO.norm();
      m.mNormalVector = PO / dist;
      if (H(i, 7 + salt) % 17 == 0) m.mNormalVector = -m.mNormalVector;             // fails the viewing-angle test
      m.mDescriptor = V.desc.row(i).clone();
      const int level = V.kps[i].octave;
      m.mfMaxDistance = dist * w.scale[level];                                        // src/MapPoint.cc UpdateNormalAndDepth
      m.mfMinDistance = m.mfMaxDistance / w.scale[w.nlevels - 1];
      if (H(i, 8 + salt) % 29 == 0) { m.mfMaxDistance *= 0.3f; m.mfMinDistance *= 0.3f; }   // outside the invariance region
      m.nObs = (H(i, 9 + salt) % 3 == 0) ? 0 : 2;
      m.mbBad = H(i, 10 + salt) % 23 == 0;
    }
  }

  void fill_common(Frame& F) {
    F.mnScaleLevels = w.nlevels; F.mfScaleFactor = w.scaleFactor; F.mfLogScaleFactor = w.logScaleFactor;
    F.mvScaleFactors = w.scale; F.mvLevelSigma2 = w.sigma2; F.mvInvLevelSigma2 = w.inv_sigma2;
    F.mb = 0.11f; F.mbf = 0.11f * 458.f;
  }

  // single-camera frame from view v; stereo: every 4th keypoint monocular, the others with a right coordinate
  void make_frame(Frame& F, int v, bool stereo, const Sophus::SE3f& Tcw) {
    const View& V = w.views[v];
    fill_common(F);
    F.mnId = nextFrameId++;   // Frame::Frame: mnId = nNextId++ (src/Frame.cc)
    F.N = V.n; F.Nleft = -1; F.Nright = -1;
    F.mvKeys = V.kps; F.mvKeysUn = V.kps; F.mDescriptors = V.desc.clone(); F.mFeatVec = V.fv;
    F.mvpMapPoints.assign(V.n, static_cast<MapPoint*>(NULL));
    F.mvbOutlier.assign(V.n, false);
    F.mvuRight.assign(V.n, -1.f);
    if (stereo)
      for (int i = 0; i < V.n; i++)
        if (i % 4 != 0) F.mvuRight[i] = V.kps[i].pt.x - F.mbf / depth_of(i, 0) + 0.25f * (float)((int)(H(i, 11) % 9) - 4);
    F.mpCamera = &cams.pinL; F.mpCamera2 = nullptr;
    F.mTcw = Tcw;
    F.AssignFeaturesToGrid();
  }

  // two-camera rig frame (src/Frame.cc:1040-1140): left keypoints from view vl, right ones from view vr
  void make_rig_frame(Frame& F, int vl, int vr, const Sophus::SE3f& Tcw) {
    const View &L = w.views[vl], &R = w.views[vr];
    fill_common(F);
    F.mnId = nextFrameId++;
    F.Nleft = L.n; F.Nright = R.n; F.N = L.n + R.n;
    F.mvKeys = L.kps; F.mvKeysRight = R.kps; F.mvKeysUn = L.kps;
    F.mDescriptors = cv::Mat(F.N, 32, CV_8U);
    std::memcpy(F.mDescriptors.data, L.desc.data, (size_t)L.n * 32);
    std::memcpy(F.mDescriptors.data + (size_t)L.n * 32, R.desc.data, (size_t)R.n * 32);
    F.mFeatVec = L.fv;
    for (const auto& kv : R.fv)
      for (unsigned f : kv.second) F.mFeatVec[kv.first].push_back(f + (unsigned)L.n);
    F.mvpMapPoints.assign(F.N, static_cast<MapPoint*>(NULL));
    F.mvbOutlier.assign(F.N, false);
    F.mvuRight.assign(F.N, -1.f);
    F.mvLeftToRightMatch.assign(L.n, -1);
    F.mvRightToLeftMatch.assign(R.n, -1);
    for (int i = 0; i < L.n; i += 9) {
      const int j = (int)(H(i, 12) % (uint32_t)R.n);
      if (F.mvRightToLeftMatch[j] != -1) continue;
      F.mvLeftToRightMatch[i] = j; F.mvRightToLeftMatch[j] = i;
    }
    F.mpCamera = &cams.fishL; F.mpCamera2 = &cams.fishR;
    F.mTcw = Tcw; F.mTrl = trl();
    F.AssignFeaturesToGrid();
  }

  // KeyFrame::KeyFrame(Frame&, ...) (src/KeyFrame.cc:41-80): copies, with the image bounds truncated to int
  KeyFrame* make_keyframe(Frame& F) {
    KeyFrame* K = new KeyFrame();
    kfs.push_back(K);
    K->mnId = nextKfId++;
    K->mfGridElementWidthInv = F.mfGridElementWidthInv; K->mfGridElementHeightInv = F.mfGridElementHeightInv;
    K->fx = cams.pinL.fx; K->fy = cams.pinL.fy; K->cx = cams.pinL.cx; K->cy = cams.pinL.cy;
    K->invfx = 1.f / K->fx; K->invfy = 1.f / K->fy; K->mbf = F.mbf; K->mb = F.mb;
    K->N = F.N;
    K->mvKeys = F.mvKeys; K->mvKeysUn = F.mvKeysUn; K->mvKeysRight = F.mvKeysRight; K->mvuRight = F.mvuRight;
    K->mDescriptors = F.mDescriptors.clone(); K->mFeatVec = F.mFeatVec;
    K->mnScaleLevels = F.mnScaleLevels; K->mfScaleFactor = F.mfScaleFactor; K->mfLogScaleFactor = F.mfLogScaleFactor;
    K->mvScaleFactors = F.mvScaleFactors; K->mvLevelSigma2 = F.mvLevelSigma2; K->mvInvLevelSigma2 = F.mvInvLevelSigma2;
    K->mnMinX = F.mnMinX; K->mnMinY = F.mnMinY; K->mnMaxX = F.mnMaxX; K->mnMaxY = F.mnMaxY;
    K->mpCamera = F.mpCamera; K->mpCamera2 = F.mpCamera2;
    K->NLeft = F.Nleft; K->NRight = F.Nright;
    K->mTcw = F.mTcw; K->mTrl = F.mTrl;
    K->mGrid.resize(K->mnGridCols);
    if (F.Nleft != -1) K->mGridRight.resize(K->mnGridCols);
    for (int i = 0; i < K->mnGridCols; i++) {
      K->mGrid[i].resize(K->mnGridRows);
      if (F.Nleft != -1) K->mGridRight[i].resize(K->mnGridRows);
      for (int j = 0; j < K->mnGridRows; j++) {
        K->mGrid[i][j] = F.mGrid[i][j];
        if (F.Nleft != -1) K->mGridRight[i][j] = F.mGridRight[i][j];
      }
    }
    K->mvpMapPoints.assign(F.N, static_cast<MapPoint*>(NULL));
    return K;
  }

  // bind map point m to feature idx of K, both directions (LocalMapping / Tracking::CreateNewKeyFrame do this)
  static void bind(KeyFrame* K, MapPoint* m, int idx) { K->AddMapPoint(m, idx); m->AddObservation(K, idx); }
};
unsigned long Scene::nextKfId = 1, Scene::nextFrameId = 1, Scene::kfIdBase = 1;

struct Out {
  FILE* f;
  void line(const std::string& name, int ret) { std::fprintf(f, "%s ret=%d\n", name.c_str(), ret); }
  void ints(const char* what, const std::vector<long>& v) {
    std::fprintf(f, "  %s[%zu]:", what, v.size());
    for (long x : v) std::fprintf(f, " %ld", x);
    std::fprintf(f, "\n");
  }
};

long mp_id(MapPoint* p) { return p ? (long)p->mnId : -1; }
std::vector<long> ids_of(const std::vector<MapPoint*>& v) { std::vector<long> r; for (MapPoint* p : v) r.push_back(mp_id(p)); return r; }

void dump_points(Out& o, const char* what, std::vector<MapPoint>& pts) {
  std::vector<long> v;
  for (MapPoint& m : pts) {
    unsigned sum = 0;
    const unsigned char* d = m.mDescriptor.ptr<unsigned char>();
    for (int i = 0; i < 32; i++) sum = sum * 131u + d[i];
    v.push_back(m.mbBad); v.push_back(m.nObs); v.push_back(mp_id(m.mpReplaced)); v.push_back((long)(sum & 0xffffff));
    v.push_back((long)m.mObservations.size());
    for (auto& kv : m.mObservations) { v.push_back((long)(kv.first->mnId - Scene::kfIdBase + 1)); v.push_back(std::get<0>(kv.second)); v.push_back(std::get<1>(kv.second)); }
  }
  o.ints(what, v);
}

// projections the tracking thread stores in the map points before SearchLocalPoints (Frame::isInFrustum, src/Frame.cc:452-560)
void set_track_fields(Scene& s, Frame& F, std::vector<MapPoint>& pts, bool rig) {
  const Sophus::SE3f Tcw = F.GetPose();
  const Eigen::Vector3f Ow = Tcw.inverse().translation();
  for (size_t i = 0; i < pts.size(); i++) {
    MapPoint& m = pts[i];
    const Eigen::Vector3f Pc = Tcw * m.mWorldPos;
    m.mbTrackInView = false; m.mbTrackInViewR = false;
    if (!(Pc(2) > 0.f)) continue;
    const Eigen::Vector2f uv = F.mpCamera->project(Pc);
    const float dist = (m.mWorldPos - Ow).norm();
    const bool in = uv(0) >= Frame::mnMinX && uv(0) <= Frame::mnMaxX && uv(1) >= Frame::mnMinY && uv(1) <= Frame::mnMaxY;
    m.mbTrackInView = in && (H((uint32_t)i, 20) % 5 != 0);
    m.mTrackProjX = uv(0); m.mTrackProjY = uv(1);
    m.mTrackProjXR = uv(0) - F.mbf / Pc(2);
    m.mTrackDepth = dist;
    m.mnTrackScaleLevel = m.PredictScale(dist, &F);
    m.mTrackViewCos = (H((uint32_t)i, 21) % 2) ? 0.9995f : 0.99f;
    if (rig) {
      const Eigen::Vector3f Pr = F.GetRelativePoseTrl() * Pc;
      if (Pr(2) > 0.f) {
        const Eigen::Vector2f uvr = F.mpCamera2->project(Pr);
        const bool inr = uvr(0) >= Frame::mnMinX && uvr(0) <= Frame::mnMaxX && uvr(1) >= Frame::mnMinY && uvr(1) <= Frame::mnMaxY;
        m.mbTrackInViewR = inr && (H((uint32_t)i, 22) % 4 != 0);
        m.mTrackProjXR = uvr(0); m.mTrackProjYR = uvr(1);
        m.mTrackDepthR = Pr.norm();
        m.mnTrackScaleLevelR = (H((uint32_t)i, 23) % 13 == 0) ? -1 : m.PredictScale(m.mTrackDepthR, &F);
        m.mTrackViewCosR = (H((uint32_t)i, 24) % 2) ? 0.9995f : 0.99f;
      }
    }
  }
}

void prebind(Scene& s, Frame& F, int every, int salt) {   // keypoints that already carry a map point before the call
  s.extra.assign(F.N, MapPoint());
  for (int i = 0; i < F.N; i++) {
    s.extra[i].mnId = 900000 + i;
    s.extra[i].nObs = (H(i, salt) % 3 == 0) ? 0 : 4;
    if (i % every == 0) F.mvpMapPoints[i] = &s.extra[i];
  }
}

// ---- id reuse (src/Tracking.cc:3819-3820: Tracking::Reset() sets KeyFrame::nNextId = 0 and Frame::nNextId = 0; Atlas.cc:242 deletes the
// keyframes): a later Frame / KeyFrame carries a recycled id, and the allocator may hand back the same buffers.  The worst case is
// produced deterministically here: the SAME object (same mnId, same count, same mvKeysUn / mDescriptors addresses) is refilled in
// place with the first n features of another view.
void refill_frame(Scene& s, Frame& F, int v, bool stereo) {
  const View& V = s.w.views[v];
  const int n = F.N;
  for (int i = 0; i < n; i++) { F.mvKeys[i] = V.kps[i]; F.mvKeysUn[i] = V.kps[i]; }
  std::memcpy(F.mDescriptors.data, V.desc.data, (size_t)n * 32);
  F.mvpMapPoints.assign(n, static_cast<MapPoint*>(NULL));
  F.mvbOutlier.assign(n, false);
  for (int i = 0; i < n; i++)
    F.mvuRight[i] = (stereo && i % 4 != 0) ? V.kps[i].pt.x - F.mbf / s.depth_of(i, 0) + 0.25f * (float)((int)(H(i, 11) % 9) - 4) : -1.f;
  F.AssignFeaturesToGrid();
}
void truncate_frame(Frame& F, int n) {   // keep the first n features (before any search)
  F.N = n;
  F.mvKeys.resize(n); F.mvKeysUn.resize(n); F.mvpMapPoints.resize(n); F.mvbOutlier.resize(n); F.mvuRight.resize(n);
  cv::Mat d(n, 32, CV_8U);
  std::memcpy(d.data, F.mDescriptors.data, (size_t)n * 32);
  F.mDescriptors = d;
  F.AssignFeaturesToGrid();
}
void refill_keyframe(KeyFrame* K, const Frame& F) {   // K <- F, in place (same id, same buffers)
  for (int i = 0; i < K->N; i++) { K->mvKeys[i] = F.mvKeys[i]; K->mvKeysUn[i] = F.mvKeysUn[i]; K->mvuRight[i] = F.mvuRight[i]; }
  std::memcpy(K->mDescriptors.data, F.mDescriptors.data, (size_t)K->N * 32);
  for (int i = 0; i < K->mnGridCols; i++)
    for (int j = 0; j < K->mnGridRows; j++) K->mGrid[i][j] = F.mGrid[i][j];
  K->mvpMapPoints.assign(K->N, static_cast<MapPoint*>(NULL));
}

}  // namespace

#endif  // ORBX_WORLD_SCENE_H
