// Exercises the header-only C++ adapters (include/ORBextractor.h, ORBmatcher.h, ORBVocabulary.h) exactly the way
// the reference's callers use those classes (src/Frame.cc:418-425 ExtractORB, :738-745 ComputeBoW), built WITHOUT
// OpenCV (include/orbx_cv_compat.h).  Writes raw results for tests/test_adapters.py to compare with the oracle.
//   adapter_demo probe
//   adapter_demo run <img.raw> <rows> <cols> <nfeatures> <lap0> <lap1> <out.bin> [voc.txt]
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <thread>
#include <vector>

#include "ORBVocabulary.h"
#include "ORBextractor.h"
#include "ORBmatcher.h"

using namespace ORB_SLAM3;

// the members of ORB_SLAM3::Frame that ORBmatcher::SearchForInitialization touches (include/Frame.h)
struct MiniFrame {
  std::vector<cv::KeyPoint> mvKeysUn;
  cv::Mat mDescriptors;
  static float mnMinX, mnMinY, mnMaxX, mnMaxY;
};
float MiniFrame::mnMinX = 0, MiniFrame::mnMinY = 0, MiniFrame::mnMaxX = 0, MiniFrame::mnMaxY = 0;

// ... and the members of Frame / MapPoint that ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&, ...) touches
struct V3 { float v[3]; float operator()(int k) const { return v[k]; } };
struct V2 { float v[2]; float operator()(int k) const { return v[k]; } };
struct MiniMapPoint {
  bool mbTrackInView = true, mbTrackInViewR = false, bad = false;
  float mTrackDepth = 1.f, mTrackProjX = 0, mTrackProjY = 0, mTrackProjXR = 0, mTrackViewCos = 1.f;
  int mnTrackScaleLevel = 0, nObs = 1;
  cv::Mat desc;
  bool isBad() const { return bad; }
  int Observations() const { return nObs; }
  cv::Mat GetDescriptor() const { return desc.clone(); }
  V3 world{{0, 0, 1}};
  V3 GetWorldPos() const { return world; }
};
// stand-ins for Eigen::Vector3f / Vector2f, Sophus::SE3f (identity rotation) and GeometricCamera (pinhole)
struct MiniSE3 {
  V3 t{{0, 0, 0}};
  MiniSE3 inverse() const { MiniSE3 r; r.t = V3{{-t.v[0], -t.v[1], -t.v[2]}}; return r; }
  V3 translation() const { return t; }
  V3 operator*(const V3& p) const { return V3{{p.v[0] + t.v[0], p.v[1] + t.v[1], p.v[2] + t.v[2]}}; }
};
struct MiniCamera {
  float fx = 500.f, cx = 320.f, cy = 240.f;
  V2 project(const V3& p) const { return V2{{fx * (p(0) / p(2)) + cx, fx * (p(1) / p(2)) + cy}}; }
};
struct MiniFrame2 : MiniFrame {
  int Nleft = -1, N = 0;
  float mb = 0.1f, mbf = 50.f;
  std::vector<float> mvuRight, mvScaleFactors;
  std::vector<MiniMapPoint*> mvpMapPoints;
  std::vector<bool> mvbOutlier;
  std::vector<cv::KeyPoint> mvKeys;
  MiniSE3 pose;
  MiniCamera cam, *mpCamera = &cam;
  MiniSE3 GetPose() const { return pose; }
};

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  const std::string mode = argv[1];
  try {
    if (mode == "probe") {
      ORBextractor ex(1000, 1.2f, 8, 20, 7);
      std::printf("DEVICE_OK levels=%d\n", ex.GetLevels());
      return 0;
    }
    if (mode == "stream" && argc >= 6) {
      // per-frame latency of ORBextractor::operator() the way Tracking sees it (steady_clock around the call, like
      // Examples/Monocular/mono_euroc.cc:133-143): adapter_demo stream <imgs.raw> <rows> <cols> <nframes> [keep_pyramid]
      const int rows = std::atoi(argv[3]), cols = std::atoi(argv[4]), nfr = std::atoi(argv[5]);
      const bool keep = argc >= 7 ? std::atoi(argv[6]) != 0 : false;
      std::vector<unsigned char> buf((size_t)rows * cols * nfr);
      { std::ifstream f(argv[2], std::ios::binary); f.read((char*)buf.data(), (std::streamsize)buf.size()); }
      ORBextractor ex(1000, 1.2f, 8, 20, 7);
      ex.SetKeepHostPyramid(keep);
      std::vector<int> lap = {0, 1000};
      std::vector<cv::KeyPoint> keys;
      cv::Mat desc;
      long total = 0;
      for (int pass = 0; pass < 2; pass++) {   // pass 0 = warm-up
        total = 0;
        const auto t0 = std::chrono::steady_clock::now();
        for (int rep = 0; rep < 20; rep++)
          for (int f = 0; f < nfr; f++) {
            cv::Mat im(rows, cols, CV_8UC1, buf.data() + (size_t)f * rows * cols);
            ex(im, cv::Mat(), keys, desc, lap);
            total += (long)keys.size();
          }
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (pass == 1) std::printf("STREAM frames=%d ms_per_frame=%.4f features_per_ms=%.1f keep_pyramid=%d\n", 20 * nfr, ms / (20 * nfr), total / ms, (int)keep);
      }
      return 0;
    }
    if (mode == "mt" && argc >= 6) {
      // Tracking / LocalMapping / LoopClosing use matchers and the vocabulary concurrently: three threads, each with
      // stack-constructed matchers, one shared vocabulary; every thread must reproduce the single-threaded results.
      //   adapter_demo mt <img.raw> <rows> <cols> <voc.txt>
      const int rows = std::atoi(argv[3]), cols = std::atoi(argv[4]);
      std::vector<unsigned char> buf((size_t)rows * cols);
      { std::ifstream f(argv[2], std::ios::binary); f.read((char*)buf.data(), (std::streamsize)buf.size()); }
      cv::Mat im(rows, cols, CV_8UC1, buf.data());
      ORBextractor ex(1000, 1.2f, 8, 20, 7);
      std::vector<cv::KeyPoint> keys;
      cv::Mat descriptors;
      std::vector<int> lap = {0, 0};
      ex(im, cv::Mat(), keys, descriptors, lap);
      const int n = (int)keys.size();
      ORBVocabulary voc;
      if (!voc.loadFromTextFile(argv[5])) return 4;
      MiniFrame::mnMaxX = (float)cols; MiniFrame::mnMaxY = (float)rows;
      auto work = [&](std::vector<int>& m12, int& nm, std::vector<double>& bowv) {
        MiniFrame F1, F2;
        F1.mvKeysUn = keys; F1.mDescriptors = descriptors.clone();
        F2.mvKeysUn = keys; F2.mDescriptors = descriptors.clone();
        std::vector<cv::Point2f> prev(n);
        for (int i = 0; i < n; i++) prev[i] = keys[i].pt;
        ORBmatcher matcher(0.9f, true);
        nm = matcher.SearchForInitialization(F1, F2, prev, m12, 100);
        std::vector<cv::Mat> vdesc;
        for (int i = 0; i < n; i++) vdesc.push_back(descriptors.row(i));
        DBoW2::BowVector bow; DBoW2::FeatureVector fv;
        voc.transform(vdesc, bow, fv, 2);
        bowv.clear();
        for (auto& kv : bow) { bowv.push_back((double)kv.first); bowv.push_back(kv.second); }
      };
      std::vector<int> ref_m12; int ref_nm = 0; std::vector<double> ref_bow;
      work(ref_m12, ref_nm, ref_bow);
      int bad = 0;
      std::vector<std::thread> ths;
      std::vector<int> bads(3, 0);
      for (int t = 0; t < 3; t++)
        ths.emplace_back([&, t] {
          try {
            for (int rep = 0; rep < 25; rep++) {
              std::vector<int> m12; int nm = 0; std::vector<double> bowv;
              work(m12, nm, bowv);
              if (nm != ref_nm || m12 != ref_m12 || bowv != ref_bow) bads[t]++;
            }
          } catch (const std::exception& e) { std::fprintf(stderr, "thread %d: %s\n", t, e.what()); bads[t] += 1000; }
        });
      for (auto& th : ths) th.join();
      for (int b : bads) bad += b;
      std::printf(bad ? "MT_FAIL %d\n" : "MT_OK n=%d matches=%d\n", bad ? bad : n, ref_nm);
      return bad ? 5 : 0;
    }
    if (mode != "run" || argc < 9) return 2;
    const int rows = std::atoi(argv[3]), cols = std::atoi(argv[4]), nf = std::atoi(argv[5]);
    std::vector<int> lap = {std::atoi(argv[6]), std::atoi(argv[7])};
    std::vector<unsigned char> buf((size_t)rows * cols);
    { std::ifstream f(argv[2], std::ios::binary); f.read((char*)buf.data(), (std::streamsize)buf.size()); }
    cv::Mat im(rows, cols, CV_8UC1, buf.data());
    ORBextractor* extractor = new ORBextractor(nf, 1.2f, 8, 20, 7);   // src/Tracking.cc:597-603
    std::vector<cv::KeyPoint> keys;
    cv::Mat descriptors;
    const int mono = (*extractor)(im, cv::Mat(), keys, descriptors, lap);  // src/Frame.cc:421-424
    const int n = (int)keys.size();
    std::ofstream o(argv[8], std::ios::binary);
    o.write((const char*)&n, 4); o.write((const char*)&mono, 4);
    o.write((const char*)keys.data(), (std::streamsize)n * sizeof(cv::KeyPoint));
    for (int i = 0; i < n; i++) o.write((const char*)descriptors.ptr<unsigned char>(i), 32);
    // mvImagePyramid side channel (src/Frame.cc:818,908-925)
    int np = (int)extractor->mvImagePyramid.size();
    o.write((const char*)&np, 4);
    for (int l = 0; l < np; l++) {
      const cv::Mat& L = extractor->mvImagePyramid[l];
      o.write((const char*)&L.rows, 4); o.write((const char*)&L.cols, 4);
      for (int r = 0; r < L.rows; r++) o.write((const char*)L.ptr<unsigned char>(r), L.cols);
    }
    // matcher statics
    int d01 = n >= 2 ? ORBmatcher::DescriptorDistance(descriptors.row(0), descriptors.row(1)) : -1;
    o.write((const char*)&d01, 4);
    // BoW (src/Frame.cc:738-745): Converter::toDescriptorVector then transform(..., 4)
    int nb = 0;
    if (argc >= 10) {
      ORBVocabulary voc;
      if (!voc.loadFromTextFile(argv[9])) { std::fprintf(stderr, "vocabulary load failed\n"); return 4; }
      std::vector<cv::Mat> vdesc;
      for (int i = 0; i < n; i++) vdesc.push_back(descriptors.row(i));
      DBoW2::BowVector bow;
      DBoW2::FeatureVector fv;
      voc.transform(vdesc, bow, fv, 2);
      nb = (int)bow.size();
      o.write((const char*)&nb, 4);
      for (auto& kv : bow) { o.write((const char*)&kv.first, 4); o.write((const char*)&kv.second, 8); }
      int nfv = 0;
      for (auto& kv : fv) nfv += (int)kv.second.size();
      o.write((const char*)&nfv, 4);
      for (auto& kv : fv) for (unsigned f : kv.second) { o.write((const char*)&kv.first, 4); o.write((const char*)&f, 4); }
      const double self = voc.score(bow, bow);
      o.write((const char*)&self, 8);
    } else {
      o.write((const char*)&nb, 4);
    }
    // SearchForInitialization of the frame against itself shifted by one row of the descriptor order (src/Tracking.cc:2494-2495)
    {
      MiniFrame::mnMaxX = (float)cols; MiniFrame::mnMaxY = (float)rows;
      MiniFrame F1, F2;
      F1.mvKeysUn = keys; F1.mDescriptors = descriptors.clone();
      F2.mvKeysUn = keys; F2.mDescriptors = descriptors.clone();
      std::vector<cv::Point2f> prev(n);
      for (int i = 0; i < n; i++) prev[i] = keys[i].pt;
      std::vector<int> m12;
      ORBmatcher matcher(0.9f, true);
      const int nm = n ? matcher.SearchForInitialization(F1, F2, prev, m12, 100) : 0;
      o.write((const char*)&nm, 4);
      o.write((const char*)m12.data(), (std::streamsize)m12.size() * 4);
    }
    // SearchByProjection: the frame's own keypoints as map points, projected 1.5 / 0.5 px away; every 3rd one unobserved,
    // every 5th not in view, every 7th bad, every 11th keypoint already bound to an observed point
    {
      MiniFrame2 F;
      F.mvKeysUn = keys; F.mDescriptors = descriptors.clone();
      F.mvScaleFactors = extractor->GetScaleFactors();
      F.mvpMapPoints.assign(n, nullptr);
      std::vector<MiniMapPoint> mps(n), bound(n);
      std::vector<MiniMapPoint*> vp(n);
      for (int i = 0; i < n; i++) {
        MiniMapPoint& m = mps[i];
        m.mTrackProjX = keys[i].pt.x + 1.5f; m.mTrackProjY = keys[i].pt.y + 0.5f;
        m.mTrackViewCos = (i & 1) ? 0.9f : 0.999f;
        m.mnTrackScaleLevel = keys[i].octave;
        m.nObs = (i % 3 == 0) ? 0 : 2;
        m.mbTrackInView = i % 5 != 0;
        m.bad = i % 7 == 0;
        m.mTrackDepth = (i % 13 == 0) ? 100.f : 1.f;
        m.desc = descriptors.row(i).clone();
        vp[i] = &m;
        if (i % 11 == 0) { bound[i].nObs = 4; F.mvpMapPoints[i] = &bound[i]; }
      }
      ORBmatcher matcher(0.8f, true);
      const int nm = n ? matcher.SearchByProjection(F, vp, 3.0f, true, 50.0f) : 0;
      o.write((const char*)&nm, 4);
      for (int i = 0; i < n; i++) {
        int who = -1;
        if (F.mvpMapPoints[i] && F.mvpMapPoints[i] >= &mps[0] && F.mvpMapPoints[i] <= &mps[n - 1]) who = (int)(F.mvpMapPoints[i] - &mps[0]);
        o.write((const char*)&who, 4);
      }
    }
    // SearchByProjection(CurrentFrame, LastFrame): the frame against itself, its keypoints back-projected at depths 2..8 m
    // and seen from a camera moved by (0.01, 0.005, 0.3) -> forward motion for the stereo case (tlc.z > mb)
    {
      MiniFrame2 Last, Cur;
      Last.mvKeysUn = keys; Last.mvKeys = keys; Last.N = n; Last.mDescriptors = descriptors.clone();
      Cur.mvKeysUn = keys; Cur.mvKeys = keys; Cur.N = n; Cur.mDescriptors = descriptors.clone();
      Cur.mvScaleFactors = extractor->GetScaleFactors(); Last.mvScaleFactors = Cur.mvScaleFactors;
      Cur.pose.t = V3{{-0.01f, -0.005f, -0.3f}};
      Cur.mvpMapPoints.assign(n, nullptr);
      Last.mvpMapPoints.assign(n, nullptr); Last.mvbOutlier.assign(n, false);
      Cur.mvuRight.resize(n);
      std::vector<MiniMapPoint> mps(n);
      for (int i = 0; i < n; i++) {
        const float z = 2.0f + (float)(i % 7);
        mps[i].world = V3{{((keys[i].pt.x - 320.f) * z) / 500.f, ((keys[i].pt.y - 240.f) * z) / 500.f, z}};
        mps[i].nObs = (i % 3 == 0) ? 0 : 2;
        mps[i].desc = descriptors.row(i).clone();
        if (i % 6 != 0) Last.mvpMapPoints[i] = &mps[i];
        Last.mvbOutlier[i] = i % 10 == 0;
        Cur.mvuRight[i] = (i % 4 == 0) ? -1.f : keys[i].pt.x - 50.f / (2.0f + (float)(i % 7));
      }
      ORBmatcher matcher(0.9f, true);
      const int nm = n ? matcher.SearchByProjection(Cur, Last, 15.0f, false) : 0;
      o.write((const char*)&nm, 4);
      for (int i = 0; i < n; i++) {
        int who = Cur.mvpMapPoints[i] ? (int)(Cur.mvpMapPoints[i] - &mps[0]) : -1;
        o.write((const char*)&who, 4);
      }
    }
    // SearchByBoW(KeyFrame*, Frame&, matches): the frame against itself as a keyframe, feature vectors from the vocabulary
    // (levelsup 2); every 4th keyframe feature has no map point, every 9th a bad one
    if (argc >= 10 && n) {
      ORBVocabulary voc;
      voc.loadFromTextFile(argv[9]);
      std::vector<cv::Mat> vdesc;
      for (int i = 0; i < n; i++) vdesc.push_back(descriptors.row(i));
      DBoW2::BowVector bow;
      struct MiniKF {
        std::vector<MiniMapPoint*> mps;
        DBoW2::FeatureVector mFeatVec;
        cv::Mat mDescriptors;
        std::vector<cv::KeyPoint> mvKeysUn;
        void* mpCamera2 = nullptr;
        std::vector<MiniMapPoint*> GetMapPointMatches() const { return mps; }
      } kf;
      struct MiniF { int N = 0, Nleft = -1; DBoW2::FeatureVector mFeatVec; cv::Mat mDescriptors; std::vector<cv::KeyPoint> mvKeys; } Fb;
      voc.transform(vdesc, bow, kf.mFeatVec, 2);
      Fb.mFeatVec = kf.mFeatVec;
      kf.mDescriptors = descriptors.clone(); kf.mvKeysUn = keys;
      Fb.mDescriptors = descriptors.clone(); Fb.mvKeys = keys; Fb.N = n;
      std::vector<MiniMapPoint> mps(n);
      kf.mps.assign(n, nullptr);
      for (int i = 0; i < n; i++) { mps[i].bad = i % 9 == 0; if (i % 4 != 0) kf.mps[i] = &mps[i]; }
      std::vector<MiniMapPoint*> matches;
      ORBmatcher matcher(0.7f, true);
      const int nm = matcher.SearchByBoW(&kf, Fb, matches);
      o.write((const char*)&nm, 4);
      for (int i = 0; i < n; i++) { int who = matches[i] ? (int)(matches[i] - &mps[0]) : -1; o.write((const char*)&who, 4); }
    } else {
      int nm = -1; o.write((const char*)&nm, 4);
    }
    // SearchForTriangulation(pKF1, pKF2, pairs, bOnlyStereo, bCoarse): the frame as two keyframes; a stand-in camera whose
    // epipolarConstrain accepts rows closer than 3 px (the real ones are the reference's GeometricCamera classes)
    if (argc >= 10 && n) {
      ORBVocabulary voc;
      voc.loadFromTextFile(argv[9]);
      std::vector<cv::Mat> vdesc;
      for (int i = 0; i < n; i++) vdesc.push_back(descriptors.row(i));
      struct TriCam {
        V2 project(const V3& p) const { return V2{{500.f * (p(0) / p(2)) + 320.f, 500.f * (p(1) / p(2)) + 240.f}}; }
        bool epipolarConstrain(TriCam*, const cv::KeyPoint& kp1, const cv::KeyPoint& kp2, const int&, const V3&, float, float) const {
          return std::fabs(kp1.pt.y - kp2.pt.y) < 3.0f;
        }
      };
      struct TriSE3 : MiniSE3 {
        int rotationMatrix() const { return 0; }
        TriSE3 operator*(const TriSE3& o) const { TriSE3 r; r.t = V3{{t.v[0] + o.t.v[0], t.v[1] + o.t.v[1], t.v[2] + o.t.v[2]}}; return r; }
        V3 operator*(const V3& p) const { return MiniSE3::operator*(p); }
      };
      struct TriKF {
        int N = 0;
        void* mpCamera2 = nullptr;
        TriCam cam, *mpCamera = &cam;
        TriSE3 pose;
        DBoW2::FeatureVector mFeatVec;
        cv::Mat mDescriptors;
        std::vector<cv::KeyPoint> mvKeysUn;
        std::vector<float> mvuRight, mvScaleFactors, mvLevelSigma2;
        std::vector<MiniMapPoint*> mps;
        TriSE3 GetPose() const { return pose; }
        TriSE3 GetPoseInverse() const { TriSE3 r; r.t = V3{{-pose.t.v[0], -pose.t.v[1], -pose.t.v[2]}}; return r; }
        V3 GetCameraCenter() const { return V3{{-pose.t.v[0], -pose.t.v[1], -pose.t.v[2]}}; }
        MiniMapPoint* GetMapPoint(size_t i) const { return mps[i]; }
      } k1, k2;
      DBoW2::BowVector bow;
      voc.transform(vdesc, bow, k1.mFeatVec, 2);
      k2.mFeatVec = k1.mFeatVec;
      MiniMapPoint some;
      for (TriKF* k : {&k1, &k2}) {
        k->N = n; k->mDescriptors = descriptors.clone(); k->mvKeysUn = keys;
        k->mvScaleFactors = extractor->GetScaleFactors(); k->mvLevelSigma2 = extractor->GetScaleSigmaSquares();
        k->mvuRight.assign(n, -1.f); k->mps.assign(n, nullptr);
      }
      k2.pose.t = V3{{0.2f, 0.f, 2.0f}};   // epipole of camera 1 in image 2: (0.2/2*500+320, 240) = (370, 240)
      for (int i = 0; i < n; i++) {
        if (i % 5 == 0) k1.mps[i] = &some;      // already has a map point
        if (i % 7 == 0) k2.mps[i] = &some;
        if (i % 3 == 0) k1.mvuRight[i] = 10.f;  // stereo observation
        if (i % 4 == 0) k2.mvuRight[i] = 10.f;
        k2.mvKeysUn[i].pt.y += (float)(i % 6) - 2.0f;   // so that the epipolar stand-in rejects some pairs
      }
      for (int pass = 0; pass < 2; pass++) {
        std::vector<std::pair<size_t, size_t> > pairs;
        ORBmatcher matcher(0.6f, pass == 0);
        const int nm = matcher.SearchForTriangulation(&k1, &k2, pairs, pass == 1, false);
        o.write((const char*)&nm, 4);
        const int np = (int)pairs.size();
        o.write((const char*)&np, 4);
        for (auto& pr : pairs) { int a = (int)pr.first, b = (int)pr.second; o.write((const char*)&a, 4); o.write((const char*)&b, 4); }
      }
    } else {
      int nm = -1; o.write((const char*)&nm, 4);
    }
    std::printf("OK n=%d mono=%d\n", n, mono);
    delete extractor;
    return 0;
  } catch (const std::exception& e) {
    std::printf("NO_DEVICE %s\n", e.what());
    return 3;
  }
}
