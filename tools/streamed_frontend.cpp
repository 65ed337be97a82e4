// BENCH INFRASTRUCTURE: the per-frame sequence Tracking runs on every image of a monocular stream (BASELINE config 3's stand-in,
// SURVEY.md §3.1), timed as ONE loop, through the classes Tracking itself uses:
//   Frame::Frame -> ExtractORB           (*mpORBextractorLeft)(im, cv::Mat(), mvKeys, mDescriptors, vLapping)   src/Frame.cc:311,418-425
//   Frame::ComputeBoW                    mpORBvocabulary->transform(vCurrentDesc, mBowVec, mFeatVec, 4)          src/Frame.cc:738-745
//   Tracking::TrackWithMotionModel       matcher.SearchByProjection(mCurrentFrame, mLastFrame, th, mono)         src/Tracking.cc:2889 (:2897 retry 2*th)
//   Tracking::SearchLocalPoints          matcher.SearchByProjection(mCurrentFrame, mvpLocalMapPoints, th, ...)   src/Tracking.cc:3416
// The SAME file is compiled twice over the object model of tests/support/ref_world/:
//   drop-in build     include/ORBextractor.h + ORBVocabulary.h + ORBmatcher.h / csrc/ref_adapter/ORBmatcher.cc, linked against liborbx.so
//   reference build   the reference's src/ORBextractor.cc and src/ORBmatcher.cc compiled where they lie (oracle/ref_fragments.mk ->
//                     oracle/_ref/ref_streamed_frontend; the five OpenCV primitives the extractor calls are the oracle's scalar
//                     restatements) + the reference's DBoW2 through oracle/_ref/libref_dbow2.so
// Both print one JSON line: ms per frame of every stage and a digest of everything the sequence produced (keypoints, descriptors,
// BoW vectors, both searches' match vectors), which must be equal between the two builds.
//   streamed_frontend <frames.raw> <rows> <cols> <nframes> <nfeatures> <voc.txt> <passes>
#include "ORBextractor.h"

#include "../tests/support/world_scene.h"

#ifdef ORBX_H
#include "ORBVocabulary.h"
#else
extern "C" {
void* ref_voc_load(const char* path);
void ref_voc_free(void* h);
int ref_voc_transform(void* h, const uint8_t* desc, int n, int levelsup, uint32_t* ids, double* vals, uint32_t* fv_node, uint32_t* fv_feat, int* n_fv);
}
#endif

namespace {

struct Fnv {
  uint64_t h = 1469598103934665603ull;
  void bytes(const void* p, size_t n) { const unsigned char* b = (const unsigned char*)p; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } }
  template <class T> void val(const T& v) { bytes(&v, sizeof(T)); }
};

double ms_since(std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }

struct Stage { double extract = 0, bow = 0, frame_host = 0, search_last = 0, frustum_host = 0, search_local = 0; long frames = 0, feats = 0, m_last = 0, m_local = 0, retries = 0; };

}  // namespace

int main(int argc, char** argv) {
  if (argc < 8) { std::fprintf(stderr, "usage: streamed_frontend <frames.raw> <rows> <cols> <nframes> <nfeatures> <voc.txt> <passes>\n"); return 2; }
  const int rows = std::atoi(argv[2]), cols = std::atoi(argv[3]), nfr = std::atoi(argv[4]), nfeatures = std::atoi(argv[5]), passes = std::atoi(argv[7]);
  std::vector<unsigned char> buf((size_t)rows * cols * nfr);
  { std::ifstream f(argv[1], std::ios::binary); f.read((char*)buf.data(), (std::streamsize)buf.size()); if (!f) { std::fprintf(stderr, "cannot read %s\n", argv[1]); return 2; } }
  try {
    ORBextractor ex(nfeatures, 1.2f, 8, 20, 7);
#ifdef ORBX_H
    ex.SetKeepHostPyramid(false);   // monocular: nobody reads mvImagePyramid
    ORBVocabulary voc;
    if (!voc.loadFromTextFile(argv[6])) { std::fprintf(stderr, "cannot load %s\n", argv[6]); return 2; }
#ifdef ORBX_STUB_BACKEND
    const char* build = "drop-in host logic over the oracle-backed C-ABI stub (CPU test build)";
#else
    const char* build = "drop-in (liborbx.so)";
#endif
#else
    void* voc = ref_voc_load(argv[6]);
    if (!voc) { std::fprintf(stderr, "cannot load %s\n", argv[6]); return 2; }
    const char* build = "reference-compiled (src/ORBextractor.cc, src/ORBmatcher.cc, DBoW2; OpenCV primitives restated)";
#endif
    World w;
    w.rows = rows; w.cols = cols; w.nlevels = ex.GetLevels();
    w.scale = ex.GetScaleFactors(); w.sigma2 = ex.GetScaleSigmaSquares(); w.inv_sigma2 = ex.GetInverseScaleSigmaSquares();
    w.scaleFactor = ex.GetScaleFactor(); w.logScaleFactor = std::log(w.scaleFactor);
    std::vector<int> lap = {0, 1000};
    Fnv digest;
    Stage st;
    for (int pass = 0; pass <= passes; pass++) {   // pass 0 = warm-up (graph capture, first-touch allocations), not timed
      const bool timed_pass = pass > 0;
      w.views.assign(nfr, View());
      Scene s(w, false);
      std::vector<Frame> frames(nfr);
      std::vector<std::vector<MapPoint> > pts(nfr);
      Fnv d;
      for (int t = 0; t < nfr; t++) {
        View& V = w.views[t];
        Frame& Cur = frames[t];
        cv::Mat im(rows, cols, CV_8UC1, buf.data() + (size_t)t * rows * cols);
        // ---- Frame::Frame: ExtractORB
        auto t0 = std::chrono::steady_clock::now();
        ex(im, cv::Mat(), V.kps, V.desc, lap);
        const double e_ms = ms_since(t0);
        V.n = (int)V.kps.size();
        // ---- Frame::ComputeBoW
        t0 = std::chrono::steady_clock::now();
        DBoW2::BowVector bow;
#ifdef ORBX_H
        {
          std::vector<cv::Mat> vCurrentDesc;   // Converter::toDescriptorVector (src/Converter.cc:29-38)
          vCurrentDesc.reserve(V.desc.rows);
          for (int j = 0; j < V.desc.rows; j++) vCurrentDesc.push_back(V.desc.row(j));
          voc.transform(vCurrentDesc, bow, V.fv, 4);
        }
#else
        {
          std::vector<uint8_t> rowsbuf((size_t)V.n * 32);
          for (int j = 0; j < V.n; j++) std::memcpy(&rowsbuf[(size_t)j * 32], V.desc.ptr((int)j), 32);
          std::vector<uint32_t> ids(V.n + 1), fn(V.n + 1), ff(V.n + 1);
          std::vector<double> vals(V.n + 1);
          int nfv = 0;
          const int nb = ref_voc_transform(voc, rowsbuf.data(), V.n, 4, ids.data(), vals.data(), fn.data(), ff.data(), &nfv);
          for (int k = 0; k < nb; k++) bow.insert(bow.end(), std::make_pair(ids[k], vals[k]));
          V.fv.clear();
          for (int k = 0; k < nfv; k++) V.fv[fn[k]].push_back(ff[k]);
        }
#endif
        const double b_ms = ms_since(t0);
        // ---- rest of Frame::Frame on the host (UndistortKeyPoints without distortion = copy, AssignFeaturesToGrid): the same code in both builds
        t0 = std::chrono::steady_clock::now();
        s.make_frame(Cur, t, false, s.pose(t));
        Cur.mDescriptors = V.desc;   // Frame::Frame hands mDescriptors itself to operator() (src/Frame.cc:311,418-425): the same buffer, not a copy
        const double f_ms = ms_since(t0);
        d.val(V.n); d.bytes(V.kps.data(), (size_t)V.n * sizeof(cv::KeyPoint));
        for (int j = 0; j < V.n; j++) d.bytes(V.desc.ptr((int)j), 32);
        for (auto& kv : bow) { d.val(kv.first); d.val(kv.second); }
        for (auto& kv : V.fv) { d.val(kv.first); d.bytes(kv.second.data(), kv.second.size() * sizeof(unsigned)); }
        // the map this frame sees: points triangulated from this frame's own keypoints (used when it is the last frame / an older keyframe)
        s.make_points(pts[t], t, s.pose(t), 1000000 * (t + 1), t);
        double sl_ms = 0, fr_ms = 0, sp_ms = 0;
        int nLast = 0, nLocal = 0;
        if (t >= 1) {
          Frame& Last = frames[t - 1];
          for (int i = 0; i < Last.N; i++) {   // what tracking the last frame left: most keypoints carry a point, a few are outliers
            Last.mvpMapPoints[i] = (H(i, 40 + t) % 6 != 0) ? &pts[t - 1][i] : static_cast<MapPoint*>(NULL);
            Last.mvbOutlier[i] = H(i, 41 + t) % 10 == 0;
          }
          // ---- TrackWithMotionModel (src/Tracking.cc:2859-2897)
          ORBmatcher matcher(0.9, true);
          const float th = 15.f;
          t0 = std::chrono::steady_clock::now();
          nLast = matcher.SearchByProjection(Cur, Last, th, true);
          if (nLast < 20) {
            std::fill(Cur.mvpMapPoints.begin(), Cur.mvpMapPoints.end(), static_cast<MapPoint*>(NULL));
            nLast = matcher.SearchByProjection(Cur, Last, 2 * th, true);
            if (timed_pass) st.retries++;
          }
          sl_ms = ms_since(t0);
          d.val(nLast);
          for (MapPoint* p : Cur.mvpMapPoints) { const long id = p ? (long)p->mnId : -1; d.val(id); }
          // ---- TrackLocalMap -> SearchLocalPoints (src/Tracking.cc:3332-3416): the local map = points of two older frames
          std::vector<MapPoint*> vpLocal;
          for (int back : {2, 5})
            if (t - back >= 0)
              for (MapPoint& m : pts[t - back]) vpLocal.push_back(&m);
          if (!vpLocal.empty()) {
            std::vector<MapPoint> local;   // set_track_fields works on a contiguous vector: a copy of the points, gate fields included
            local.reserve(vpLocal.size());
            for (MapPoint* p : vpLocal) local.push_back(*p);
            t0 = std::chrono::steady_clock::now();
            set_track_fields(s, Cur, local, false);   // Frame::isInFrustum of every local point
            fr_ms = ms_since(t0);
            std::vector<MapPoint*> vp;
            for (MapPoint& m : local) vp.push_back(&m);
            ORBmatcher m2(0.8);
            t0 = std::chrono::steady_clock::now();
            nLocal = m2.SearchByProjection(Cur, vp, 1, false, 50.0f);
            sp_ms = ms_since(t0);
            d.val(nLocal);
            // `local` dies with this scope: record the matches by id, then unbind
            for (MapPoint*& p : Cur.mvpMapPoints) {
              const long id = p ? (long)p->mnId : -1;
              d.val(id);
              if (p && p >= &local.front() && p <= &local.back()) p = NULL;
            }
          }
        }
        if (timed_pass && t >= 5) {   // steady state: a last frame and both local-map sources exist
          st.extract += e_ms; st.bow += b_ms; st.frame_host += f_ms; st.search_last += sl_ms; st.frustum_host += fr_ms; st.search_local += sp_ms;
          st.frames++; st.feats += V.n; st.m_last += nLast; st.m_local += nLocal;
        }
      }
      if (pass == 0) digest = d;
      else if (d.h != digest.h) { std::fprintf(stderr, "pass %d produced different results than pass 0\n", pass); return 4; }
    }
    const double n = (double)std::max(st.frames, 1L);
    const double device_path = (st.extract + st.bow + st.search_last + st.search_local) / n;
    std::printf("{\"build\": \"%s\", \"frames_timed\": %ld, \"ms_per_frame\": %.4f, \"extract_ms\": %.4f, \"bow_ms\": %.4f, \"search_last_ms\": %.4f, "
                "\"search_local_ms\": %.4f, \"host_frame_ms\": %.4f, \"host_frustum_ms\": %.4f, \"features_per_frame\": %.1f, "
                "\"matches_last_per_frame\": %.1f, \"matches_local_per_frame\": %.1f, \"wide_retries\": %ld, \"results_digest\": \"%016llx\"}\n",
                build, st.frames, device_path, st.extract / n, st.bow / n, st.search_last / n, st.search_local / n, st.frame_host / n, st.frustum_host / n,
                st.feats / n, st.m_last / n, st.m_local / n, st.retries, (unsigned long long)digest.h);
#ifndef ORBX_H
    ref_voc_free(voc);
#endif
  } catch (const std::exception& e) {
    std::fprintf(stderr, "streamed_frontend: %s\n", e.what());
    return 3;
  }
  return 0;
}
