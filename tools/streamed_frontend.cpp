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
//   streamed_frontend <frames.raw> <rows> <cols> <nframes> <nfeatures> <voc.txt> <passes> [--frames T] [--timestamps <stamps.txt>] [--pace 0|1|2]
// Long form (BASELINE config 3 at length: Examples/Monocular/mono_euroc.cc:84-160 feeds 3 682 images of MH_01 at the rate of their
// time stamps): `--frames T` streams T frames per pass by walking the <nframes> images forth and back (0 … n-1, n-2 … 1, 0 …: the
// camera of the synthetic stream pans one way, then the other, so consecutive frames always overlap), keeping only the last 8 frames and
// their points alive, as Tracking does (mCurrentFrame / mLastFrame + the local map); `--timestamps` + `--pace 1` sleep after every frame
// until the next stamp is due, exactly the wait of mono_euroc.cc:150-160 (T = stamp[ni+1] - stamp[ni]; if ttrack < T usleep(T - ttrack));
// every call's time is kept per frame and reported as p50 / p90 / p99 / max.
#include "ORBextractor.h"

#include <thread>

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

// every timed frame's time of one call, for the percentiles of the long form
struct Series {
  std::vector<float> v;
  void add(double x) { v.push_back((float)x); }
  std::string json() {
    if (v.empty()) return "null";
    std::vector<float> s(v);
    std::sort(s.begin(), s.end());
    auto q = [&](double p) { return s[std::min(s.size() - 1, (size_t)(p * (double)s.size()))]; };
    double sum = 0;
    for (float x : s) sum += x;
    char b[200];
    std::snprintf(b, sizeof b, "{\"mean\": %.4f, \"p50\": %.4f, \"p90\": %.4f, \"p99\": %.4f, \"max\": %.4f}", sum / (double)s.size(), q(0.50), q(0.90), q(0.99), s.back());
    return b;
  }
};

}  // namespace

int main(int argc, char** argv) {
  if (argc < 8) {
    std::fprintf(stderr, "usage: streamed_frontend <frames.raw> <rows> <cols> <nframes> <nfeatures> <voc.txt> <passes> [--frames T] [--timestamps <stamps.txt>] [--pace 0|1|2]\n");
    return 2;
  }
  const int rows = std::atoi(argv[2]), cols = std::atoi(argv[3]), nfr = std::atoi(argv[4]), nfeatures = std::atoi(argv[5]), passes = std::atoi(argv[7]);
  int long_frames = 0, pace = 0;
  std::vector<double> stamps;   // seconds
  for (int a = 8; a + 1 < argc; a += 2) {
    const std::string k = argv[a];
    if (k == "--frames") long_frames = std::atoi(argv[a + 1]);
    else if (k == "--pace") pace = std::atoi(argv[a + 1]);
    else if (k == "--timestamps") {   // Examples/Monocular/mono_euroc.cc:193-199 (LoadImages): one integer of nanoseconds per line, t = ns / 1e9
      std::ifstream f(argv[a + 1]);
      std::string line;
      while (std::getline(f, line))
        if (!line.empty()) stamps.push_back(std::strtod(line.c_str(), nullptr) / 1e9);
      if (stamps.empty()) { std::fprintf(stderr, "cannot read time stamps from %s\n", argv[a + 1]); return 2; }
    } else { std::fprintf(stderr, "unknown option %s\n", k.c_str()); return 2; }
  }
  if (pace && stamps.empty()) { std::fprintf(stderr, "--pace 1 needs --timestamps\n"); return 2; }
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
    // short form: every pass walks the nfr images once and keeps all of them; long form: T frames per timed pass over a ring of 8
    // (Tracking keeps mCurrentFrame, mLastFrame and the local map's points: here the points of frames t-2 and t-5)
    const bool longform = long_frames > 0;
    const int ring = longform ? 8 : nfr;
    const int warm_frames = longform ? std::min(long_frames, 32) : nfr;
    const int period = nfr > 1 ? 2 * (nfr - 1) : 1;
    const int first_timed = longform ? 8 : 5;   // steady state: a last frame and both local-map sources exist
    Fnv digest;
    Stage st;
    Series q_extract, q_bow, q_last, q_local, q_total, q_wall;
    double wall_s = 0, slept_s = 0;
    for (int pass = 0; pass <= passes; pass++) {   // pass 0 = warm-up (graph capture, first-touch allocations), not timed
      const bool timed_pass = pass > 0;
      const int T = longform ? (timed_pass ? long_frames : warm_frames) : nfr;
      w.views.assign(ring, View());
      Scene s(w, false);
      std::vector<Frame> frames(ring);
      std::vector<std::vector<MapPoint> > pts(ring);
      Fnv d;
      const auto pass_t0 = std::chrono::steady_clock::now();
      for (int t = 0; t < T; t++) {
        const auto frame_t0 = std::chrono::steady_clock::now();
        const int slot = t % ring;
        int v = t % period;           // image of frame t: forth and back through the set
        if (v >= nfr) v = period - v;
        View& V = w.views[slot];
        Frame& Cur = frames[slot];
        if (longform) { V = View(); Cur = Frame(); }
        cv::Mat im(rows, cols, CV_8UC1, buf.data() + (size_t)v * rows * cols);
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
        s.make_frame(Cur, slot, false, s.pose(v));
        Cur.mDescriptors = V.desc;   // Frame::Frame hands mDescriptors itself to operator() (src/Frame.cc:311,418-425): the same buffer, not a copy
        const double f_ms = ms_since(t0);
        d.val(V.n); d.bytes(V.kps.data(), (size_t)V.n * sizeof(cv::KeyPoint));
        for (int j = 0; j < V.n; j++) d.bytes(V.desc.ptr((int)j), 32);
        for (auto& kv : bow) { d.val(kv.first); d.val(kv.second); }
        for (auto& kv : V.fv) { d.val(kv.first); d.bytes(kv.second.data(), kv.second.size() * sizeof(unsigned)); }
        // the map this frame sees: points triangulated from this frame's own keypoints (used when it is the last frame / an older keyframe)
        s.make_points(pts[slot], slot, s.pose(v), 1000000 * (t % 2000 + 1), t);
        double sl_ms = 0, fr_ms = 0, sp_ms = 0;
        int nLast = 0, nLocal = 0;
        if (t >= 1) {
          Frame& Last = frames[(t - 1) % ring];
          std::vector<MapPoint>& lastPts = pts[(t - 1) % ring];
          for (int i = 0; i < Last.N; i++) {   // what tracking the last frame left: most keypoints carry a point, a few are outliers
            Last.mvpMapPoints[i] = (H(i, 40 + t) % 6 != 0) ? &lastPts[i] : static_cast<MapPoint*>(NULL);
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
              for (MapPoint& m : pts[(t - back) % ring]) vpLocal.push_back(&m);
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
        if (timed_pass && t >= first_timed) {
          st.extract += e_ms; st.bow += b_ms; st.frame_host += f_ms; st.search_last += sl_ms; st.frustum_host += fr_ms; st.search_local += sp_ms;
          st.frames++; st.feats += V.n; st.m_last += nLast; st.m_local += nLocal;
          if (longform) {
            q_extract.add(e_ms); q_bow.add(b_ms); q_last.add(sl_ms); q_local.add(sp_ms); q_total.add(e_ms + b_ms + sl_ms + sp_ms);
            q_wall.add(ms_since(frame_t0));
          }
        }
        if (pace && timed_pass && t + 1 < T) {   // mono_euroc.cc:150-160: wait for the next image
          const size_t ni = (size_t)t % (stamps.size() - 1);
          const double Tnext = stamps[ni + 1] - stamps[ni];
          const double ttrack = ms_since(frame_t0) * 1e-3;
          if (ttrack < Tnext) {
            if (pace == 2) {   // attribution only: the same 20 Hz with the host core kept busy (GPU idle between frames as before, no CPU sleep state)
              while (ms_since(frame_t0) * 1e-3 < Tnext) {}
            } else std::this_thread::sleep_for(std::chrono::duration<double>(Tnext - ttrack));
            slept_s += Tnext - ttrack;
          }
        }
      }
      if (timed_pass) wall_s += ms_since(pass_t0) * 1e-3;
      if (pass == 0 && !longform) digest = d;
      else if (pass == 1 && longform) digest = d;
      else if (pass > 0 && d.h != digest.h) { std::fprintf(stderr, "pass %d produced different results than pass %d\n", pass, longform ? 1 : 0); return 4; }
    }
    const double n = (double)std::max(st.frames, 1L);
    const double device_path = (st.extract + st.bow + st.search_last + st.search_local) / n;
    std::printf("{\"build\": \"%s\", \"frames_timed\": %ld, \"ms_per_frame\": %.4f, \"extract_ms\": %.4f, \"bow_ms\": %.4f, \"search_last_ms\": %.4f, "
                "\"search_local_ms\": %.4f, \"host_frame_ms\": %.4f, \"host_frustum_ms\": %.4f, \"features_per_frame\": %.1f, "
                "\"matches_last_per_frame\": %.1f, \"matches_local_per_frame\": %.1f, \"wide_retries\": %ld, \"results_digest\": \"%016llx\"",
                build, st.frames, device_path, st.extract / n, st.bow / n, st.search_last / n, st.search_local / n, st.frame_host / n, st.frustum_host / n,
                st.feats / n, st.m_last / n, st.m_local / n, st.retries, (unsigned long long)digest.h);
    if (longform)
      std::printf(", \"stream\": {\"frames_per_pass\": %d, \"images\": %d, \"order\": \"forth and back\", \"ring\": %d, \"paced\": %s, \"wait\": \"%s\", \"stamps\": %zu, \"wall_s\": %.2f, \"slept_s\": %.2f}, "
                  "\"percentiles\": {\"extract_ms\": %s, \"bow_ms\": %s, \"search_last_ms\": %s, \"search_local_ms\": %s, \"four_calls_ms\": %s, \"frame_wall_ms\": %s}",
                  long_frames, nfr, ring, pace ? "true" : "false", pace == 2 ? "busy-wait" : pace ? "sleep" : "none", stamps.size(), wall_s, slept_s, q_extract.json().c_str(), q_bow.json().c_str(),
                  q_last.json().c_str(), q_local.json().c_str(), q_total.json().c_str(), q_wall.json().c_str());
    std::printf("}\n");
#ifndef ORBX_H
    ref_voc_free(voc);
#endif
  } catch (const std::exception& e) {
    std::fprintf(stderr, "streamed_frontend: %s\n", e.what());
    return 3;
  }
  return 0;
}
