// Exercises the C++ adapters (include/ORBextractor.h, ORBVocabulary.h, and the drop-in ORBmatcher) exactly the way the
// reference's callers use those classes (src/Frame.cc:418-425 ExtractORB, :738-745 ComputeBoW), built WITHOUT OpenCV (the
// container shim of oracle/ref_shims; the Frame type is the one of tests/support/ref_world).  Writes raw results for
// tests/test_adapters.py to compare with the oracle.  The twelve matcher routines themselves are compared with the
// reference's own src/ORBmatcher.cc by tests/test_matcher_world.py.
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
      Frame::mnMinX = 0; Frame::mnMinY = 0; Frame::mnMaxX = (float)cols; Frame::mnMaxY = (float)rows;
      Frame::mfGridElementWidthInv = static_cast<float>(FRAME_GRID_COLS) / static_cast<float>(Frame::mnMaxX - Frame::mnMinX);
      Frame::mfGridElementHeightInv = static_cast<float>(FRAME_GRID_ROWS) / static_cast<float>(Frame::mnMaxY - Frame::mnMinY);
      auto work = [&](std::vector<int>& m12, int& nm, std::vector<double>& bowv) {
        Frame F1, F2;
        for (Frame* F : {&F1, &F2}) {
          F->N = n; F->mvKeys = keys; F->mvKeysUn = keys; F->mDescriptors = descriptors.clone();
          F->AssignFeaturesToGrid();
        }
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
    std::printf("OK n=%d mono=%d\n", n, mono);
    delete extractor;
    return 0;
  } catch (const std::exception& e) {
    std::printf("NO_DEVICE %s\n", e.what());
    return 3;
  }
}
