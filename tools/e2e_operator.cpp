// End-to-end latency of ORB_SLAM3::ORBextractor::operator() through the C++ adapter (include/ORBextractor.h), host
// buffers in and out — what Tracking sees per frame (steady_clock around the call, like
// Examples/Monocular/mono_euroc.cc:133-143): H2D image, the kernels, D2H keypoints + descriptors.  Built and run by bench.py.
//   e2e_operator <frames.raw> <rows> <cols> <nframes> <nfeatures> <reps>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <vector>

#include "ORBextractor.h"

int main(int argc, char** argv) {
  if (argc < 7) return 2;
  const int rows = std::atoi(argv[2]), cols = std::atoi(argv[3]), nfr = std::atoi(argv[4]), nfeat = std::atoi(argv[5]), reps = std::atoi(argv[6]);
  std::vector<unsigned char> buf((size_t)rows * cols * nfr);
  { std::ifstream f(argv[1], std::ios::binary); f.read((char*)buf.data(), (std::streamsize)buf.size()); if (!f) return 2; }
  try {
    ORB_SLAM3::ORBextractor ex(nfeat, 1.2f, 8, 20, 7);
    std::vector<int> lap = {0, 1000};
    std::vector<cv::KeyPoint> keys;
    cv::Mat desc;
    long total = 0;
    double ms = 0, ms_mono = 0;
    // mode 0: as constructed — the host mirror of mvImagePyramid refreshed on every call (stereo reads it, src/Frame.cc:818,908-925);
    // mode 1: SetKeepHostPyramid(false), what a monocular configuration would set
    for (int mode = 0; mode < 2; mode++) {
      ex.SetKeepHostPyramid(mode == 0);
      for (int pass = 0; pass < 2; pass++) {   // pass 0 = warm-up
        total = 0;
        const auto t0 = std::chrono::steady_clock::now();
        for (int rep = 0; rep < (pass ? reps : 2); rep++)
          for (int f = 0; f < nfr; f++) {
            cv::Mat im(rows, cols, CV_8UC1, buf.data() + (size_t)f * rows * cols);
            ex(im, cv::Mat(), keys, desc, lap);
            total += (long)keys.size();
          }
        (mode ? ms_mono : ms) = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      }
    }
    // the device share of a call: events around the replayed graph (a separate, short pass: the two event records cost the host a little)
    double dev_us[2] = {0, 0};
    orbx_set_option(ex.Context(), "graph_timing", 1);
    for (int mode = 0; mode < 2; mode++) {
      ex.SetKeepHostPyramid(mode == 0);
      double acc = 0;
      int n = 0;
      for (int rep = 0; rep < 3; rep++)
        for (int f = 0; f < nfr; f++) {
          cv::Mat im(rows, cols, CV_8UC1, buf.data() + (size_t)f * rows * cols);
          ex(im, cv::Mat(), keys, desc, lap);
          const double us = orbx_last_graph_device_us(ex.Context());
          if (rep && us > 0) { acc += us; n++; }
        }
      dev_us[mode] = n ? acc / n : -1;
    }
    std::printf("{\"frames\": %d, \"ms_per_frame\": %.5f, \"features_per_ms\": %.2f, \"features_per_frame\": %.1f, \"ms_per_frame_without_host_pyramid\": %.5f, "
                "\"device_ms_per_frame\": %.5f, \"device_ms_per_frame_without_host_pyramid\": %.5f}\n",
                reps * nfr, ms / (reps * nfr), total / ms, (double)total / (reps * nfr), ms_mono / (reps * nfr), dev_us[0] * 1e-3, dev_us[1] * 1e-3);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "e2e_operator: %s\n", e.what());
    return 3;
  }
  return 0;
}
