// TEST INFRASTRUCTURE: include/ORBextractor.h over an "OpenCV" that no known variant reproduces (the shim's cv::GaussianBlur with
// -DORBO_SHIM_UNKNOWN_BLUR: every 97th byte one off).  The constructor must REFUSE unless ORBX_ALLOW_UNPINNED=1.
// Links the oracle-backed stub of the C-ABI: no GPU involved, the refusal happens before any compute.
#include <cstdio>
#include <stdexcept>

#include "ORBextractor.h"

int main() {
#ifndef ORBX_CV_CALIBRATION
  std::printf("no-calibration\n");
  return 2;
#else
  try {
    ORB_SLAM3::ORBextractor ex(500, 1.2f, 4, 20, 7);
    std::printf("constructed pinned=%d\n", (int)orbx_cv::pinned(orbx_cv::opencv_calibration()));
    return 0;
  } catch (const std::exception& e) {
    std::printf("refused: %s\n", e.what());
    return 7;
  }
#endif
}
