// TEST INFRASTRUCTURE: include/orbx_cv_calibrate.h against an "OpenCV" whose variant is known — the shim of oracle/ref_shims forwards
// cv::GaussianBlur / cv::fastAtan2 to the oracle, whose switches (orbo_set_gauss_variant / _tail / orbo_set_atan_fma) this driver sets.
// Prints, per requested variant, what the calibration detects.  tests/test_opencv_variants.py compares the two.
#include <cstdio>
#include <cstdlib>

#include "orbx_cv_calibrate.h"

extern "C" {
int orbo_set_gauss_variant(int kernel, int round);
int orbo_set_gauss_tail(int v);
int orbo_set_atan_fma(int on);
int orbx_set_option(orbx_ctx*, const char*, int) { return 0; }   // apply() is not under test here
}

int main(int argc, char** argv) {
#ifndef ORBX_CV_CALIBRATION
  std::printf("no-opencv\n");
  return 2;
#else
  // arguments: groups of four integers  kernel round tail atan_fma
  for (int i = 1; i + 3 < argc; i += 4) {
    const int k = std::atoi(argv[i]), r = std::atoi(argv[i + 1]), t = std::atoi(argv[i + 2]), f = std::atoi(argv[i + 3]);
    if (orbo_set_gauss_variant(k, r) != 0 || orbo_set_gauss_tail(t) != 0 || orbo_set_atan_fma(f) != 0) { std::printf("bad-variant\n"); return 3; }
    const orbx_cv::Calibration c = orbx_cv::calibrate(
        [](const uint8_t* src, int w, int h, uint8_t* dst) {
          cv::Mat s(h, w, CV_8UC1, (void*)src, (size_t)w);
          cv::Mat work = s.clone();
          cv::GaussianBlur(work, work, cv::Size(7, 7), 2, 2, cv::BORDER_REFLECT_101);
          for (int y = 0; y < h; y++) std::memcpy(dst + (size_t)y * w, work.ptr<unsigned char>(y), (size_t)w);
        },
        [](float y, float x) { return cv::fastAtan2(y, x); });
    std::printf("%d %d %d %d -> %d %d %d %d exact %d %d candidates %d contracts %d form %d frame %dx%d mismatch %d\n", k, r, t, f, c.gauss_kernel, c.gauss_round,
                c.gauss_tail, c.atan_fma, (int)c.gauss_exact, (int)c.atan_exact, c.gauss_candidates, c.brief_fma, c.brief_form, c.frame_w, c.frame_h, c.frame_mismatch);
  }
  return 0;
#endif
}
