/* orbx adapter — drop-in for the reference's include/ORBextractor.h (lturing/ORB_SLAM3_modified,
 * include/ORBextractor.h:44-112): same class name, namespace, constructor, operator(), getters and the public
 * mvImagePyramid member, so src/Frame.cc / src/Tracking.cc / src/CloudPoint.cc compile unchanged.  Header-only;
 * every computation is a call into the C ABI of liborbx.so (include/orbx.h, HIP kernels for gfx950).
 *
 * Differences a caller cannot observe through this interface:
 *   - the pyramid lives in persistent device buffers (the reference reallocates it per call);
 *     mvImagePyramid[l] is a host copy made after each call (needed only by the stereo matcher,
 *     src/Frame.cc:818,908-925) — switch it off for mono with SetKeepHostPyramid(false); a build with -DORBX_DEVICE_STEREO (and
 *     integration/Frame_stereo.patch applied to src/Frame.cc: four lines) never makes it: ComputeStereoMatches then runs on the two
 *     extractors' device pyramids (DeviceStereoMatches below);
 *   - mvImagePyramid[0] is a header over the caller's image and mvImagePyramid[l >= 1] are headers over a pinned host
 *     mirror that one asynchronous copy inside the call refreshes; there is no EDGE_THRESHOLD padding around them
 *     (nothing reads the pad), and like the reference's they are valid until the next call.
 * Which OpenCV: the constructor calibrates the context against the OpenCV this translation unit is built with (the 8-bit GaussianBlur
 * and fastAtan2 differ between releases / builds, and -march=native changes the reference's own pattern rotation): see
 * include/orbx_cv_calibrate.h and INTEGRATION.md section 6.
 * Errors: an empty image returns -1 like the reference (src/ORBextractor.cc:1090-1091); a missing GPU or a HIP
 * failure throws std::runtime_error from the constructor / operator() (the reference has no failure path at all;
 * there is deliberately no CPU fallback).
 */
#ifndef ORBEXTRACTOR_H
#define ORBEXTRACTOR_H

#include <cassert>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "orbx.h"
#include "orbx_cv_compat.h"
#include "orbx_cv_calibrate.h"

namespace ORB_SLAM3 {

class ORBextractor {
 public:
  enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

  ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int device_id = -1)
      : nfeatures(nfeatures), scaleFactor(scaleFactor), nlevels(nlevels), iniThFAST(iniThFAST), minThFAST(minThFAST) {
    const int rc = orbx_create(&ctx_, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device_id);
    if (rc != ORBX_OK) throw std::runtime_error("ORBextractor: orbx_create failed with code " + std::to_string(rc) +
                                                (rc == ORBX_E_DEVICE ? " (no MI355X / HIP device; there is no CPU fallback)" : ""));
    mvScaleFactor.resize(nlevels); mvInvScaleFactor.resize(nlevels); mvLevelSigma2.resize(nlevels);
    mvInvLevelSigma2.resize(nlevels); mnFeaturesPerLevel.resize(nlevels);
    orbx_scale_tables(ctx_, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(), mvInvLevelSigma2.data(),
                      mnFeaturesPerLevel.data());
#ifdef ORBX_DEVICE_STEREO
    keep_host_pyramid_ = false;   // the only reader of mvImagePyramid (Frame::ComputeStereoMatches) is patched to DeviceStereoMatches
#endif
    orbx_set_host_pyramid(ctx_, keep_host_pyramid_ ? 1 : 0);
#ifdef ORBX_CV_CALIBRATION
    // which OpenCV release / build (and which compiler flags) is the CPU path this object replaces: found out once per process by running
    // the real cv::GaussianBlur / cv::fastAtan2 on a probe (include/orbx_cv_calibrate.h); the context then computes exactly those bytes
    const orbx_cv::Calibration& cal = orbx_cv::opencv_calibration();
    if (!orbx_cv::pinned(cal) && !orbx_cv::allow_unpinned()) {
      orbx_destroy(ctx_);
      ctx_ = nullptr;
      throw std::runtime_error("ORBextractor: the OpenCV / toolchain this program is built with is not one liborbx can reproduce bit for bit: " +
                               orbx_cv::why_unpinned(cal) + "run tools/opencv_pin/run.sh (INTEGRATION.md section 6); ORBX_ALLOW_UNPINNED=1 accepts the difference");
    }
    if (orbx_cv::apply(ctx_, cal) != ORBX_OK)
      throw std::runtime_error(std::string("ORBextractor: ") + orbx_last_error(ctx_));
#endif
    cap_ = orbx_keypoint_capacity(ctx_);
    kps_.resize(cap_);
    desc_.resize((size_t)cap_ * 32);
    mvImagePyramid.resize(nlevels);
  }
  ~ORBextractor() { orbx_destroy(ctx_); }
  ORBextractor(const ORBextractor&) = delete;
  ORBextractor& operator=(const ORBextractor&) = delete;

  // Compute the ORB features and descriptors on an image; mask is ignored (as in the reference).
  // Returns monoIndex, or -1 for an empty image (src/ORBextractor.cc:1086-1168).
  int operator()(cv::InputArray _image, cv::InputArray /*_mask*/, std::vector<cv::KeyPoint>& _keypoints,
                 cv::OutputArray _descriptors, std::vector<int>& vLappingArea) {
    if (_image.empty()) return -1;
    cv::Mat image = _image.getMat();
    assert(image.type() == CV_8UC1);
    int n = 0, mono = 0;
    static_assert(sizeof(cv::KeyPoint) == sizeof(orbx_keypoint), "cv::KeyPoint must be the 28-byte POD orbx_keypoint mirrors");
    const int rc = orbx_extract(ctx_, image.data, image.rows, image.cols, (size_t)image.step, vLappingArea[0], vLappingArea[1],
                                kps_.data(), desc_.data(), &n, &mono);
    if (rc == ORBX_E_EMPTY) return -1;
    if (rc != ORBX_OK) throw std::runtime_error(std::string("ORBextractor: ") + orbx_last_error(ctx_));
    if (n == 0) _descriptors.release();
    else {
      _descriptors.create(n, 32, CV_8U);
      cv::Mat d = _descriptors.getMat();
      for (int i = 0; i < n; i++) std::memcpy(d.ptr<unsigned char>(i), desc_.data() + (size_t)i * 32, 32);
      // this buffer (Frame::mDescriptors) holds the rows of the context's last extraction: ORBVocabulary::transform on it (Frame::ComputeBoW) finds
      // the records the extraction graph already computed
      if (d.isContinuous()) orbx_publish_descriptors(ctx_, d.ptr<unsigned char>(0), n);
    }
    _keypoints.resize(n);
    if (n) std::memcpy((void*)_keypoints.data(), kps_.data(), (size_t)n * sizeof(orbx_keypoint));
    if (keep_host_pyramid_) {
      // level 0 is the caller's image (a header over it, like the reference's ROI into its padded copy); levels >= 1 are
      // headers over the context's pinned host mirror, filled by one asynchronous copy inside orbx_extract
      mvImagePyramid[0] = image;
      for (int l = 1; l < nlevels; l++) {
        const uint8_t* p = nullptr;
        size_t stride = 0;
        int w = 0, h = 0;
        if (orbx_host_pyramid_level(ctx_, l, &p, &stride, &w, &h) != ORBX_OK)
          throw std::runtime_error(std::string("ORBextractor: ") + orbx_last_error(ctx_));
        mvImagePyramid[l] = cv::Mat(h, w, CV_8UC1, (void*)p, stride);
      }
    }
    return mono;
  }

  int inline GetLevels() { return nlevels; }
  float inline GetScaleFactor() { return (float)scaleFactor; }
  std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
  std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
  std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
  std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

  std::vector<cv::Mat> mvImagePyramid;

  // orbx extensions (not in the reference)
  void SetKeepHostPyramid(bool on) { keep_host_pyramid_ = on; orbx_set_host_pyramid(ctx_, on ? 1 : 0); }
  orbx_ctx* Context() { return ctx_; }
  // Frame::ComputeStereoMatches (src/Frame.cc:811-981) as one call on the device pyramids the two extractors hold from their last
  // operator(): what integration/Frame_stereo.patch puts at the top of that routine (arguments = the Frame's own members, so the routine's
  // read of `mb` before the constructor assigns it, src/Frame.cc:840 vs :178, is kept as it is).  mvuRight / mvDepth bit-identical.
  static void DeviceStereoMatches(ORBextractor* left, ORBextractor* right, const std::vector<cv::KeyPoint>& mvKeys, const cv::Mat& mDescriptors,
                                  const std::vector<cv::KeyPoint>& mvKeysRight, const cv::Mat& mDescriptorsRight, float mb, float mbf,
                                  std::vector<float>& mvuRight, std::vector<float>& mvDepth) {
    const int N = (int)mvKeys.size(), Nr = (int)mvKeysRight.size();
    mvuRight.assign(N, -1.0f);
    mvDepth.assign(N, -1.0f);
    if (N == 0) return;
    if ((N && !mDescriptors.isContinuous()) || (Nr && !mDescriptorsRight.isContinuous()))
      throw std::runtime_error("ORBextractor::DeviceStereoMatches: descriptor matrices must be continuous (operator() makes them so)");
    int kept = 0;
    const int rc = orbx_stereo_matches(left->ctx_, right->ctx_, (const orbx_keypoint*)mvKeys.data(), mDescriptors.data, N,
                                       (const orbx_keypoint*)mvKeysRight.data(), Nr ? mDescriptorsRight.data : nullptr, Nr, mb, mbf, mvuRight.data(),
                                       mvDepth.data(), &kept);
    if (rc != ORBX_OK) throw std::runtime_error(std::string("ORBextractor::DeviceStereoMatches: ") + orbx_last_error(left->ctx_));
  }

 protected:
  int nfeatures;
  double scaleFactor;
  int nlevels;
  int iniThFAST;
  int minThFAST;
  std::vector<int> mnFeaturesPerLevel;
  std::vector<float> mvScaleFactor;
  std::vector<float> mvInvScaleFactor;
  std::vector<float> mvLevelSigma2;
  std::vector<float> mvInvLevelSigma2;

 private:
  orbx_ctx* ctx_ = nullptr;
  int cap_ = 0;
  bool keep_host_pyramid_ = true;
  std::vector<orbx_keypoint> kps_;
  std::vector<uint8_t> desc_;
};

}  // namespace ORB_SLAM3

#endif  // ORBEXTRACTOR_H
