// TEST INFRASTRUCTURE: the OpenCV calls of src/Frame.cc beyond the container shim (oracle/ref_shims/opencv2): cv::norm(NORM_L1) of two
// CV_8U patches (:918), cv::vconcat (:1113), cv::undistortPoints (:766,:793, forwarded to the oracle's restatement mo_undistort_points —
// "recalled OpenCV semantics") and cv::BFMatcher(NORM_HAMMING).knnMatch(k = 2) (:43,:1144, forwarded to mo_knn2).
// Own code.
#pragma once
#include <vector>
#include <opencv2/core/core.hpp>

extern "C" {
void mo_undistort_points(const float* xy_in, int n, float fx, float fy, float cx, float cy, const float* dist, int ndist, float* xy_out);
void mo_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx, int32_t* dist);
}

namespace cv {

enum { NORM_L1 = 2, NORM_HAMMING = 6 };

static inline double norm(const Mat& a, const Mat& b, int type) {
  assert(type == NORM_L1 && a.type() == CV_8U && b.type() == CV_8U && a.rows == b.rows && a.cols == b.cols);
  (void)type;
  long long s = 0;
  for (int r = 0; r < a.rows; r++) {
    const unsigned char* pa = a.ptr(r);
    const unsigned char* pb = b.ptr(r);
    for (int c = 0; c < a.cols; c++) s += pa[c] > pb[c] ? pa[c] - pb[c] : pb[c] - pa[c];
  }
  return (double)s;
}

static inline void vconcat(const Mat& a, const Mat& b, Mat& dst) {
  assert(a.cols == b.cols || a.empty() || b.empty());
  Mat out(a.rows + b.rows, a.empty() ? b.cols : a.cols, a.empty() ? b.type() : a.type());
  for (int r = 0; r < a.rows; r++) std::memcpy(out.ptr(r), a.ptr(r), (size_t)a.cols * a.elemSize());
  for (int r = 0; r < b.rows; r++) std::memcpy(out.ptr(a.rows + r), b.ptr(r), (size_t)b.cols * b.elemSize());
  dst = out;
}

// points: N x 2 CV_32F (the reshape(2) / reshape(1) pair around the call is the identity on this shim's Mat); K, P: 3 x 3 CV_32F
static inline void undistortPoints(const Mat& src, Mat& dst, const Mat& K, const Mat& dist, const Mat& R, const Mat& P) {
  assert(src.type() == CV_32F && src.cols == 2 && R.empty() && K.type() == CV_32F && P.type() == CV_32F);
  // Frame.cc passes P = mK, the same intrinsics as K (the oracle's restatement takes one set)
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) assert(K.at<float>(i, j) == P.at<float>(i, j));
  (void)R; (void)P;
  std::vector<float> in((size_t)src.rows * 2), out((size_t)src.rows * 2), d;
  for (int i = 0; i < src.rows; i++) { in[2 * i] = src.at<float>(i, 0); in[2 * i + 1] = src.at<float>(i, 1); }
  const int nd = dist.rows * dist.cols;
  for (int i = 0; i < nd; i++) d.push_back(dist.at<float>(i));
  mo_undistort_points(in.data(), src.rows, K.at<float>(0, 0), K.at<float>(1, 1), K.at<float>(0, 2), K.at<float>(1, 2), d.data(), nd, out.data());
  dst.create(src.rows, 2, CV_32F);
  for (int i = 0; i < src.rows; i++) { dst.at<float>(i, 0) = out[2 * i]; dst.at<float>(i, 1) = out[2 * i + 1]; }
}

struct DMatch {
  int queryIdx = -1, trainIdx = -1, imgIdx = -1;
  float distance = 0.f;
};

class BFMatcher {
 public:
  explicit BFMatcher(int normType = NORM_HAMMING) { assert(normType == NORM_HAMMING); (void)normType; }
  void knnMatch(const Mat& q, const Mat& t, std::vector<std::vector<DMatch> >& matches, int k) const {
    assert(k == 2 && (q.empty() || q.cols == 32) && (t.empty() || t.cols == 32));
    (void)k;
    matches.clear();
    const int nq = q.rows, nt = t.rows;
    if (nq == 0) return;
    std::vector<unsigned char> qb((size_t)nq * 32), tb((size_t)nt * 32);
    for (int i = 0; i < nq; i++) std::memcpy(&qb[(size_t)i * 32], q.ptr(i), 32);
    for (int i = 0; i < nt; i++) std::memcpy(&tb[(size_t)i * 32], t.ptr(i), 32);
    std::vector<int32_t> idx((size_t)nq * 2, -1), dist((size_t)nq * 2, 0);
    mo_knn2(qb.data(), nq, tb.data(), nt, idx.data(), dist.data());
    matches.resize(nq);
    for (int i = 0; i < nq; i++)
      for (int j = 0; j < 2; j++)
        if (idx[2 * i + j] >= 0) { DMatch m; m.queryIdx = i; m.trainIdx = idx[2 * i + j]; m.imgIdx = 0; m.distance = (float)dist[2 * i + j]; matches[i].push_back(m); }
  }
};

}  // namespace cv
