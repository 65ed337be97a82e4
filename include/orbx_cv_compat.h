/* orbx — the smallest cv:: surface the adapter headers (ORBextractor.h / ORBmatcher.h / ORBVocabulary.h) touch.
 *
 * When the real OpenCV core headers are on the include path they are used and this file defines nothing; the
 * look-alikes below exist so the adapters (and their tests) build in environments without OpenCV, such as this
 * repository's build container.  Own code: only the members the adapters and the reference's callers of this path
 * use (cv::Mat rows/cols/step/data/ptr/create/clone, cv::KeyPoint's seven fields, Input/OutputArray::getMat/create).
 */
#ifndef ORBX_CV_COMPAT_H
#define ORBX_CV_COMPAT_H

#if defined(__has_include)
#if __has_include(<opencv2/core/core.hpp>) && !defined(ORBX_FORCE_CV_COMPAT)
#include <opencv2/core/core.hpp>
#define ORBX_HAVE_OPENCV 1
#endif
#endif

#ifndef ORBX_HAVE_OPENCV
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_8UC1 0
#define CV_32F 5

namespace cv {

struct Point2f { float x = 0, y = 0; Point2f() {} Point2f(float x_, float y_) : x(x_), y(y_) {} };
struct Point { int x = 0, y = 0; Point() {} Point(int x_, int y_) : x(x_), y(y_) {} };

/* Field order and sizes match cv::KeyPoint (28 bytes): pt.x, pt.y, size, angle, response, octave, class_id. */
struct KeyPoint {
  Point2f pt;
  float size = 0, angle = -1, response = 0;
  int octave = 0, class_id = -1;
  KeyPoint() {}
  KeyPoint(float x, float y, float size_, float angle_ = -1, float response_ = 0, int octave_ = 0, int class_id_ = -1)
      : pt(x, y), size(size_), angle(angle_), response(response_), octave(octave_), class_id(class_id_) {}
};
static_assert(sizeof(KeyPoint) == 28, "cv::KeyPoint layout");

class Mat {
 public:
  int rows = 0, cols = 0;
  size_t step = 0;
  unsigned char* data = nullptr;
  Mat() {}
  Mat(int r, int c, int type) { create(r, c, type); }
  /* header over caller memory (not owned), like cv::Mat(rows, cols, type, data, step) */
  Mat(int r, int c, int type, void* d, size_t step_ = 0) : rows(r), cols(c), type_(type) {
    step = step_ ? step_ : (size_t)c * esz();
    data = (unsigned char*)d;
  }
  void create(int r, int c, int type) {
    if (r == rows && c == cols && type == type_ && data && own_) return;
    rows = r; cols = c; type_ = type; step = (size_t)c * esz();
    own_ = std::shared_ptr<unsigned char>(new unsigned char[(size_t)r * step + 1], std::default_delete<unsigned char[]>());
    data = own_.get();
  }
  void release() { own_.reset(); data = nullptr; rows = cols = 0; step = 0; }
  bool empty() const { return !data || rows * cols == 0; }
  int type() const { return type_; }
  bool isContinuous() const { return step == (size_t)cols * esz(); }
  Mat row(int r) const { Mat m(1, cols, type_, data + (size_t)r * step, step); m.own_ = own_; return m; }
  Mat clone() const {
    Mat m;
    if (data) { m.create(rows, cols, type_); for (int r = 0; r < rows; r++) std::memcpy(m.data + (size_t)r * m.step, data + (size_t)r * step, (size_t)cols * esz()); }
    return m;
  }
  template <typename T> T* ptr(int r = 0) { return reinterpret_cast<T*>(data + (size_t)r * step); }
  template <typename T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data + (size_t)r * step); }

 private:
  size_t esz() const { return type_ == CV_32F ? 4 : 1; }
  int type_ = CV_8U;
  std::shared_ptr<unsigned char> own_;
};

class _InputArray {
 public:
  _InputArray() {}
  _InputArray(const Mat& m) : m_(&m) {}
  Mat getMat() const { return m_ ? *m_ : Mat(); }
  bool empty() const { return !m_ || m_->empty(); }
 private:
  const Mat* m_ = nullptr;
};
class _OutputArray {
 public:
  _OutputArray(Mat& m) : m_(&m) {}
  void create(int r, int c, int type) const { m_->create(r, c, type); }
  void release() const { m_->release(); }
  Mat getMat() const { return *m_; }
 private:
  Mat* m_;
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;

}  // namespace cv
#endif /* !ORBX_HAVE_OPENCV */
#endif /* ORBX_CV_COMPAT_H */
