// TEST INFRASTRUCTURE: an OpenCV *container* shim — own code, not OpenCV's — so that files of the reference can be
// compiled FROM /root/reference, unmodified, into oracle/_ref/ without OpenCV:
//   * Thirdparty/DBoW2/DBoW2/{FORB.cpp,TemplatedVocabulary.h,...}      (oracle/ref_fragments.mk: libref_dbow2.so)
//   * src/ORBmatcher.cc                                                (libref_orbmatcher / ref_matcher_world)
//   * src/ORBextractor.cc                                              (libref_orbextractor.so)
// It holds containers only: cv::Mat (shared buffer, ROI views, step), KeyPoint, Point, Size, Rect, Input/OutputArray
// and the cvRound family.  The five ALGORITHMS the extractor calls (FAST, resize, GaussianBlur, fastAtan2 and the
// reflect-101 border) are declared in opencv2/imgproc/imgproc.hpp / features2d.hpp and forward to the oracle's
// isolated primitives (orbo_*), which remain "recalled OpenCV semantics" (DESIGN.md §2).
// cv::FileStorage/FileNode only have to parse (DBoW2's YAML save/load members are never called).
#pragma once
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#define CV_8U 0
#define CV_8UC1 0
#define CV_32F 5
#define CV_PI 3.1415926535897932384626433832795

typedef unsigned char uchar;

// cvRound: round half to even (lrint / cvtss2si), for float and double; cvFloor / cvCeil as in OpenCV's fast_math.hpp
static inline int cvRound(float v) { return (int)lrintf(v); }
static inline int cvRound(double v) { return (int)lrint(v); }
static inline int cvRound(int v) { return v; }
static inline int cvFloor(float v) { int i = (int)v; return i - (i > v); }
static inline int cvFloor(double v) { int i = (int)v; return i - (i > v); }
static inline int cvCeil(float v) { int i = (int)v; return i + (i < v); }
static inline int cvCeil(double v) { int i = (int)v; return i + (i < v); }

namespace cv {

template <typename T> struct Point_ {
  T x, y;
  Point_() : x(0), y(0) {}
  Point_(T x_, T y_) : x(x_), y(y_) {}
  template <typename U> Point_(const Point_<U>& o) : x((T)o.x), y((T)o.y) {}
  Point_& operator*=(float s) { x = (T)(x * s); y = (T)(y * s); return *this; }
};
// int Point from float arguments truncates like saturate_cast<int>(float) does NOT: OpenCV's Point2i(float, float) is an
// implicit float -> int conversion of each argument (truncation), which is what `cv::Point2i(hX*static_cast<float>(i),0)`
// at src/ORBextractor.cc:571 relies on; the template constructor above gives exactly that.
typedef Point_<int> Point;
typedef Point_<int> Point2i;
typedef Point_<float> Point2f;
template <typename T> static inline Point_<T> operator+(const Point_<T>& a, const Point_<T>& b) { return Point_<T>(a.x + b.x, a.y + b.y); }
template <typename T> static inline Point_<T> operator-(const Point_<T>& a, const Point_<T>& b) { return Point_<T>(a.x - b.x, a.y - b.y); }

struct Point3f { float x = 0, y = 0, z = 0; Point3f() {} Point3f(float a, float b, float c) : x(a), y(b), z(c) {} };

struct Size { int width = 0, height = 0; Size() {} Size(int w, int h) : width(w), height(h) {} };
struct Rect { int x = 0, y = 0, width = 0, height = 0; Rect() {} Rect(int x_, int y_, int w, int h) : x(x_), y(y_), width(w), height(h) {} };

/* Field order and sizes match cv::KeyPoint (28 bytes): pt.x, pt.y, size, angle, response, octave, class_id. */
struct KeyPoint {
  Point2f pt;
  float size = 0, angle = -1, response = 0;
  int octave = 0, class_id = -1;
  KeyPoint() {}
  KeyPoint(Point2f p, float size_, float angle_ = -1, float response_ = 0, int octave_ = 0, int class_id_ = -1)
      : pt(p), size(size_), angle(angle_), response(response_), octave(octave_), class_id(class_id_) {}
  KeyPoint(float x, float y, float size_, float angle_ = -1, float response_ = 0, int octave_ = 0, int class_id_ = -1)
      : pt(x, y), size(size_), angle(angle_), response(response_), octave(octave_), class_id(class_id_) {}
};
static_assert(sizeof(KeyPoint) == 28, "cv::KeyPoint layout");

class _OutputArray;

class Mat {
 public:
  int rows = 0, cols = 0;
  size_t step = 0;
  unsigned char* data = nullptr;
  Mat() {}
  Mat(int r, int c, int type) { create(r, c, type); }
  Mat(Size s, int type) { create(s.height, s.width, type); }
  /* header over caller memory (not owned), like cv::Mat(rows, cols, type, data, step) */
  Mat(int r, int c, int type, void* d, size_t step_ = 0) : rows(r), cols(c), type_(type) {
    step = step_ ? step_ : (size_t)c * esz();
    data = (unsigned char*)d;
  }
  void create(int r, int c, int type) {
    if (r == rows && c == cols && type == type_ && data) return;   // cv::Mat::create keeps a matching buffer (ROI or not)
    rows = r; cols = c; type_ = type; step = (size_t)c * esz();
    own_ = std::shared_ptr<unsigned char>(new unsigned char[(size_t)r * step + 8], std::default_delete<unsigned char[]>());
    data = own_.get();
  }
  void create(Size s, int type) { create(s.height, s.width, type); }
  void release() { own_.reset(); data = nullptr; rows = cols = 0; step = 0; }
  bool empty() const { return !data || rows * cols == 0; }
  size_t total() const { return (size_t)rows * cols; }
  int type() const { return type_; }
  Size size() const { return Size(cols, rows); }
  size_t step1() const { return step / esz(); }
  size_t elemSize() const { return esz(); }
  bool isContinuous() const { return rows <= 1 || step == (size_t)cols * esz(); }
  bool isSubmatrix() const { return sub_; }
  Mat operator()(const Rect& r) const {
    Mat m(r.height, r.width, type_, data + (size_t)r.y * step + (size_t)r.x * esz(), step);
    m.own_ = own_; m.sub_ = true;
    return m;
  }
  Mat rowRange(int a, int b) const { return (*this)(Rect(0, a, cols, b - a)); }
  Mat colRange(int a, int b) const { return (*this)(Rect(a, 0, b - a, rows)); }
  Mat row(int r) const { return rowRange(r, r + 1); }
  Mat clone() const {
    Mat m;
    if (data) { m.create(rows, cols, type_); for (int r = 0; r < rows; r++) std::memcpy(m.data + (size_t)r * m.step, data + (size_t)r * step, (size_t)cols * esz()); }
    return m;
  }
  void copyTo(Mat& dst) const {
    dst.create(rows, cols, type_);
    for (int r = 0; r < rows; r++) std::memmove(dst.data + (size_t)r * dst.step, data + (size_t)r * step, (size_t)cols * esz());
  }
  inline void copyTo(const _OutputArray& dst) const;   // cv::Mat::copyTo(OutputArray): also takes a temporary header (a row view)
  static Mat zeros(int r, int c, int type) {
    Mat m(r, c, type);
    std::memset(m.data, 0, (size_t)r * m.step);
    return m;
  }
  template <typename T> T& at(int r, int c) { return *reinterpret_cast<T*>(data + (size_t)r * step + (size_t)c * sizeof(T)); }
  template <typename T> const T& at(int r, int c) const { return *reinterpret_cast<const T*>(data + (size_t)r * step + (size_t)c * sizeof(T)); }
  /* at(i): element i of a single-row or single-column matrix (cv::Mat::at(int i0)), as src/Frame.cc:749 reads mDistCoef */
  template <typename T> T& at(int i) { return rows == 1 ? at<T>(0, i) : at<T>(i, 0); }
  template <typename T> const T& at(int i) const { return rows == 1 ? at<T>(0, i) : at<T>(i, 0); }
  /* reshape(cn): this shim has no channel dimension — an N x 2 float matrix stays N x 2 whether it is read as 1 or 2 channels
     (src/Frame.cc:765-767: reshape(2), undistortPoints, reshape(1)) */
  Mat reshape(int) const { return *this; }
  unsigned char* ptr(int r = 0) { return data + (size_t)r * step; }
  const unsigned char* ptr(int r = 0) const { return data + (size_t)r * step; }
  template <typename T> T* ptr(int r = 0) { return reinterpret_cast<T*>(data + (size_t)r * step); }
  template <typename T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data + (size_t)r * step); }

 private:
  size_t esz() const { return type_ == CV_32F ? 4 : 1; }
  int type_ = CV_8U;
  bool sub_ = false;
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
  _OutputArray(const Mat& m) : m_(const_cast<Mat*>(&m)) {}   // like OpenCV: a header whose buffer is written through
  void create(int r, int c, int type) const { m_->create(r, c, type); }
  void release() const { m_->release(); }
  Mat getMat() const { return *m_; }
 private:
  Mat* m_;
};
inline void Mat::copyTo(const _OutputArray& dst) const { Mat d = dst.getMat(); copyTo(d); }
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;

class FileNode {
 public:
  FileNode operator[](const char*) const { std::abort(); }
  FileNode operator[](const std::string&) const { std::abort(); }
  FileNode operator[](int) const { std::abort(); }
  size_t size() const { std::abort(); }
  operator int() const { std::abort(); }
  operator double() const { std::abort(); }
  operator float() const { std::abort(); }
  operator std::string() const { std::abort(); }
};

class FileStorage {
 public:
  enum { READ = 0, WRITE = 1 };
  FileStorage(const char*, int) { std::abort(); }
  FileStorage(const std::string&, int) { std::abort(); }
  bool isOpened() const { return false; }
  void release() {}
  FileNode operator[](const char*) const { std::abort(); }
  FileNode operator[](const std::string&) const { std::abort(); }
};
template <typename T> FileStorage& operator<<(FileStorage& fs, const T&) { return fs; }

}  // namespace cv
