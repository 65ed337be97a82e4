// TEST INFRASTRUCTURE — declarations only, never linked.  The part of OpenCV 4.x's public class interface the adapter headers touch,
// written down with the SIGNATURES of the real library (opencv2/core/mat.hpp, types.hpp, base.hpp as published: _InputArray /
// _OutputArray with their defaulted index arguments, Mat::step as a MatStep object, AUTO_STEP, the templated ptr<>, KeyPoint over
// Point2f) instead of the simplified look-alikes of include/orbx_cv_compat.h and oracle/ref_shims.  tests/test_adapters.py compiles
// include/ORBextractor.h, include/ORBVocabulary.h and include/orbx_cv_calibrate.h against it with -fsyntax-only: overload resolution,
// implicit conversions (Mat -> InputArray / OutputArray, MatStep -> size_t) and const-correctness are checked against the shapes a real
// OpenCV build presents.  Own text; no OpenCV source is copied (declarations restated from the documented API).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#define CV_VERSION_MAJOR 4
#define CV_VERSION_MINOR 4
#define CV_VERSION_REVISION 0
#define CV_VERSION "4.4.0-signatures-only"
#define CV_CN_SHIFT 3
#define CV_8U 0
#define CV_32F 5
#define CV_MAKETYPE(depth, cn) (((depth) & 7) + (((cn) - 1) << CV_CN_SHIFT))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_EXPORTS
#define CV_EXPORTS_W

typedef unsigned char uchar;

namespace cv {

template <typename _Tp> class Point_ { public: Point_(); Point_(_Tp _x, _Tp _y); _Tp x, y; };
typedef Point_<int> Point2i;
typedef Point_<float> Point2f;
typedef Point2i Point;
template <typename _Tp> class Size_ { public: Size_(); Size_(_Tp _width, _Tp _height); _Tp width, height; };
typedef Size_<int> Size;

class CV_EXPORTS_W KeyPoint {
 public:
  KeyPoint();
  KeyPoint(Point2f pt, float size, float angle = -1, float response = 0, int octave = 0, int class_id = -1);
  KeyPoint(float x, float y, float size, float angle = -1, float response = 0, int octave = 0, int class_id = -1);
  Point2f pt;
  float size, angle, response;
  int octave, class_id;
};

enum BorderTypes { BORDER_CONSTANT = 0, BORDER_REPLICATE = 1, BORDER_REFLECT = 2, BORDER_WRAP = 3, BORDER_REFLECT_101 = 4, BORDER_TRANSPARENT = 5,
                   BORDER_REFLECT101 = BORDER_REFLECT_101, BORDER_DEFAULT = BORDER_REFLECT_101, BORDER_ISOLATED = 16 };

class Mat;
class _OutputArray;

class CV_EXPORTS _InputArray {
 public:
  _InputArray();
  _InputArray(const Mat& m);
  template <typename _Tp> _InputArray(const std::vector<_Tp>& vec);
  Mat getMat(int idx = -1) const;
  bool empty() const;
  int type(int i = -1) const;
  bool isMat() const;
  ~_InputArray();
 protected:
  int flags;
  void* obj;
};

class CV_EXPORTS _OutputArray : public _InputArray {
 public:
  enum DepthMask { DEPTH_MASK_8U = 1, DEPTH_MASK_ALL = 127 };
  _OutputArray();
  _OutputArray(Mat& m);
  template <typename _Tp> _OutputArray(std::vector<_Tp>& vec);
  void create(Size sz, int type, int i = -1, bool allowTransposed = false, _OutputArray::DepthMask fixedDepthMask = static_cast<_OutputArray::DepthMask>(0)) const;
  void create(int rows, int cols, int type, int i = -1, bool allowTransposed = false, _OutputArray::DepthMask fixedDepthMask = static_cast<_OutputArray::DepthMask>(0)) const;
  void release() const;
  bool needed() const;
};

typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;
typedef const _OutputArray& InputOutputArray;

struct CV_EXPORTS MatStep {
  MatStep();
  explicit MatStep(size_t s);
  const size_t& operator[](int i) const;
  size_t& operator[](int i);
  operator size_t() const;
  MatStep& operator=(size_t s);
  size_t* p;
  size_t buf[2];
};

class CV_EXPORTS Mat {
 public:
  enum { AUTO_STEP = 0 };
  Mat();
  Mat(int rows, int cols, int type);
  Mat(Size size, int type);
  Mat(const Mat& m);
  Mat(int rows, int cols, int type, void* data, size_t step = AUTO_STEP);
  ~Mat();
  Mat& operator=(const Mat& m);
  Mat row(int y) const;
  Mat clone() const;
  void copyTo(OutputArray m) const;
  void create(int rows, int cols, int type);
  void release();
  bool isContinuous() const;
  bool isSubmatrix() const;
  size_t elemSize() const;
  int type() const;
  bool empty() const;
  uchar* ptr(int i0 = 0);
  const uchar* ptr(int i0 = 0) const;
  template <typename _Tp> _Tp* ptr(int i0 = 0);
  template <typename _Tp> const _Tp* ptr(int i0 = 0) const;
  int flags, dims, rows, cols;
  uchar* data;
  MatStep step;
};

CV_EXPORTS_W float fastAtan2(float y, float x);
int cvRound(double value);
int cvRound(float value);

}  // namespace cv
