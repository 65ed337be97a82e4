// TEST INFRASTRUCTURE: the smallest cv:: surface the reference's vendored DBoW2 sources touch
// (Thirdparty/DBoW2/DBoW2/{FORB.cpp,TemplatedVocabulary.h}), so those files can be compiled FROM
// /root/reference, unmodified, into oracle/_ref/ without OpenCV.  Own code, not OpenCV's.
// cv::FileStorage/FileNode only have to parse (the YAML save/load members are never called).
#pragma once
#include <cmath>
#include <cstdlib>
#include <iostream>
#include <sstream>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#define CV_8U 0
#define CV_32F 5

namespace cv {

class Mat {
 public:
  int rows = 0, cols = 0;
  Mat() {}
  Mat(int r, int c, int type) { create(r, c, type); }
  void create(int r, int c, int type) {
    if (r == rows && c == cols && type == type_ && buf_) return;
    rows = r; cols = c; type_ = type;
    buf_ = std::shared_ptr<unsigned char>(new unsigned char[(size_t)r * c * esz()], std::default_delete<unsigned char[]>());
  }
  void release() { buf_.reset(); rows = cols = 0; }
  bool empty() const { return !buf_ || rows * cols == 0; }
  Mat clone() const {
    Mat m;
    if (buf_) { m.create(rows, cols, type_); std::memcpy(m.buf_.get(), buf_.get(), (size_t)rows * cols * esz()); }
    return m;
  }
  static Mat zeros(int r, int c, int type) {
    Mat m(r, c, type);
    std::memset(m.buf_.get(), 0, (size_t)r * c * m.esz());
    return m;
  }
  template <typename T> T* ptr(int row = 0) { return reinterpret_cast<T*>(buf_.get() + (size_t)row * cols * esz()); }
  template <typename T> const T* ptr(int row = 0) const { return reinterpret_cast<const T*>(buf_.get() + (size_t)row * cols * esz()); }

 private:
  size_t esz() const { return type_ == CV_32F ? 4 : 1; }
  int type_ = CV_8U;
  std::shared_ptr<unsigned char> buf_;
};

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
