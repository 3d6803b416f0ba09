// Minimal stand-in for <opencv2/opencv.hpp>: only what the reference's HEADERS on the hot path's include chain mention
// (utils/sensor_utils/cameras.hpp, utils/utils.h, data_loader/data_parsers/base_parser.h).  Syntax-only test infrastructure
// (tests/test_reference_compiles_against_boundary.py): declarations, no behaviour.
#pragma once
#include <ostream>
#include <string>
#include <vector>
namespace cv {
struct Size { Size(int = 0, int = 0) {} int width = 0, height = 0; };
struct Mat {
  Mat() = default;
  Mat(int, int, int) {}
  int rows = 0, cols = 0;
  bool empty() const { return true; }
  int channels() const { return 0; }
  int type() const { return 0; }
  unsigned char *data = nullptr;
  template <class T> T &at(int, int = 0) { static T t; return t; }
  template <class T> const T &at(int, int = 0) const { static T t; return t; }
  Mat clone() const { return *this; }
  void convertTo(Mat &, int, double = 1, double = 0) const {}
  template <class T> T *ptr(int = 0) { return nullptr; }
  size_t total() const { return 0; }
  void setTo(double) {}
  void setTo(double, const Mat &) {}
};
inline Mat operator>(const Mat &m, double) { return m; }
inline Mat operator-(const Mat &m, double) { return m; }
inline Mat operator/(const Mat &m, double) { return m; }
inline Mat operator~(const Mat &m) { return m; }
inline std::ostream &operator<<(std::ostream &o, const Mat &) { return o; }
template <class T> struct Mat_ : Mat {
  Mat_() = default;
  Mat_(int, int) {}
  static Mat_ eye(int, int) { return Mat_(); }
  struct Init { Init &operator,(double) { return *this; } operator Mat() const { return Mat(); } operator Mat_<T>() const { return Mat_<T>(); } };
  Init operator<<(double) { return Init(); }
};
enum InterpolationFlags { INTER_LINEAR = 1 };
enum ColormapTypes { COLORMAP_TURBO = 20 };
enum ColorConversionCodes { COLOR_RGB2BGR = 4 };
inline void minMaxLoc(const Mat &, double *, double *, void * = nullptr, void * = nullptr, const Mat & = Mat()) {}
inline void applyColorMap(const Mat &, Mat &, int) {}
inline void cvtColor(const Mat &, Mat &, int) {}
inline bool imwrite(const std::string &, const Mat &) { return true; }
inline void resize(const Mat &, Mat &, Size, double = 0, double = 0, int = 1) {}
struct VideoWriter {
  VideoWriter() = default;
  VideoWriter(const std::string &, int, double, Size, bool = true) {}
  static int fourcc(char, char, char, char) { return 0; }
  bool isOpened() const { return false; }
  void release() {}
  void write(const Mat &) {}
};
enum { CV_32FC1 = 5, CV_32FC3 = 21, CV_8UC3 = 16, CV_16SC2 = 11, CV_32F = 5 };
inline Mat getOptimalNewCameraMatrix(const Mat &, const Mat &, Size, double, Size = Size(), void * = nullptr, bool = false) { return Mat(); }
inline void initUndistortRectifyMap(const Mat &, const Mat &, const Mat &, const Mat &, Size, int, Mat &, Mat &) {}
inline void remap(const Mat &, Mat &, const Mat &, const Mat &, int) {}
namespace fisheye {
inline void initUndistortRectifyMap(const Mat &, const Mat &, const Mat &, const Mat &, Size, int, Mat &, Mat &) {}
}  // namespace fisheye
}  // namespace cv
#ifndef CV_32FC1
#define CV_32FC1 5
#define CV_16SC2 11
#endif
#ifndef CV_8UC1
#define CV_8UC1 0
#define CV_8UC3 16
#define CV_16UC1 2
#endif
