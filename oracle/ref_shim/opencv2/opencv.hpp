// -*- C++ -*-
// oracle/_ref build shim (TEST INFRASTRUCTURE): just enough of the cv:: names for CameraSystem.cpp and
// TimeSurfaceObservation.h to compile unmodified.  OpenCV is NOT re-implemented here: the calibration
// products that CameraSystem::preComputeRectifiedCoordinate obtains from cv::undistortPoints /
// cv::initUndistortRectifyMap / cv::remap / cv::threshold are INJECTED by the harness
// (esvo_ref_shim::inject(), filled from esvo_amd/calib.py -- the same arrays the oracle and the GPU get), so
// the reference's own loops around those calls run on the same inputs.  Image filters (GaussianBlur, Sobel)
// abort: the harness hands the mapper already-filtered Time Surfaces.
#ifndef ESVO_REF_SHIM_OPENCV
#define ESVO_REF_SHIM_OPENCV
#include <Eigen/Eigen>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <memory>
#include <vector>

#define CV_MAJOR_VERSION 4
#define CV_8U 0
#define CV_32F 5
#define CV_64F 6
#define CV_32FC1 CV_32F
#define CV_8UC1 CV_8U

namespace esvo_ref_shim {
struct Injected {
  const short* disparity = nullptr;  // what the stand-in StereoSGBM::compute returns (node build: InitializationAtTime)

  const float* lut[2] = {nullptr, nullptr};     // per camera: 2*W*H, what undistortPoints returns for every raw pixel
  const uint8_t* mask[2] = {nullptr, nullptr};  // per camera: W*H 0/255, the thresholded remap of the all-ones image
                                                // (NULL: all valid)
  int W = 0, H = 0;
  int cam = 0;  // camera being set up: CameraSystem::loadCalibInfo does left, then right; remap() is the last
                // stand-in call of one preComputeRectifiedCoordinate and advances it
};
inline Injected& inject() { static Injected g; return g; }
}  // namespace esvo_ref_shim

typedef unsigned char uchar;
namespace cv {
enum { INTER_LINEAR = 1, THRESH_BINARY = 0 };

struct Size {
  int width = 0, height = 0;
  Size() {}
  Size(int w, int h) : width(w), height(h) {}
};
struct Point2f {
  float x = 0, y = 0;
  Point2f() {}
  Point2f(float x_, float y_) : x(x_), y(y_) {}
};

struct Scalar { double v0 = 0; Scalar() {} Scalar(double a) : v0(a) {} Scalar(double a, double, double) : v0(a) {} };
// dense 2-D image, one channel, stored as doubles whatever the nominal type
class Mat {
 public:
  int rows = 0, cols = 0, type_ = CV_64F;
  std::vector<double> v;
  Mat() {}
  Mat(int r, int c, int t) : rows(r), cols(c), type_(t), v((size_t)r * c, 0.0) {}
  Mat(Size s, int t, Scalar sc = Scalar()) : rows(s.height), cols(s.width), type_(t), v((size_t)s.height * s.width, sc.v0) {}
  template <class T> double& at(int r, int c) { return v[(size_t)r * cols + c]; }
  template <class T> double at(int r, int c) const { return v[(size_t)r * cols + c]; }
  Mat clone() const { return *this; }
  bool empty() const { return v.empty(); }
  static Mat ones(int r, int c, int t) { Mat m(r, c, t); m.v.assign(m.v.size(), 1.0); return m; }
  static Mat zeros(int r, int c, int t) { return Mat(r, c, t); }
  double& at(int r, int c) { return v[(size_t)r * cols + c]; }
  double at(int r, int c) const { return v[(size_t)r * cols + c]; }
  void convertTo(Mat& dst, int t, double alpha = 1.0) const {
    Mat o(rows, cols, t);
    for (size_t i = 0; i < v.size(); ++i) {
      double x = v[i] * alpha;
      if (t == CV_8U) { x = std::nearbyint(x); x = x < 0 ? 0 : (x > 255 ? 255 : x); }
      o.v[i] = x;
    }
    dst = o;
  }
};
template <class T> class Mat_ {
 public:
  int rows = 0, cols = 0;
  std::vector<T> v;
  Mat_() {}
  Mat_(int r, int c) : rows(r), cols(c), v((size_t)r * c) {}
  T& operator()(int i) { return v[(size_t)i]; }
  const T& operator()(int i) const { return v[(size_t)i]; }
};

template <class S, int R, int C, int O> void eigen2cv(const Eigen::Matrix<S, R, C, O>& src, Mat& dst) {
  dst = Mat((int)src.rows(), (int)src.cols(), CV_64F);
  for (int i = 0; i < dst.rows; ++i)
    for (int j = 0; j < dst.cols; ++j) dst.at(i, j) = (double)src(i, j);
}
template <class S, int R, int C, int O> void cv2eigen(const Mat& src, Eigen::Matrix<S, R, C, O>& dst) {
  dst.resize(src.rows, src.cols);
  for (int i = 0; i < src.rows; ++i)
    for (int j = 0; j < src.cols; ++j) dst(i, j) = (S)src.at(i, j);
}

inline void shim_lut(const Mat_<Point2f>& src, Mat_<Point2f>& dst) {
  const esvo_ref_shim::Injected& g = esvo_ref_shim::inject();
  if (g.cam > 1 || !g.lut[g.cam] || (int)src.v.size() != g.W * g.H) std::abort();
  const float* lut = g.lut[g.cam];
  for (size_t i = 0; i < src.v.size(); ++i) dst(i) = Point2f(lut[2 * i], lut[2 * i + 1]);
}
inline void undistortPoints(const Mat_<Point2f>& src, Mat_<Point2f>& dst, const Mat&, const Mat&, const Mat&, const Mat&) {
  shim_lut(src, dst);
}
inline void initUndistortRectifyMap(const Mat&, const Mat&, const Mat&, const Mat&, Size, int, Mat& m1, Mat& m2) {
  m1 = Mat();
  m2 = Mat();
}
// only ever called on the all-ones image: the result is the injected mask before thresholding (0 / 1)
inline void remap(const Mat& src, Mat& dst, const Mat&, const Mat&, int) {
  esvo_ref_shim::Injected& g = esvo_ref_shim::inject();
  if (g.cam > 1 || src.rows != g.H || src.cols != g.W) std::abort();
  const uint8_t* mask = g.mask[g.cam];
  dst = Mat(src.rows, src.cols, CV_32F);
  for (size_t i = 0; i < dst.v.size(); ++i) dst.v[i] = (!mask || mask[i]) ? 1.0 : 0.0;
  g.cam++;
}
inline double threshold(const Mat& src, Mat& dst, double thresh, double maxval, int) {
  Mat o(src.rows, src.cols, src.type_);
  for (size_t i = 0; i < o.v.size(); ++i) o.v[i] = src.v[i] > thresh ? maxval : 0.0;
  dst = o;
  return thresh;
}
inline void GaussianBlur(const Mat&, Mat&, Size, double) { std::abort(); }
inline void Sobel(const Mat&, Mat&, int, int, int) { std::abort(); }

// ---- node build (esvo_Mapping.cpp): stand-ins, NOT restatements of OpenCV ----------------------------------------
// medianBlur(3) on a 0/255 event mask (createDenoisingMask): 3x3 median, borders replicated
inline void medianBlur(const Mat& src, Mat& dst, int k) {
  if (k != 3) std::abort();
  Mat o(src.rows, src.cols, src.type_);
  for (int y = 0; y < src.rows; ++y)
    for (int x = 0; x < src.cols; ++x) {
      double w[9];
      int n = 0;
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          const int yy = y + dy < 0 ? 0 : (y + dy >= src.rows ? src.rows - 1 : y + dy);
          const int xx = x + dx < 0 ? 0 : (x + dx >= src.cols ? src.cols - 1 : x + dx);
          w[n++] = src.v[(size_t)yy * src.cols + xx];
        }
      for (int i = 1; i < 9; ++i) { const double t = w[i]; int j = i - 1; while (j >= 0 && w[j] > t) { w[j + 1] = w[j]; --j; } w[j + 1] = t; }
      o.v[(size_t)y * src.cols + x] = w[4];
    }
  dst = o;
}
template <class T> struct Ptr {
  std::shared_ptr<T> p;
  Ptr() {}
  Ptr(std::shared_ptr<T> q) : p(q) {}
  T* operator->() const { return p.get(); }
  explicit operator bool() const { return (bool)p; }
};
// the disparity image comes from the harness (esvo_ref_shim::inject().disparity: W*H int16, disparity * 16)
struct StereoSGBM {
  static Ptr<StereoSGBM> create(int, int, int, int = 0, int = 0, int = 0, int = 0, int = 0, int = 0, int = 0, int = 0) {
    return Ptr<StereoSGBM>(std::make_shared<StereoSGBM>());
  }
  void compute(const Mat& left, const Mat&, Mat& disp) const {
    disp = Mat(left.rows, left.cols, CV_64F);
    const short* d = esvo_ref_shim::inject().disparity;
    if (!d) std::abort();
    for (size_t i = 0; i < disp.v.size(); ++i) disp.v[i] = (double)d[i];
  }
};

namespace fisheye {
inline void undistortPoints(const Mat_<Point2f>& src, Mat_<Point2f>& dst, const Mat&, const Mat&, const Mat&, const Mat&) {
  shim_lut(src, dst);
}
inline void initUndistortRectifyMap(const Mat&, const Mat&, const Mat&, const Mat&, Size, int, Mat& m1, Mat& m2) {
  m1 = Mat();
  m2 = Mat();
}
}  // namespace fisheye
}  // namespace cv
#endif
