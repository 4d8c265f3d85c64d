// oracle/_ref build shim for the Time-Surface node (TEST INFRASTRUCTURE): just enough of cv::Mat for TimeSurface.cpp to
// compile and run its per-pixel loop.  OpenCV itself (convertTo's rounding, medianBlur, remap, the undistortion maps) is
// NOT restated here: convertTo records the f64 image it is handed (esvo_ts_shim::captured) -- that is the product of the
// reference's own arithmetic -- and the later stages pass the image through unchanged.
#ifndef ESVO_REF_SHIM_TS_CV
#define ESVO_REF_SHIM_TS_CV
#include <cmath>
#include <memory>
#include <vector>
#define CV_8U 0
#define CV_64F 6
#define CV_32FC1 5
#define CV_INTER_LINEAR 1
namespace esvo_ts_shim {
inline std::vector<double>& captured() { static std::vector<double> v; return v; }
}
namespace cv {
struct Size { int width = 0, height = 0; Size() {} Size(int w, int h) : width(w), height(h) {} };
struct Point { int x = 0, y = 0; Point() {} Point(int x_, int y_) : x(x_), y(y_) {} };
struct Point2f { float x = 0, y = 0; Point2f() {} Point2f(float x_, float y_) : x(x_), y(y_) {} };
class Mat {
 public:
  int rows = 0, cols = 0, type_ = CV_64F;
  Mat() {}
  Mat(int r, int c, int t) : rows(r), cols(c), type_(t), d_(std::make_shared<std::vector<double>>((size_t)r * c, 0.0)) {}
  static Mat zeros(Size s, int t) { return Mat(s.height, s.width, t); }
  template <class T> T& at(int y, int x) { return (*d_)[(size_t)y * cols + x]; }
  template <class T> T& at(Point p) { return (*d_)[(size_t)p.y * cols + p.x]; }
  template <class T> T& at(int i) { return (*d_)[(size_t)i]; }
  Mat clone() const { Mat m(rows, cols, type_); if (d_) *m.d_ = *d_; return m; }
  void convertTo(Mat& dst, int t) const {
    esvo_ts_shim::captured() = d_ ? *d_ : std::vector<double>();
    Mat m(rows, cols, t);
    for (size_t i = 0; d_ && i < d_->size(); ++i) { double v = std::nearbyint((*d_)[i]); (*m.d_)[i] = v < 0 ? 0 : (v > 255 ? 255 : v); }
    dst = m;
  }
  Mat map(double a, double b) const { Mat m(rows, cols, type_); for (size_t i = 0; d_ && i < d_->size(); ++i) (*m.d_)[i] = a * (*d_)[i] + b; return m; }
  std::shared_ptr<std::vector<double>> d_;
};
// element-wise, in source order (cv::MatExpr folds scale and shift into one pass: only the ignore_polarity = true
// expression 255.0 * m, a single multiplication either way, is compared against)
inline Mat operator*(double a, const Mat& m) { return m.map(a, 0.0); }
inline Mat operator+(const Mat& m, double b) { return m.map(1.0, b); }
inline Mat operator/(const Mat& m, double b) { Mat r = m.clone(); for (auto& v : *r.d_) v = v / b; return r; }
template <class T> class Mat_ : public Mat {
 public:
  Mat_(int r, int c) : Mat(r, c, CV_64F), p_((size_t)r * c) {}
  T& operator()(int i) { return p_[(size_t)i]; }
  std::vector<T> p_;
};
inline void medianBlur(const Mat&, Mat&, int) {}
inline void remap(const Mat& src, Mat& dst, const Mat&, const Mat&, int) { dst = src; }
inline void initUndistortRectifyMap(const Mat&, const Mat&, const Mat&, const Mat&, Size, int, Mat&, Mat&) {}
template <class A, class B> void undistortPoints(const A&, B&, const Mat&, const Mat&, const Mat&, const Mat&) {}
namespace fisheye {
inline void initUndistortRectifyMap(const Mat&, const Mat&, const Mat&, const Mat&, Size, int, Mat&, Mat&) {}
template <class A, class B> void undistortPoints(const A&, B&, const Mat&, const Mat&, const Mat&, const Mat&) {}
}
}  // namespace cv
#endif
