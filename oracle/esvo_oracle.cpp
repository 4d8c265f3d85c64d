// esvo_oracle.cpp — CPU oracle (TEST INFRASTRUCTURE ONLY; see esvo_oracle.h).
//
// A dependency-free, double-precision restatement of the ESVO reference hot path.  Every
// function cites the reference file:line it follows (paths relative to the ESVO repository).
// PARITY: the mapper (block matching, residual functor, fusion, clean, regulariser) is pinned to the
// reference's own sources through oracle/_ref (esvo_oracle.h).  UNPINNED: the
// third-party pieces absent from /root/reference (OpenCV image ops and StereoSGBM, the Eigen LM
// driver, PCL VoxelGrid), restated from their published algorithms.
//
// Build: g++ -O2 -std=c++17 -ffp-contract=off -fPIC -shared   (see oracle/Makefile)
// -ffp-contract=off matters: the GPU kernels are built the same way so that identical
// expression sequences give identical IEEE-754 results.

#include "esvo_oracle.h"

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <deque>
#include <limits>
#include <map>
#include <thread>
#include <vector>

namespace {

// ---------------------------------------------------------------------------------------------
// ros::Time / ros::Duration arithmetic (roscpp time.h)
// ---------------------------------------------------------------------------------------------
inline double time_to_sec(uint32_t sec, uint32_t nsec) { return (double)sec + 1e-9 * (double)nsec; }
inline double ns_to_sec(uint64_t ns) {
  return time_to_sec((uint32_t)(ns / 1000000000ull), (uint32_t)(ns % 1000000000ull));
}
inline uint64_t ev_ns(const esvo_event_t& e) { return (uint64_t)e.sec * 1000000000ull + e.nsec; }
inline double ev_sec(const esvo_event_t& e) { return time_to_sec(e.sec, e.nsec); }
// ros::Time(double): TimeBase::fromSec
inline uint64_t ros_time_from_sec(double t) {
  int64_t sec64 = (int64_t)std::floor(t);
  uint32_t sec = (uint32_t)sec64;
  uint32_t nsec = (uint32_t)std::round((t - sec) * 1e9);
  sec += (nsec / 1000000000ul);
  nsec %= 1000000000ul;
  return (uint64_t)sec * 1000000000ull + nsec;
}
// (T - t).toSec() for T > t: Duration{sec,nsec normalised}.toSec()
inline double duration_to_sec(uint64_t later_ns, uint64_t earlier_ns) {
  int64_t d = (int64_t)(later_ns - earlier_ns);
  int64_t sec = d / 1000000000ll, nsec = d % 1000000000ll;
  if (nsec < 0) { nsec += 1000000000ll; sec -= 1; }
  return (double)sec + 1e-9 * (double)nsec;
}

// cvRound: round half to even (OpenCV uses lrint/SSE cvtsd2si under the default rounding mode)
inline int cv_round(double v) { return (int)std::nearbyint(v); }
inline int cv_round_f(float v) { return (int)std::nearbyintf(v); }
// x*x stands for the reference's sq(x) (identical up to libm's last bit; the GPU uses x*x)
inline double sq(double x) { return x * x; }
inline uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// ---------------------------------------------------------------------------------------------
// OpenCV image primitives
// ---------------------------------------------------------------------------------------------
void median_u8(const uint8_t* src, uint8_t* dst, int w, int h, int k) {
  // cv::medianBlur(ksize = 2k + 1) on CV_8U (TimeSurface.cpp:130-131): the exact (2k+1)^2 median, BORDER_REPLICATE
  const int n = (2 * k + 1) * (2 * k + 1);
  std::vector<uint8_t> v((size_t)n);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      int q = 0;
      for (int dy = -k; dy <= k; ++dy)
        for (int dx = -k; dx <= k; ++dx) {
          int yy = std::min(std::max(y + dy, 0), h - 1);
          int xx = std::min(std::max(x + dx, 0), w - 1);
          v[q++] = src[yy * w + xx];
        }
      std::nth_element(v.begin(), v.begin() + n / 2, v.end());
      dst[y * w + x] = v[n / 2];
    }
}
void median3_u8(const uint8_t* src, uint8_t* dst, int w, int h) { median_u8(src, dst, w, h, 1); }

void remap_bilinear_u8(const uint8_t* src, uint8_t* dst, int w, int h, const float* map_x,
                       const float* map_y) {
  // cv::remap(CV_8U, CV_32FC1 maps, INTER_LINEAR, BORDER_CONSTANT 0): coordinates quantised
  // to 1/32 px (INTER_BITS=5), 15-bit fixed-point weights (INTER_REMAP_COEF_SCALE=32768).
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      int i = y * w + x;
      int sx = cv_round_f(map_x[i] * 32.f);
      int sy = cv_round_f(map_y[i] * 32.f);
      int ix = sx >> 5, iy = sy >> 5;
      int fx = sx & 31, fy = sy & 31;
      int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32;
      int w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
      auto tap = [&](int xx, int yy) -> int {
        return (xx >= 0 && xx < w && yy >= 0 && yy < h) ? src[yy * w + xx] : 0;
      };
      int v = w00 * tap(ix, iy) + w01 * tap(ix + 1, iy) + w10 * tap(ix, iy + 1) +
              w11 * tap(ix + 1, iy + 1);
      dst[i] = (uint8_t)((v + 16384) >> 15);
    }
}

inline int reflect101(int p, int n) {
  if (n == 1) return 0;
  while (p < 0 || p >= n) {
    if (p < 0) p = -p;
    if (p >= n) p = 2 * (n - 1) - p;
  }
  return p;
}

void gaussian5_u8(const uint8_t* src, uint8_t* dst, int w, int h) {
  // cv::GaussianBlur(8u, 5x5, sigma=0): separable [1 4 6 4 1]/16, BORDER_REFLECT_101; the
  // oracle's documented choice for the version-dependent last bit: integer accumulate,
  // (sum + 128) >> 8 once at the end.
  static const int k[5] = {1, 4, 6, 4, 1};
  std::vector<int> tmp((size_t)w * h);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      int s = 0;
      for (int d = -2; d <= 2; ++d) s += k[d + 2] * src[y * w + reflect101(x + d, w)];
      tmp[y * w + x] = s;
    }
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      int s = 0;
      for (int d = -2; d <= 2; ++d) s += k[d + 2] * tmp[reflect101(y + d, h) * w + x];
      dst[y * w + x] = sat_u8((s + 128) >> 8);
    }
}

}  // namespace

// =============================================================================================
// Time Surface (esvo_time_surface/)
// =============================================================================================
struct orc_ts {
  int W, H, qlen;
  std::vector<std::deque<esvo_event_t>> q;  // EventQueueMat::eqMat_, TimeSurface.h:95
  bool have_newest = false;
  esvo_event_t newest;                      // events_.back(): the event with the largest stamp so far (the later arrival on a tie)
};

extern "C" orc_ts_handle orc_ts_create(int width, int height, int queue_len) {
  orc_ts* t = new orc_ts();
  t->W = width; t->H = height; t->qlen = queue_len;
  t->q.assign((size_t)width * height, {});
  return t;
}
extern "C" void orc_ts_destroy(orc_ts_handle h) { delete h; }
extern "C" void orc_ts_clear(orc_ts_handle h) { h->q.assign((size_t)h->W * h->H, {}); h->have_newest = false; }

extern "C" void orc_ts_push(orc_ts_handle h, const esvo_event_t* ev, size_t n) {
  // TimeSurface::eventsCallback (TimeSurface.cpp:403-425), literally: the arriving event is insertion-sorted into events_
  // (`while (events_[i].ts > e.ts)`: behind every event with a stamp <= its own) and then events_.BACK() -- not the arriving
  // event -- goes into EventQueueMat::insertEvent (TimeSurface.h:39-50: bounds check, push_back, trim the per-pixel queue to
  // qlen).  events_.back() is the event with the largest stamp so far, the later arrival on a tie; only it is kept here.  For
  // time-sorted input that is the arriving event; an event that arrives LATE is never inserted, the newest one is inserted again
  // (Appendix A-1).
  auto ns = [](const esvo_event_t& e) { return (unsigned long long)e.sec * 1000000000ull + e.nsec; };
  for (size_t i = 0; i < n; ++i) {
    if (!h->have_newest || ns(ev[i]) >= ns(h->newest)) { h->newest = ev[i]; h->have_newest = true; }
    const esvo_event_t& e = h->newest;
    if (e.x >= h->W || e.y >= h->H) continue;
    auto& eq = h->q[(size_t)e.x + (size_t)h->W * e.y];
    eq.push_back(e);
    while ((int)eq.size() > h->qlen) eq.pop_front();
  }
}

extern "C" void orc_ts_render(orc_ts_handle h, uint64_t t_ns, double decay_ms, int ignore_polarity,
                              int median_blur_kernel_size, const float* map_x, const float* map_y,
                              uint8_t* out, uint8_t* out_prefilter) {
  // TimeSurface::createTimeSurfaceAtTime, TimeSurface.cpp:52-152, BACKWARD mode
  const int W = h->W, H = h->H;
  const double decay_sec = decay_ms / 1000.0;  // :60
  std::vector<double> ts((size_t)W * H, 0.0);  // :62
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      // EventQueueMat::getMostRecentEventBeforeT, TimeSurface.h:52-75: reverse scan, strict <
      const auto& eq = h->q[(size_t)x + (size_t)W * y];
      const esvo_event_t* found = nullptr;
      for (auto it = eq.rbegin(); it != eq.rend(); ++it)
        if (ev_ns(*it) < t_ns) { found = &*it; break; }
      if (!found) continue;
      if (!(ev_sec(*found) > 0)) continue;  // :73
      const double dt = duration_to_sec(t_ns, ev_ns(*found));  // :75
      double polarity = found->polarity ? 1.0 : -1.0;          // :76
      double expVal = std::exp(-dt / decay_sec);               // :77
      if (!ignore_polarity) expVal *= polarity;                // :78-79
      ts[(size_t)y * W + x] = expVal;                          // :83
    }
  std::vector<uint8_t> img((size_t)W * H), med((size_t)W * H);
  for (size_t i = 0; i < ts.size(); ++i) {
    double v = ignore_polarity ? 255.0 * ts[i] : 255.0 * (ts[i] + 1.0) / 2.0;  // :123-126
    img[i] = sat_u8(cv_round(v));                                               // :127 convertTo
  }
  if (out_prefilter) std::memcpy(out_prefilter, img.data(), img.size());
  const uint8_t* cur = img.data();
  if (median_blur_kernel_size > 0) {  // :130-131 (kernel 2k+1; only k=1 shipped)
    median_u8(img.data(), med.data(), W, H, median_blur_kernel_size);
    cur = med.data();
  }
  if (map_x && map_y)
    remap_bilinear_u8(cur, out, W, H, map_x, map_y);  // :149
  else
    std::memcpy(out, cur, (size_t)W * H);
}

extern "C" void orc_ts_render_forward(orc_ts_handle h, uint64_t t_ns, double decay_ms, int ignore_polarity,
                                      int median_blur_kernel_size, const float* rect_lut, uint8_t* out, double* out_f64) {
  // TimeSurface::createTimeSurfaceAtTime, TimeSurface.cpp:52-135, FORWARD mode (:85-116): every pixel's decayed value is
  // splatted bilinearly at the pixel's rectified position (precomputed_rectified_points_, :363-399: the float output of
  // cv::undistortPoints, here rect_lut = (u, v) per raw pixel), in raster order of the SOURCE pixels, with a clamp to 1
  // after every add.  The image is rectified by construction: no remap follows (:137-141).
  const int W = h->W, H = h->H;
  const double decay_sec = decay_ms / 1000.0;
  std::vector<double> ts((size_t)W * H, 0.0);
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      const auto& eq = h->q[(size_t)x + (size_t)W * y];
      const esvo_event_t* found = nullptr;
      for (auto it = eq.rbegin(); it != eq.rend(); ++it)
        if (ev_ns(*it) < t_ns) { found = &*it; break; }
      if (!found) continue;
      if (!(ev_sec(*found) > 0)) continue;
      const double dt = duration_to_sec(t_ns, ev_ns(*found));
      double polarity = found->polarity ? 1.0 : -1.0;
      double expVal = std::exp(-dt / decay_sec);
      if (!ignore_polarity) expVal *= polarity;
      const double u = (double)rect_lut[2 * ((size_t)y * W + x)], v = (double)rect_lut[2 * ((size_t)y * W + x) + 1];  // :88
      if (u >= 0 && v >= 0) {
        const size_t u_i = (size_t)std::floor(u), v_i = (size_t)std::floor(v);  // :92-93
        if (u_i + 1 < (size_t)W && v_i + 1 < (size_t)H) {
          const double fu = u - u_i, fv = v - v_i, fu1 = 1.0 - fu, fv1 = 1.0 - fv;
          double* p00 = &ts[v_i * W + u_i];
          double* p01 = &ts[v_i * W + u_i + 1];
          double* p10 = &ts[(v_i + 1) * W + u_i];
          double* p11 = &ts[(v_i + 1) * W + u_i + 1];
          *p00 += fu1 * fv1 * expVal;  // :101-104
          *p01 += fu * fv1 * expVal;
          *p10 += fu1 * fv * expVal;
          *p11 += fu * fv * expVal;
          if (*p00 > 1) *p00 = 1;      // :106-113
          if (*p01 > 1) *p01 = 1;
          if (*p10 > 1) *p10 = 1;
          if (*p11 > 1) *p11 = 1;
        }
      }
    }
  std::vector<uint8_t> img((size_t)W * H);
  for (size_t i = 0; i < ts.size(); ++i) {
    const double v = ignore_polarity ? 255.0 * ts[i] : 255.0 * (ts[i] + 1.0) / 2.0;  // :123-126
    if (out_f64) out_f64[i] = v;
    img[i] = sat_u8(cv_round(v));
  }
  if (median_blur_kernel_size > 0) median_u8(img.data(), out, W, H, median_blur_kernel_size);
  else std::memcpy(out, img.data(), img.size());
}

extern "C" void orc_median3_u8(const uint8_t* s, uint8_t* d, int w, int h) { median3_u8(s, d, w, h); }
extern "C" void orc_remap_bilinear_u8(const uint8_t* s, uint8_t* d, int w, int h, const float* mx,
                                      const float* my) { remap_bilinear_u8(s, d, w, h, mx, my); }
extern "C" void orc_gaussian5_u8(const uint8_t* s, uint8_t* d, int w, int h) { gaussian5_u8(s, d, w, h); }

// =============================================================================================
// Mapper (esvo_core/)
// =============================================================================================
namespace {

struct Mat4 { double m[16]; };

inline Mat4 mat4_mul(const Mat4& a, const Mat4& b) {
  Mat4 c;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      c.m[i * 4 + j] = ((a.m[i * 4 + 0] * b.m[0 * 4 + j] + a.m[i * 4 + 1] * b.m[1 * 4 + j]) +
                        a.m[i * 4 + 2] * b.m[2 * 4 + j]) + a.m[i * 4 + 3] * b.m[3 * 4 + j];
  return c;
}
// kindr QuatTransformation::inverse(): rigid inverse [R^T | -R^T t]
inline Mat4 rigid_inverse(const Mat4& a) {
  Mat4 c;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) c.m[i * 4 + j] = a.m[j * 4 + i];
  for (int i = 0; i < 3; ++i)
    c.m[i * 4 + 3] = -((c.m[i * 4 + 0] * a.m[3] + c.m[i * 4 + 1] * a.m[7]) + c.m[i * 4 + 2] * a.m[11]);
  c.m[12] = c.m[13] = c.m[14] = 0.0; c.m[15] = 1.0;
  return c;
}

// PerspectiveCamera (CameraSystem.cpp:114-148)
struct Camera {
  int W = 0, H = 0;
  double P[12];
  double Kinv[9];   // inverse of P[:, :3]
  double Kinv_t[3]; // Kinv * P[:, 3]
  std::vector<float> lut;     // 2*W*H
  std::vector<uint8_t> mask;  // W*H
  void init(const esvo_calib_t* c) {
    W = c->width; H = c->height;
    std::memcpy(P, c->P, sizeof(P));
    const double a = P[0], b = P[1], cc = P[2], d = P[4], e = P[5], f = P[6], g = P[8], hh = P[9], i = P[10];
    const double det = a * (e * i - f * hh) - b * (d * i - f * g) + cc * (d * hh - e * g);
    const double id = 1.0 / det;
    Kinv[0] = (e * i - f * hh) * id; Kinv[1] = (cc * hh - b * i) * id; Kinv[2] = (b * f - cc * e) * id;
    Kinv[3] = (f * g - d * i) * id;  Kinv[4] = (a * i - cc * g) * id;  Kinv[5] = (cc * d - a * f) * id;
    Kinv[6] = (d * hh - e * g) * id; Kinv[7] = (b * g - a * hh) * id;  Kinv[8] = (a * e - b * d) * id;
    for (int r = 0; r < 3; ++r)
      Kinv_t[r] = (Kinv[r * 3 + 0] * P[3] + Kinv[r * 3 + 1] * P[7]) + Kinv[r * 3 + 2] * P[11];
    if (c->rect_lut) lut.assign(c->rect_lut, c->rect_lut + (size_t)2 * W * H);
    if (c->rect_mask) mask.assign(c->rect_mask, c->rect_mask + (size_t)W * H);
  }
  // cam2World (CameraSystem.cpp:121-139).  The reference inverts the 4x4 [P; 0 0 0 z] per call;
  // the oracle (and the GPU) use the algebraically identical closed form
  //   p = z * K'^-1 [u v 1]^T - K'^-1 P[:,3]
  void cam2World(const double x[2], double invDepth, double p[3]) const {
    const double z = 1.0 / invDepth;
    for (int r = 0; r < 3; ++r) {
      const double ray = (Kinv[r * 3 + 0] * x[0] + Kinv[r * 3 + 1] * x[1]) + Kinv[r * 3 + 2];
      p[r] = z * ray - Kinv_t[r];
    }
  }
  // world2Cam (CameraSystem.cpp:142-148)
  void world2Cam(const double p[3], double x[2]) const {
    double hmg[3];
    for (int r = 0; r < 3; ++r)
      hmg[r] = ((P[r * 4 + 0] * p[0] + P[r * 4 + 1] * p[1]) + P[r * 4 + 2] * p[2]) + P[r * 4 + 3];
    x[0] = hmg[0] / hmg[2];
    x[1] = hmg[1] / hmg[2];
  }
};

// DepthPoint (DepthPoint.h:70-88, DepthPoint.cpp)
struct DP {
  size_t row = 0, col = 0;
  double x[2] = {0.5, 0.5};
  double invDepth = -1.0, scale2 = 0.0, nu = 0.0, variance = 0.0, residual = 0.0;
  size_t age = 0;
  double p_cam[3] = {0, 0, 0};
  uint32_t pose_idx = 0;
  DP() {}
  DP(size_t r, size_t c) : row(r), col(c) { x[0] = c + 0.5; x[1] = r + 0.5; }
  // DepthPoint::update_studentT, DepthPoint.cpp:167-188
  void update_studentT(double invD, double s2, double var, double nu_in) {
    if (invDepth > -1e-6) {
      double nu_update = std::min(nu_in, nu);
      double invDepth_update = (s2 * invDepth + scale2 * invD) / (scale2 + s2);
      double scale2_update = (nu_update + sq(invDepth - invD) / (scale2 + s2)) /
                             (nu_update + 1) * (scale2 * s2) / (scale2 + s2);
      invDepth = invDepth_update;
      scale2 = scale2_update;
      nu = nu_update + 1;
      variance = nu / (nu - 2) * scale2;
      age++;
    } else {
      invDepth = invD; scale2 = s2; variance = var; nu = nu_in;
    }
  }
  // DepthPoint::update (Gaussian, LSnorm == "l2"), DepthPoint.cpp:146-164 + boundVariance :140-144
  void update(double invD, double var) {
    if (invDepth > -1e-6) {
      double temp = invDepth;
      invDepth = (variance * invD + var * temp) / (variance + var);
      temp = variance;
      variance = (temp * var) / (temp + var);
    } else {
      invDepth = invD;
      variance = var;
    }
    if (variance < 1e-6) variance = 1e-6;
  }
  bool valid() const { return invDepth > -1e-6; }  // DepthPoint.cpp:215-218
  bool valid(double var_thr, double age_thr, double dmax, double dmin) const {  // :221-230
    return invDepth > -1e-6 && (double)age >= age_thr && variance <= var_thr && invDepth <= dmax &&
           invDepth >= dmin;
  }
  // DepthPoint::copy (DepthPoint.cpp:233-245): everything except row/col
  void copy_from(const DP& o) {
    invDepth = o.invDepth; variance = o.variance; scale2 = o.scale2; nu = o.nu;
    x[0] = o.x[0]; x[1] = o.x[1];
    p_cam[0] = o.p_cam[0]; p_cam[1] = o.p_cam[1]; p_cam[2] = o.p_cam[2];
    pose_idx = o.pose_idx; residual = o.residual; age = o.age;
  }
};

// SmartGrid<DepthPoint> (SmartGrid.h).  std::list is emulated by a vector + alive flags +
// creation order; the pointer grid by int indices.  A cell whose element was erased reads as
// empty (the oracle's definition for the dangling pointer of Appendix A-7).
struct Grid {
  int W = 0, H = 0;
  std::vector<DP> elems;
  std::vector<char> alive;
  std::vector<int> grid;  // -1 = NULL
  void init(int w, int h) { W = w; H = h; elems.clear(); alive.clear(); grid.assign((size_t)w * h, -1); }
  bool exists(size_t r, size_t c) const { return grid[r * W + c] >= 0; }  // SmartGrid.h:318-325
  DP& get(size_t r, size_t c) { return elems[grid[r * W + c]]; }
  void set(size_t r, size_t c, const DP& v) {  // SmartGrid.h:306-316
    if (grid[r * W + c] < 0) {
      elems.push_back(DP(r, c));
      alive.push_back(1);
      grid[r * W + c] = (int)elems.size() - 1;
    }
    elems[grid[r * W + c]].copy_from(v);
  }
  size_t size() const { size_t n = 0; for (char a : alive) n += a; return n; }
  // SmartGrid::clean, SmartGrid.h:222-243
  void clean(double var_thr, double age_thr, double dmax, double dmin) {
    for (size_t i = 0; i < elems.size(); ++i) {
      if (!alive[i]) continue;
      if (!elems[i].valid(var_thr, age_thr, dmax, dmin)) {
        grid[elems[i].row * W + elems[i].col] = -1;  // the cell the element BELIEVES it is in
        alive[i] = 0;
      }
    }
    // resolve dangling pointers: any grid entry that points to an erased element is empty
    for (auto& g : grid) if (g >= 0 && !alive[g]) g = -1;
  }
  // SmartGrid::getNeighbourhood, SmartGrid.h:367-386.  NOTE the loop bounds mix int and
  // size_t: `for (int r = row - radius; r <= row + radius; r++)` compares r converted to
  // size_t, so for row < radius (r negative) the row loop body never runs, and likewise the
  // column loop for col < radius.  Pixels within `radius` of the top/left border therefore get
  // NO neighbours.  Reproduced literally.
  void neighbourhood(size_t row, size_t col, size_t radius, std::vector<const DP*>& out) const {
    out.clear();
    for (int r = (int)(row - radius); (size_t)(int64_t)r <= row + radius; r++) {
      for (int c = (int)(col - radius); (size_t)(int64_t)c <= col + radius; c++) {
        if (r >= 0 && r < H && c >= 0 && c < W) {
          int g = grid[(size_t)r * W + c];
          if (g >= 0 && elems[g].valid()) out.push_back(&elems[g]);
        }
      }
    }
  }
};

struct Frame {
  std::vector<DP> pts;
  std::vector<Mat4> poses;  // T_world_cam table, indexed by DP::pose_idx
};

}  // namespace

struct orc_mapper {
  esvo_params_t prm;
  Camera camL, camR;
  double baseline = 0;
  int real_threads = 1;
  int bm_exact_int = 0;   // 1: ZNCC cost from exact integer moments (what the GPU computes)
  int lm_canonical = 0;   // 1: canonical reduction order for the LM sums (see reduce_patch)
  // observation
  uint64_t obs_t_ns = 0;
  std::vector<double> tsL, tsR;  // row-major W*H doubles 0..255 (TimeSurfaceObservation.h:68-69)
  Mat4 T_world_obs;
  // pose table (st_map_)
  std::vector<uint64_t> pose_t;
  std::vector<Mat4> pose_T;
  // window + map
  std::deque<Frame> window;  // dqvDepthPoints_
  Grid map;
  Mat4 T_world_frame;
  // counters
  uint64_t n_replace = 0, n_replace_displaced = 0, max_scale_iters = 0, n_evals = 0;

  int W() const { return camL.W; }
  int H() const { return camL.H; }
};

namespace {

// ---------------------------------------------------------------------------------------------
// EventBM (EventBM.cpp)
// ---------------------------------------------------------------------------------------------
// tools::meanStdDev + normalizePatch (utils.h:74-92) on a wy x wx patch stored row-major.
// Eigen's MatrixXd is column-major and .sum() is a (vectorised) reduction whose order is not
// specified; the oracle sums in Eigen's storage order (column-major, sequential).
void normalize_patch(const double* p, int wx, int wy, double* out) {
  const size_t n = (size_t)wx * wy;
  double sum = 0;
  for (int x = 0; x < wx; ++x) for (int y = 0; y < wy; ++y) sum += p[y * wx + x];
  const double mean = sum / n;
  double ss = 0;
  for (int x = 0; x < wx; ++x) for (int y = 0; y < wy; ++y) { double s = p[y * wx + x] - mean; ss += s * s; }
  const double sigma = std::sqrt(ss / n) + 1e-6;
  for (size_t i = 0; i < n; ++i) out[i] = (p[i] - mean) / sigma;
}
// EventBM::zncc_cost, EventBM.cpp:317-333 (normalized == false)
double zncc_cost(const double* l, const double* r, int wx, int wy, double* tmpl, double* tmpr) {
  normalize_patch(l, wx, wy, tmpl);
  normalize_patch(r, wx, wy, tmpr);
  double s = 0;
  for (int x = 0; x < wx; ++x) for (int y = 0; y < wy; ++y) s += tmpl[y * wx + x] * tmpr[y * wx + x];
  return 0.5 * (1 - s / ((double)wy * wx));
}
// The same quantity from exact integer moments (TS values are integers 0..255): used by the
// GPU kernel; exposed through orc_mapper_match_costs(exact_int=1) to bound the difference.
//   sum((l-ml)(r-mr)) = Slr - Sl*Sr/N ;  sigma = sqrt(Sxx/N - (Sx/N)^2) + 1e-6
double zncc_cost_int(int64_t Sl, int64_t Sll, int64_t Sr, int64_t Srr, int64_t Slr, int N) {
  const double n = (double)N;
  const double varl = (double)(N * Sll - Sl * Sl) / (n * n);
  const double varr = (double)(N * Srr - Sr * Sr) / (n * n);
  const double sigl = std::sqrt(varl) + 1e-6, sigr = std::sqrt(varr) + 1e-6;
  const double cov = (double)(N * Slr - Sl * Sr) / n;  // = sum((l-ml)(r-mr))
  return 0.5 * (1 - cov / (sigl * sigr) / n);
}


// ---------------------------------------------------------------------------------------------
// Patch reductions.  The reference sums residuals either with Eigen reductions (stableNorm,
// blueNorm, dot: vectorised, order unspecified) or with a plain (y,x) loop (t-scale update,
// DepthProblem.cpp:101-116).  mode 0 = sequential in index order (the literal loop order).
// mode 1 = "canonical": per-column partial sums over the rows (top to bottom), then a fixed
// pairwise tree over the columns padded with zeros to a power of two.  The GPU kernel reduces
// in exactly the canonical order, so oracle(mode 1) == GPU bit for bit; tests bound
// |mode 0 - mode 1|.
// ---------------------------------------------------------------------------------------------
double reduce_patch(const double* v, int wx, int wy, int mode) {
  if (mode == 0) {
    double s = 0;
    for (int i = 0; i < wx * wy; ++i) s += v[i];
    return s;
  }
  int P = 1;
  while (P < wx) P <<= 1;
  double col[64];
  for (int x = 0; x < P; ++x) {
    double a = 0.0;
    if (x < wx) { a = v[x]; for (int y = 1; y < wy; ++y) a = a + v[y * wx + x]; }
    col[x] = a;
  }
  for (int w = 1; w < P; w <<= 1)            // xor-butterfly: lane x gets col[x] + col[x ^ w]
    for (int x = 0; x < P; x += 2 * w)
      for (int k = 0; k < w; ++k) { double t = col[x + k] + col[x + k + w]; col[x + k] = t; col[x + k + w] = t; }
  return col[0];
}

struct BM {
  const orc_mapper* M;
  int wx, wy, W, H;
  // EventBM::isValidPatch, EventBM.cpp:251-267
  bool isValidPatch(int x, int y, int& ltx, int& lty) const {
    int hx = (wx - 1) / 2, hy = (wy - 1) / 2;
    ltx = x - hx; lty = y - hy;
    int rbx = x + hx, rby = y + hy;
    if (ltx < 1 || lty < 1 || rbx >= W - 1 || rby >= H - 1) return false;
    return true;
  }
  void block(const std::vector<double>& img, int ltx, int lty, double* out) const {
    for (int y = 0; y < wy; ++y)
      for (int x = 0; x < wx; ++x) out[y * wx + x] = img[(size_t)(lty + y) * W + (ltx + x)];
  }
  // EventBM::epipolarSearching, EventBM.cpp:170-226
  bool epipolarSearching(double& min_cost, int& bestX, int& bestY, size_t& bestDisp, size_t start,
                         size_t end, size_t step, int x1x, int x1y, const double* patch_src,
                         double* all_costs /*nullable*/, int exact_int) const {
    bool found = false;
    std::map<size_t, double> mDispCost;
    std::vector<double> patch_dst((size_t)wx * wy), t1((size_t)wx * wy), t2((size_t)wx * wy);
    const double ZNCC_MAX = 1.0, thr = M->prm.bm_zncc_threshold;
    for (size_t disp = start; disp <= end; disp += step) {
      int x2x, x2y;  // :180-183
      if (!M->prm.bm_updown) { x2x = (int)((size_t)x1x - disp); x2y = x1y; }
      else { x2x = x1x; x2y = (int)((size_t)x1y - disp); }
      int ltx, lty;
      if (!isValidPatch(x2x, x2y, ltx, lty)) {
        mDispCost.emplace(disp, ZNCC_MAX);
        if (all_costs) all_costs[disp - start] = ZNCC_MAX;
        continue;
      }
      block(M->tsR, ltx, lty, patch_dst.data());
      double cost;
      if (!exact_int) {
        cost = zncc_cost(patch_src, patch_dst.data(), wx, wy, t1.data(), t2.data());
      } else {
        int64_t Sl = 0, Sll = 0, Sr = 0, Srr = 0, Slr = 0;
        for (int i = 0; i < wx * wy; ++i) {
          int64_t l = (int64_t)patch_src[i], r = (int64_t)patch_dst[i];
          Sl += l; Sll += l * l; Sr += r; Srr += r * r; Slr += l * r;
        }
        cost = zncc_cost_int(Sl, Sll, Sr, Srr, Slr, wx * wy);
      }
      mDispCost.emplace(disp, cost);
      if (all_costs) all_costs[disp - start] = cost;
      if (cost <= min_cost) { min_cost = cost; bestX = x2x; bestY = x2y; bestDisp = disp; }
    }
    if (step > 1) {  // coarse
      if (mDispCost.find(bestDisp - step) != mDispCost.end() &&
          mDispCost.find(bestDisp + step) != mDispCost.end()) {
        if (mDispCost[bestDisp - step] < ZNCC_MAX && mDispCost[bestDisp + step] < ZNCC_MAX)
          if (min_cost < thr) found = true;
      }
    } else {
      if (min_cost < thr) found = true;
    }
    return found;
  }
  // EventBM::match_an_event, EventBM.cpp:80-168
  bool match_an_event(const esvo_event_t& e, uint32_t event_idx, esvo_match_t& out,
                      double* all_costs = nullptr, int exact_int = 0) const {
    const Camera& cam = M->camL;
    if (e.x >= W || e.y >= H) return false;  // (the reference would index out of bounds)
    const size_t li = ((size_t)e.y * W + e.x) * 2;
    const double xr[2] = {(double)cam.lut[li], (double)cam.lut[li + 1]};  // :88
    if (xr[0] < 0 || xr[0] > (double)(W - 1) || xr[1] < 0 || xr[1] > (double)(H - 1)) return false;  // :90-92
    if (!cam.mask.empty() && cam.mask[(size_t)((long)xr[1]) * W + (size_t)((long)xr[0])] <= 125) return false;  // :94 (truncation)
    const int x1x = (int)std::floor(xr[0]), x1y = (int)std::floor(xr[1]);  // :96
    int ltx, lty;
    if (!isValidPatch(x1x, x1y, ltx, lty)) return false;  // :98
    std::vector<double> patch_src((size_t)wx * wy);
    block(M->tsL, ltx, lty, patch_src.data());  // :101
    size_t cnt = 0;
    for (double v : patch_src) cnt += (v < 1);
    if ((double)cnt > 0.95 * (double)patch_src.size()) return false;  // :104-109
    double min_cost = 1.0;
    int bx = 0, by = 0;
    size_t bestDisp = 0;
    bool any = false;
    const size_t lo = (size_t)M->prm.bm_min_disparity, hi = (size_t)M->prm.bm_max_disparity;
    const size_t step = (size_t)M->prm.bm_step;
    // coarse :119-126
    {
      // (bestDisp is uninitialised in the reference if no candidate is valid; then min_cost
      //  stays 1.0 and the search fails for step==1.  For step>1 the oracle also fails.)
      size_t bd = (size_t)-1;
      double mc = min_cost;
      if (!epipolarSearching(mc, bx, by, bd, lo, hi, step, x1x, x1y, patch_src.data(), all_costs, exact_int))
        return false;
      min_cost = mc; bestDisp = bd; any = true;
    }
    (void)any;
    // fine :128-138   (size_t arithmetic; `>= 0` is always true, Appendix A-9)
    size_t fine_start = bestDisp - (step - 1);
    if (!epipolarSearching(min_cost, bx, by, bestDisp, fine_start, bestDisp + (step - 1), 1, x1x, x1y,
                           patch_src.data(), nullptr, exact_int))
      return false;
    if (min_cost <= M->prm.bm_zncc_threshold) {  // :141
      const double disparity = M->prm.bm_updown ? (double)(x1y - by) : (double)(x1x - bx);  // :146-151
      const double depth = M->baseline * cam.P[0] / disparity;  // :152
      // StampTransformationMap_lower_bound (utils.h:66-71): first stamp with toSec >= event toSec
      const double te = ev_sec(e);
      size_t k = 0;
      {
        size_t lo_i = 0, hi_i = M->pose_t.size();
        while (lo_i < hi_i) {
          size_t mid = (lo_i + hi_i) / 2;
          if (ns_to_sec(M->pose_t[mid]) < te) lo_i = mid + 1; else hi_i = mid;
        }
        k = lo_i;
      }
      if (k == M->pose_t.size()) return false;  // :155-156
      out.x_left[0] = xr[0]; out.x_left[1] = xr[1];
      out.inv_depth = 1.0 / depth;  // :158
      out.cost = min_cost;
      out.disp = disparity;
      out.event_idx = event_idx;
      out.pose_idx = (uint32_t)k;
      return true;
    }
    return false;
  }
};

// ---------------------------------------------------------------------------------------------
// DepthProblem (DepthProblem.cpp) — the residual functor
// ---------------------------------------------------------------------------------------------
struct DepthProblem {
  orc_mapper* M;
  int wx, wy;
  double coor[2];
  double T_left_virtual[12];  // 3x4

  // DepthProblem::setProblem, DepthProblem.cpp:17-32
  void setProblem(const double c[2], const Mat4& T_world_virtual) {
    coor[0] = c[0]; coor[1] = c[1];
    Mat4 T_left_world = rigid_inverse(M->T_world_obs);
    Mat4 T = mat4_mul(T_left_world, T_world_virtual);
    std::memcpy(T_left_virtual, T.m, sizeof(T_left_virtual));
  }
  // DepthProblem::warping, DepthProblem.cpp:162-191
  bool warping(double d, double x1[2], double x2[2]) const {
    double p_rv[3];
    M->camL.cam2World(coor, d, p_rv);
    double pl[3];
    for (int r = 0; r < 3; ++r)
      pl[r] = ((T_left_virtual[r * 4 + 0] * p_rv[0] + T_left_virtual[r * 4 + 1] * p_rv[1]) +
               T_left_virtual[r * 4 + 2] * p_rv[2]) + T_left_virtual[r * 4 + 3];
    M->camL.world2Cam(pl, x1);
    M->camR.world2Cam(pl, x2);
    const int W = M->W(), H = M->H();
    const int hx = (wx - 1) / 2, hy = (wy - 1) / 2;
    // oracle definition for non-finite coordinates (UB in the reference): warping fails
    if (!std::isfinite(x1[0]) || !std::isfinite(x1[1]) || !std::isfinite(x2[0]) || !std::isfinite(x2[1]))
      return false;
    if (x1[0] < hx || x1[0] > W - hx || x1[1] < hy || x1[1] > H - hy) return false;
    if (x2[0] < hx || x2[0] > W - hx || x2[1] < hy || x2[1] > H - hy) return false;
    return true;
  }
  // DepthProblem::patchInterpolation, DepthProblem.cpp:193-262
  bool patchInterpolation(const std::vector<double>& img, const double loc[2], double* patch) const {
    const int W = M->W(), H = M->H();
    const int ulx = (int)std::floor(loc[0]) - (wx - 1) / 2, uly = (int)std::floor(loc[1]) - (wy - 1) / 2;
    const int drx = (int)std::floor(loc[0]) + (wx - 1) / 2, dry = (int)std::floor(loc[1]) + (wy - 1) / 2;
    if (ulx < 0 || uly < 0) return false;
    if (drx >= W || dry >= H) return false;
    const double di0 = loc[1], di1 = loc[0];
    const int l0 = (int)std::floor(di0), l1 = (int)std::floor(di1);
    const int u0 = l0 + 1, u1 = l1 + 1;
    const double q1 = u1 - di1, q2 = di1 - l1, q3 = u0 - di0, q4 = di0 - l0;
    if (uly + wy >= H || ulx + wx >= W) return false;
    // R = q1*Src[:, 0:wx] + q2*Src[:, 1:wx+1]   ((wy+1) x wx);  F = q3*R[0:wy] + q4*R[1:wy+1]
    std::vector<double> R((size_t)(wy + 1) * wx);
    for (int y = 0; y <= wy; ++y)
      for (int x = 0; x < wx; ++x)
        R[y * wx + x] = q1 * img[(size_t)(uly + y) * W + ulx + x] + q2 * img[(size_t)(uly + y) * W + ulx + x + 1];
    for (int y = 0; y < wy; ++y)
      for (int x = 0; x < wx; ++x) patch[y * wx + x] = q3 * R[y * wx + x] + q4 * R[(y + 1) * wx + x];
    return true;
  }
  // DepthProblem::operator(), DepthProblem.cpp:34-160, LSnorm == "Tdist"
  int operator()(double x, double* fvec) const {
    M->n_evals++;
    const int N = wx * wy;
    const double nu = M->prm.td_nu, scale = M->prm.td_scale;
    auto fail_fill = [&]() {
      for (int i = 0; i < N; ++i) {
        double residual = 255;
        double weight = (nu + 1) / (nu + sq(residual / scale));
        fvec[i] = std::sqrt(weight) * residual;
      }
    };
    double x1[2], x2[2];
    if (M->prm.ls_norm == ESVO_LSNORM_L2) {  // :43-45, 67-75, 143-147: the plain temporal residual, 255 on failure
      std::vector<double> a1(N), a2(N);
      if (warping(x, x1, x2) && patchInterpolation(M->tsL, x1, a1.data()) && patchInterpolation(M->tsR, x2, a2.data())) {
        for (int i = 0; i < N; ++i) fvec[i] = a1[i] - a2[i];
        return 1;
      }
      for (int i = 0; i < N; ++i) fvec[i] = 255;
      return 0;
    }
    if (!warping(x, x1, x2)) { fail_fill(); return 0; }
    std::vector<double> tau1(N), tau2(N);
    if (patchInterpolation(M->tsL, x1, tau1.data()) && patchInterpolation(M->tsR, x2, tau2.data())) {
      std::vector<double> vR(N), vR2(N), terms(N);
      const double scale2_0 = sq(scale);  // td_scaleSquared_
      double s1 = scale2_0, s2 = -1.0;
      bool first = true;
      uint64_t iters = 0;
      while (std::fabs(s2 - s1) / s1 > 0.05 || first) {  // :96
        if (!first) s1 = s2;
        double sum = 0;
        if (!M->lm_canonical) {
          for (int i = 0; i < N; ++i) {  // y-major then x == index order
            if (first) { vR[i] = tau1[i] - tau2[i]; vR2[i] = sq(vR[i]); }
            if (vR[i] != 0) sum += vR2[i] * (nu + 1) / (nu + vR2[i] / s1);
          }
        } else {
          for (int i = 0; i < N; ++i) {
            if (first) { vR[i] = tau1[i] - tau2[i]; vR2[i] = sq(vR[i]); }
            terms[i] = (vR[i] != 0) ? vR2[i] * (nu + 1) / (nu + vR2[i] / s1) : 0.0;
          }
          sum = reduce_patch(terms.data(), wx, wy, 1);
        }
        if (sum == 0) { s2 = scale2_0; break; }
        s2 = sum / N;
        first = false;
        ++iters;
      }
      if (iters > M->max_scale_iters) M->max_scale_iters = iters;
      for (int i = 0; i < N; ++i) {
        double weight = (nu + 1) / (nu + vR2[i] / s2);
        fvec[i] = std::sqrt(weight) * vR[i];
      }
      return 1;
    }
    fail_fill();
    return 0;
  }
};

// ---------------------------------------------------------------------------------------------
// Eigen::LevenbergMarquardt<NumericalDiff<DepthProblem>> for n = 1
// ---------------------------------------------------------------------------------------------
struct LM1 {
  const DepthProblem& F;
  int m;
  // parameters (resetParameters() + the reference's overrides, DepthProblemSolver.cpp:147-150)
  double factor = 100., ftol, xtol, gtol = 0., epsfcn = 0.;
  int maxfev;
  // state
  std::vector<double> fvec, fjac, wa4, val2;
  double diag = 0, qtf = 0, r = 0;  // r = R(0,0) of the QR of the m x 1 Jacobian (|r| = ||J||)
  double par = 0, fnorm = 0, gnorm = 0, xnorm = 0, delta = 0;
  int nfev = 0, iter = 0;
  LM1(const DepthProblem& f, int m_, double ftol_, double xtol_, int maxfev_)
      : F(f), m(m_), ftol(ftol_), xtol(xtol_), maxfev(maxfev_), fvec(m_), fjac(m_), wa4(m_), val2(m_) {}

  int wx = 0, wy = 0, mode = 0;  // reduction geometry/mode (see reduce_patch)
  double dotp(const double* a, const double* b) const {
    std::vector<double> t(m);
    for (int i = 0; i < m; ++i) t[i] = a[i] * b[i];
    return reduce_patch(t.data(), wx, wy, mode);
  }
  double norm(const double* v, int) const { return std::sqrt(dotp(v, v)); }

  int minimizeInit(double& x) {
    nfev = 1;
    F(x, fvec.data());
    fnorm = norm(fvec.data(), m);
    par = 0.; iter = 1;
    return -2;  // NotStarted
  }
  // internal::lmpar2 for n == 1
  void lmpar2(double& x_out) {
    const double dwarf = std::numeric_limits<double>::min();
    double x = qtf / r;  // Gauss-Newton direction (rank == 1; r != 0 guaranteed by the caller)
    double wa2 = diag * x;
    double dxnorm = std::fabs(wa2);
    double fp = dxnorm - delta;
    if (fp <= 0.1 * delta) { par = 0; x_out = x; return; }
    double wa1 = diag * wa2 / dxnorm;
    wa1 = wa1 / r;
    double temp = std::fabs(wa1);
    double parl = fp / delta / temp / temp;
    wa1 = r * qtf / diag;
    double gn = std::fabs(wa1);
    double paru = gn / delta;
    if (paru == 0.) paru = dwarf / std::min(delta, 0.1);
    par = std::max(par, parl);
    par = std::min(par, paru);
    if (par == 0.) par = gn / dxnorm;
    int it = 0;
    while (true) {
      ++it;
      if (par == 0.) par = std::max(dwarf, 0.001 * paru);
      const double ds = std::sqrt(par) * diag;
      // qrsolv, n == 1: one Givens rotation eliminates ds against r
      const double sdiag2 = r * r + ds * ds;
      const double sdiag = std::sqrt(sdiag2);
      x = r * qtf / sdiag2;
      wa2 = diag * x;
      dxnorm = std::fabs(wa2);
      temp = fp;
      fp = dxnorm - delta;
      if (std::fabs(fp) <= 0.1 * delta || (parl == 0. && fp <= temp && temp < 0.) || it == 10) break;
      wa1 = diag * (wa2 / dxnorm);
      wa1 = wa1 / sdiag;
      temp = std::fabs(wa1);
      const double parc = fp / delta / temp / temp;
      if (fp > 0.) parl = std::max(parl, par);
      if (fp < 0.) paru = std::min(paru, par);
      par = std::max(parl, par + parc);
    }
    x_out = x;
  }
  int minimizeOneStep(double& x) {
    // NumericalDiff<Forward>::df: 2 evaluations
    {
      const double eps = std::sqrt(std::max(epsfcn, std::numeric_limits<double>::epsilon()));
      F(x, wa4.data());  // val1 (== fvec numerically; re-evaluated as Eigen does)
      double h = eps * std::fabs(x);
      if (h == 0.) h = eps;
      F(x + h, val2.data());
      for (int i = 0; i < m; ++i) fjac[i] = (val2[i] - wa4[i]) / h;
      nfev += 2;
    }
    const double wa2n = norm(fjac.data(), m);  // column norm
    const double jtf = dotp(fjac.data(), fvec.data());
    r = wa2n;  // sign convention: +||J|| (results are sign-invariant)
    qtf = (r != 0.) ? jtf / r : fvec[0];
    if (iter == 1) {
      diag = (wa2n == 0.) ? 1. : wa2n;
      xnorm = std::fabs(diag * x);
      delta = factor * xnorm;
      if (delta == 0.) delta = factor;
    }
    gnorm = 0.;
    if (fnorm != 0.)
      if (wa2n != 0.) gnorm = std::max(gnorm, std::fabs(r * (qtf / fnorm) / wa2n));
    if (gnorm <= gtol) return 4;  // CosinusTooSmall
    diag = std::max(diag, wa2n);
    double ratio;
    do {
      double p;
      lmpar2(p);
      const double wa1 = -p;
      const double xnew = x + wa1;
      const double pnorm = std::fabs(diag * wa1);
      if (iter == 1) delta = std::min(delta, pnorm);
      F(xnew, wa4.data());
      ++nfev;
      const double fnorm1 = norm(wa4.data(), m);
      double actred = -1.;
      if (0.1 * fnorm1 < fnorm) actred = 1. - (fnorm1 / fnorm) * (fnorm1 / fnorm);
      const double wa3 = r * wa1;
      const double t1 = std::fabs(wa3) / fnorm, temp1 = t1 * t1;
      const double t2 = std::sqrt(par) * pnorm / fnorm, temp2 = t2 * t2;
      const double prered = temp1 + temp2 / 0.5;
      const double dirder = -(temp1 + temp2);
      ratio = 0.;
      if (prered != 0.) ratio = actred / prered;
      if (ratio <= 0.25) {
        double temp = 0.5;
        if (actred >= 0.) temp = 0.5;
        if (actred < 0.) temp = 0.5 * dirder / (dirder + 0.5 * actred);
        if (0.1 * fnorm1 >= fnorm || temp < 0.1) temp = 0.1;
        delta = temp * std::min(delta, pnorm / 0.1);
        par /= temp;
      } else if (!(par != 0. && ratio < 0.75)) {
        delta = pnorm / 0.5;
        par = 0.5 * par;
      }
      if (ratio >= 1e-4) {
        x = xnew;
        fvec.swap(wa4);
        xnorm = std::fabs(diag * x);
        fnorm = fnorm1;
        ++iter;
      }
      const double eps = std::numeric_limits<double>::epsilon();
      if (std::fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1. && delta <= xtol * xnorm) return 3;
      if (std::fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1.) return 1;
      if (delta <= xtol * xnorm) return 2;
      if (nfev >= maxfev) return 5;
      if (std::fabs(actred) <= eps && prered <= eps && 0.5 * ratio <= 1.) return 6;
      if (delta <= eps * xnorm) return 7;
      if (gnorm <= eps) return 8;
    } while (ratio < 1e-4);
    return -1;  // Running
  }
};

// DepthProblemSolver::solve_single_problem_numerical, DepthProblemSolver.cpp:138-214
bool solve_single(orc_mapper* M, const DepthProblem& prob, double d_init, double result[3], double info[4]) {
  const int N = M->prm.patch_size_x * M->prm.patch_size_y;
  LM1 lm(prob, N, 1e-6, 1e-6, M->prm.lm_max_iteration * 3);
  lm.wx = M->prm.patch_size_x; lm.wy = M->prm.patch_size_y; lm.mode = M->lm_canonical;
  double x = d_init;
  lm.minimizeInit(x);
  size_t iteration = 0;
  int optimizationState = 0, status = -2;
  while (true) {
    status = lm.minimizeOneStep(x);
    iteration++;
    if (iteration >= (size_t)M->prm.lm_max_iteration) break;
    bool terminate = false;
    if (status == 2 || status == 3) {
      if (optimizationState == 0) optimizationState++;
      else terminate = true;
    }
    if (terminate) break;
  }
  if (info) { info[0] = (double)iteration; info[1] = lm.nfev; info[2] = status; info[3] = 0; }
  if (x <= 0.001) return false;  // :192
  result[0] = x;
  // internal::covar for n == 1 -> 1/r^2 (0 if r == 0); Tdist: variance = td_stdvar^2 * that
  const double nu = M->prm.td_nu;
  const double td_stdvar = std::sqrt(nu / (nu - 2) * sq(M->prm.td_scale));  // DepthProblem.h:34
  const double invJtJ = (lm.r != 0.) ? (1. / lm.r) * (1. / lm.r) : 0.;
  if (M->prm.ls_norm == ESVO_LSNORM_L2) {  // :200-206: cov = |f|^2 / (values - inputs) * (J^T J)^-1
    const double fnorm = lm.fnorm;       // lm.fvec.blueNorm(): the norm of the final residual vector
    const double covfac = fnorm * fnorm / (double)(lm.m - 1);
    result[1] = covfac * invJtJ;
  } else {
    result[1] = sq(td_stdvar) * invJtJ;  // :210
  }
  result[2] = lm.fnorm * lm.fnorm;              // :212
  if (info) info[3] = 1;
  return true;
}

// stride-N thread emulation: indices i_thread, i_thread+N, ... per thread, threads concatenated
std::vector<size_t> stride_order(size_t n, int T) {
  std::vector<size_t> o;
  o.reserve(n);
  for (int t = 0; t < T; ++t) for (size_t i = t; i < n; i += T) o.push_back(i);
  return o;
}

// ---------------------------------------------------------------------------------------------
// DepthFusion (DepthFusion.cpp)
// ---------------------------------------------------------------------------------------------
inline bool boundaryCheck(double x, double y, size_t w, size_t h) {  // :194-205
  return !(x < 0 || x >= (double)w || y < 0 || y >= (double)h);
}
inline bool studentTCompatibleTest(double d1, double d2, double v1, double v2) {  // :220-231
  double s1 = std::sqrt(v1), s2 = std::sqrt(v2), diff = std::fabs(d1 - d2);
  return diff < 2 * s1 || diff < 2 * s2;
}

inline bool chiSquareTest(double d1, double d2, double v1, double v2) {  // :207-218
  const double dd = sq(d1 - d2);
  return dd / v1 + dd / v2 < 5.99;
}

// DepthFusion::propagate_one_point, DepthFusion.cpp:18-68
bool propagate_one_point(const orc_mapper* M, const DP& prior, DP& prop, const Mat4& T) {
  double pp[3];
  for (int r = 0; r < 3; ++r)
    pp[r] = ((T.m[r * 4 + 0] * prior.p_cam[0] + T.m[r * 4 + 1] * prior.p_cam[1]) + T.m[r * 4 + 2] * prior.p_cam[2]) + T.m[r * 4 + 3];
  double xp[2];
  M->camL.world2Cam(pp, xp);
  if (!boundaryCheck(xp[0], xp[1], M->W(), M->H())) return false;
  if (!(xp[0] == xp[0]) || !(xp[1] == xp[1])) return false;  // NaN passes the reference's test (UB after); defined: rejected
  prop = DP((size_t)std::floor(xp[1]), (size_t)std::floor(xp[0]));
  prop.x[0] = xp[0]; prop.x[1] = xp[1];
  const double invDepth = 1.0 / pp[2];
  double denominator = (T.m[8] * prior.p_cam[0] + T.m[9] * prior.p_cam[1]) + T.m[11];
  denominator /= prior.p_cam[2];
  denominator += T.m[10];
  const double J = T.m[10] / sq(denominator);
  if (M->prm.ls_norm == ESVO_LSNORM_L2) {  // :49-53
    prop.update(invDepth, J * J * prior.variance);
  } else {
    const double scale2 = J * J * prior.scale2;
    const double nu = prior.nu;
    const double variance = nu / (nu - 2) * scale2;
    prop.update_studentT(invDepth, scale2, variance, nu);
  }
  prop.p_cam[0] = pp[0]; prop.p_cam[1] = pp[1]; prop.p_cam[2] = pp[2];
  prop.residual = prior.residual;
  prop.age = prior.age;
  return true;
}

// DepthFusion::fusion, DepthFusion.cpp:90-192 (Tdist)
int fusion(orc_mapper* M, const DP& prop, int fusion_radius) {
  int numFusion = 0;
  Grid& dm = M->map;
  const bool l2 = M->prm.ls_norm == ESVO_LSNORM_L2;
  std::vector<std::pair<size_t, size_t>> coords;
  if (fusion_radius == 0) {
    for (int dy = 0; dy <= 1; dy++) for (int dx = 0; dx <= 1; dx++) coords.emplace_back(prop.row + dy, prop.col + dx);
  } else {
    for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) coords.emplace_back(prop.row + dy, prop.col + dx);
  }
  for (auto& rc : coords) {
    const size_t row = rc.first, col = rc.second;
    if (!boundaryCheck((double)col, (double)row, M->W(), M->H())) continue;
    if (!dm.exists(row, col)) {  // case 1
      DP dp_new(row, col);
      if (l2) dp_new.update(prop.invDepth, prop.variance);  // :130-131
      else dp_new.update_studentT(prop.invDepth, prop.scale2, prop.variance, prop.nu);
      dp_new.residual = prop.residual;
      dp_new.age = prop.age;
      M->camL.cam2World(dp_new.x, prop.invDepth, dp_new.p_cam);
      dm.set(row, col, dp_new);
    } else {
      DP& c = dm.get(row, col);
      const bool compatible = l2 ? chiSquareTest(prop.invDepth, c.invDepth, prop.variance, c.variance)   // :150-152
                                 : studentTCompatibleTest(prop.invDepth, c.invDepth, prop.variance, c.variance);
      if (compatible) {  // 2.1
        if (l2) c.update(prop.invDepth, prop.variance);  // :164-165
        else c.update_studentT(prop.invDepth, prop.scale2, prop.variance, prop.nu);
        c.age++;
        c.residual = std::min(c.residual, prop.residual);
        M->camL.cam2World(c.x, prop.invDepth, c.p_cam);
        numFusion++;
      } else {  // 2.2
        if (c.invDepth - 2 * std::sqrt(c.variance) > prop.invDepth) continue;
        if (prop.variance < c.variance && prop.residual < c.residual) {
          M->n_replace++;
          if (prop.row != row || prop.col != col) M->n_replace_displaced++;
          c = prop;  // operator=: row/col/x of the propagated point travel too (Appendix A-7)
        }
      }
    }
  }
  return numFusion;
}

// DepthFusion::update, DepthFusion.cpp:71-87
int fusion_update(orc_mapper* M, const Frame& fr, int fusion_radius) {
  int numFusion = 0;
  Mat4 T_frame_world = rigid_inverse(M->T_world_frame);
  for (size_t i = 0; i < fr.pts.size(); ++i) {
    Mat4 T_frame_obs = mat4_mul(T_frame_world, fr.poses[fr.pts[i].pose_idx]);
    DP prop;
    if (!propagate_one_point(M, fr.pts[i], prop, T_frame_obs)) continue;
    numFusion += fusion(M, prop, fusion_radius);
  }
  return numFusion;
}

// DepthRegularization::apply, DepthRegularization.cpp:19-110 (Tdist)
void regularize(orc_mapper* M) {
  Grid& dm = M->map;
  Grid tmp;
  tmp.init(dm.W, dm.H);
  const size_t radius = (size_t)M->prm.reg_radius;
  const size_t minN = (size_t)M->prm.reg_min_neighbours, minClose = (size_t)M->prm.reg_min_close_neighbours;
  std::vector<const DP*> nb, close;
  for (size_t e = 0; e < dm.elems.size(); ++e) {
    if (!dm.alive[e]) continue;
    const DP& it = dm.elems[e];
    tmp.set(it.row, it.col, it);
    DP& newDp = tmp.get(it.row, it.col);
    if (it.valid()) {
      dm.neighbourhood(it.row, it.col, radius, nb);
      bool isSet = false;
      if (nb.size() > minN) {
        close.clear();
        for (const DP* n : nb)
          if (n->valid()) {
            double diff = std::fabs(it.invDepth - n->invDepth);
            if (diff < 2.0 * std::sqrt(it.variance) || diff < 2.0 * std::sqrt(n->variance)) close.push_back(n);
          }
        if (close.size() > minClose && M->prm.ls_norm == ESVO_LSNORM_L2) {  // :56-65: inverse-variance weighted mean
          double totalInvVariances = 0.0, statisticalMean = 0.0;
          for (size_t i = 0; i < close.size(); ++i) totalInvVariances += 1.0 / close[i]->variance;
          for (size_t i = 0; i < close.size(); ++i)
            statisticalMean += close[i]->invDepth * (1.0 / close[i]->variance) / totalInvVariances;
          newDp.invDepth = statisticalMean;
          isSet = true;
        } else if (close.size() > minClose) {
          double nu_post = close[0]->nu, invDepth_post = close[0]->invDepth, scale2_post = close[0]->scale2;
          for (size_t i = 1; i < close.size(); ++i) {
            double nu_prior = nu_post, invDepth_prior = invDepth_post, scale2_prior = scale2_post;
            double nu_obs = close[i]->nu, invDepth_obs = close[i]->invDepth, scale2_obs = close[i]->scale2;
            nu_post = std::min(nu_prior, nu_obs);
            invDepth_post = (scale2_obs * invDepth_prior + scale2_prior * invDepth_obs) / (scale2_obs + scale2_prior);
            scale2_post = (nu_post + sq(invDepth_prior - invDepth_obs) / (scale2_prior + scale2_obs)) /
                          (nu_post + 1) * (scale2_prior * scale2_obs) / (scale2_prior + scale2_obs);
          }
          newDp.invDepth = invDepth_post;
          isSet = true;
        }
      }
      if (!isSet) newDp.invDepth = -1.0;
    }
  }
  dm = tmp;  // SmartGrid::operator= : elements copied, grid rebuilt from the elements' row/col
}

void set_ts(std::vector<double>& dst, const uint8_t* src, int W, int H, bool smooth) {
  std::vector<uint8_t> tmp;
  if (smooth) {  // TimeSurfaceObservation::GaussianBlurTS(5), TimeSurfaceObservation.h:107-116
    tmp.resize((size_t)W * H);
    gaussian5_u8(src, tmp.data(), W, H);
    src = tmp.data();
  }
  dst.resize((size_t)W * H);
  for (size_t i = 0; i < dst.size(); ++i) dst[i] = (double)src[i];
}

inline void dp_to_pod(const DP& d, esvo_depth_point_t& o, uint32_t seq) {
  o.row = (uint32_t)d.row; o.col = (uint32_t)d.col;
  o.x[0] = d.x[0]; o.x[1] = d.x[1];
  o.inv_depth = d.invDepth; o.scale2 = d.scale2; o.nu = d.nu; o.variance = d.variance; o.residual = d.residual;
  o.age = d.age;
  o.p_cam[0] = d.p_cam[0]; o.p_cam[1] = d.p_cam[1]; o.p_cam[2] = d.p_cam[2];
  o.pose_idx = d.pose_idx; o.seq = seq;
}
inline DP pod_to_dp(const esvo_depth_point_t& o) {
  DP d((size_t)o.row, (size_t)o.col);
  d.x[0] = o.x[0]; d.x[1] = o.x[1];
  d.invDepth = o.inv_depth; d.scale2 = o.scale2; d.nu = o.nu; d.variance = o.variance; d.residual = o.residual;
  d.age = (size_t)o.age;
  d.p_cam[0] = o.p_cam[0]; d.p_cam[1] = o.p_cam[1]; d.p_cam[2] = o.p_cam[2];
  d.pose_idx = o.pose_idx;
  return d;
}

}  // namespace

// =============================================================================================
// C API
// =============================================================================================
extern "C" orc_mapper_handle orc_mapper_create(const esvo_params_t* p, const esvo_calib_t* left,
                                               const esvo_calib_t* right) {
  orc_mapper* M = new orc_mapper();
  M->prm = *p;
  M->camL.init(left);
  M->camR.init(right);
  // CameraSystem::computeBaseline, CameraSystem.cpp:161-166: || P_r[:, :3]^-1 P_r[:, 3] ||
  M->baseline = std::sqrt((M->camR.Kinv_t[0] * M->camR.Kinv_t[0] + M->camR.Kinv_t[1] * M->camR.Kinv_t[1]) +
                          M->camR.Kinv_t[2] * M->camR.Kinv_t[2]);
  M->map.init(M->camL.W, M->camL.H);
  for (int i = 0; i < 16; ++i) M->T_world_obs.m[i] = M->T_world_frame.m[i] = (i % 5 == 0) ? 1.0 : 0.0;
  return M;
}
extern "C" void orc_mapper_destroy(orc_mapper_handle h) { delete h; }
extern "C" void orc_mapper_reset(orc_mapper_handle h) {
  h->window.clear();
  h->map.init(h->W(), h->H());
  h->n_replace = h->n_replace_displaced = h->max_scale_iters = h->n_evals = 0;
}
extern "C" void orc_mapper_set_params(orc_mapper_handle h, const esvo_params_t* p) { h->prm = *p; }
extern "C" void orc_mapper_set_threads(orc_mapper_handle h, int t) { h->real_threads = t < 1 ? 1 : t; }
extern "C" double orc_mapper_baseline(orc_mapper_handle h) { return h->baseline; }
extern "C" void orc_mapper_set_mode(orc_mapper_handle h, int bm_exact_int, int lm_canonical) {
  h->bm_exact_int = bm_exact_int; h->lm_canonical = lm_canonical;
}

extern "C" void orc_mapper_set_observation(orc_mapper_handle h, uint64_t t_ns, const uint8_t* l,
                                           const uint8_t* r, const double T[16]) {
  h->obs_t_ns = t_ns;
  // createMatchProblem applies GaussianBlurTS when SmoothTimeSurface (EventBM.cpp:68-72); the
  // blurred matrices are then also what DepthProblem reads (TS_left_/TS_right_ are replaced).
  set_ts(h->tsL, l, h->W(), h->H(), h->prm.smooth_time_surface != 0);
  set_ts(h->tsR, r, h->W(), h->H(), h->prm.smooth_time_surface != 0);
  std::memcpy(h->T_world_obs.m, T, sizeof(double) * 16);
}
extern "C" void orc_mapper_set_poses(orc_mapper_handle h, const uint64_t* t, const double* T, size_t m) {
  h->pose_t.assign(t, t + m);
  h->pose_T.resize(m);
  for (size_t i = 0; i < m; ++i) std::memcpy(h->pose_T[i].m, T + 16 * i, sizeof(double) * 16);
}

extern "C" size_t orc_select_events(const esvo_event_t* ev, size_t n, uint64_t t_ns, double half_slice,
                                    size_t max_num, uint32_t* out_idx, size_t cap) {
  // esvo_Mapping::dataTransferring, esvo_Mapping.cpp:562-574 (Appendix A-3)
  const double t_end = ns_to_sec(t_ns);
  const uint64_t t_begin_ns = ros_time_from_sec(std::max(0.0, t_end - 10 * half_slice));
  const double t_begin = ns_to_sec(t_begin_ns);
  auto lb = [&](double t) {  // tools::EventBuffer_lower_bound, utils.h:51-56
    size_t lo = 0, hi = n;
    while (lo < hi) { size_t mid = (lo + hi) / 2; if (ev_sec(ev[mid]) < t) lo = mid + 1; else hi = mid; }
    return lo;
  };
  size_t it_end = lb(t_end), it_begin = lb(t_begin);
  size_t cnt = 0;
  while (it_end != it_begin && cnt < max_num) {
    // the reference dereferences end() here when it_end == n (UB); the oracle skips that slot
    if (it_end < n) { if (cnt < cap) out_idx[cnt] = (uint32_t)it_end; cnt++; }
    it_end--;
  }
  return cnt;
}

extern "C" size_t orc_denoise_events(const esvo_event_t* ev, const uint32_t* idx, size_t n, int w, int h,
                                     size_t max_num, uint32_t* out_idx) {
  // createDenoisingMask (esvo_Mapping.cpp:1046-1054, Visualization.cpp:96-104) +
  // extractDenoisedEvents (:1056-1072); the mask is indexed by the RAW pixel (Appendix A-15)
  std::vector<uint8_t> em((size_t)w * h, 0), mask((size_t)w * h);
  for (size_t i = 0; i < n; ++i) { const esvo_event_t& e = ev[idx[i]]; if (e.x < w && e.y < h) em[(size_t)e.y * w + e.x] = 255; }
  median3_u8(em.data(), mask.data(), w, h);
  size_t cnt = 0;
  for (size_t i = 0; i < n; ++i) {
    if (cnt >= max_num) break;
    const esvo_event_t& e = ev[idx[i]];
    if (e.x < w && e.y < h && mask[(size_t)e.y * w + e.x] == 255) out_idx[cnt++] = idx[i];
  }
  return cnt;
}

extern "C" size_t orc_mapper_match(orc_mapper_handle h, const esvo_event_t* ev, size_t n, esvo_match_t* out,
                                   size_t cap) {
  // EventBM::match_all_HyperThread + match, EventBM.cpp:269-315: thread t handles events
  // t, t+N, ...; the per-thread result vectors are concatenated in thread order.
  BM bm{h, h->prm.patch_size_x, h->prm.patch_size_y, h->W(), h->H()};
  const int T = std::max(1, h->prm.num_threads);
  std::vector<std::vector<esvo_match_t>> per(T);
  auto job = [&](int t) {
    for (size_t i = t; i < n; i += T) {
      esvo_match_t m;
      if (bm.match_an_event(ev[i], (uint32_t)i, m, nullptr, h->bm_exact_int)) per[t].push_back(m);
    }
  };
  if (h->real_threads > 1) {
    // real threads only change wall time: each logical thread's list is still built in order
    std::vector<std::thread> th;
    int R = h->real_threads;
    // split every logical thread's stride list into R contiguous chunks
    std::vector<std::vector<std::vector<esvo_match_t>>> parts(T, std::vector<std::vector<esvo_match_t>>(R));
    for (int rt = 0; rt < R; ++rt)
      th.emplace_back([&, rt]() {
        for (int t = 0; t < T; ++t) {
          size_t cnt = (n > (size_t)t) ? (n - t + T - 1) / T : 0;
          size_t b = cnt * rt / R, e = cnt * (rt + 1) / R;
          for (size_t k = b; k < e; ++k) {
            size_t i = t + k * T;
            esvo_match_t m;
            if (bm.match_an_event(ev[i], (uint32_t)i, m, nullptr, h->bm_exact_int)) parts[t][rt].push_back(m);
          }
        }
      });
    for (auto& t : th) t.join();
    for (int t = 0; t < T; ++t) for (int rt = 0; rt < R; ++rt) per[t].insert(per[t].end(), parts[t][rt].begin(), parts[t][rt].end());
  } else {
    for (int t = 0; t < T; ++t) job(t);
  }
  size_t cnt = 0;
  for (int t = 0; t < T; ++t) for (auto& m : per[t]) { if (cnt < cap) out[cnt] = m; cnt++; }
  return cnt;
}

extern "C" int orc_mapper_match_costs(orc_mapper_handle h, const esvo_event_t* ev, double* costs, int exact_int) {
  BM bm{h, h->prm.patch_size_x, h->prm.patch_size_y, h->W(), h->H()};
  const int nd = h->prm.bm_max_disparity - h->prm.bm_min_disparity + 1;
  for (int i = 0; i < nd; ++i) costs[i] = std::numeric_limits<double>::quiet_NaN();
  esvo_match_t m;
  // run the search with cost capture (returns whether the event got far enough to search)
  bool ok = bm.match_an_event(*ev, 0, m, costs, exact_int);
  (void)ok;
  return std::isnan(costs[0]) ? 0 : 1;
}

extern "C" size_t orc_mapper_refine(orc_mapper_handle h, const esvo_match_t* matches, size_t n, int cull,
                                    esvo_depth_point_t* out, size_t cap, double* lm_info) {
  // DepthProblemSolver::solve + solve_multiple_problems, DepthProblemSolver.cpp:28-136
  const int T = std::max(1, h->prm.num_threads);
  const double nu = h->prm.td_nu;
  std::vector<char> solved(n, 0);
  std::vector<DP> res(n);
  auto solve_one = [&](size_t i, orc_mapper* ctx) {
    DepthProblem prob{ctx, h->prm.patch_size_x, h->prm.patch_size_y, {0, 0}, {0}};
    const esvo_match_t& m = matches[i];
    prob.setProblem(m.x_left, h->pose_T[m.pose_idx]);
    double result[3];
    if (solve_single(ctx, prob, m.inv_depth, result, lm_info ? lm_info + 4 * i : nullptr)) {
      DP dp((size_t)std::floor(m.x_left[1]), (size_t)std::floor(m.x_left[0]));  // :116
      dp.x[0] = m.x_left[0]; dp.x[1] = m.x_left[1];
      h->camL.cam2World(m.x_left, result[0], dp.p_cam);  // :119
      if (h->prm.ls_norm == ESVO_LSNORM_L2) {
        dp.update(result[0], result[1]);  // :121-122
      } else {
        const double scale2_rho = result[1] * (nu - 2) / nu;  // :125
        dp.update_studentT(result[0], scale2_rho, result[1], nu);
      }
      dp.residual = result[2];
      dp.pose_idx = m.pose_idx;
      res[i] = dp;
      solved[i] = 1;
    }
  };
  if (h->real_threads > 1) {
    int R = h->real_threads;
    std::vector<std::thread> th;
    for (int rt = 0; rt < R; ++rt)
      th.emplace_back([&, rt]() {
        orc_mapper local;  // only the counters are written through ctx; share read-only data
        local.prm = h->prm; local.camL = h->camL; local.camR = h->camR;
        local.tsL = h->tsL; local.tsR = h->tsR; local.T_world_obs = h->T_world_obs;
        local.lm_canonical = h->lm_canonical; local.bm_exact_int = h->bm_exact_int;
        for (size_t i = rt; i < n; i += R) solve_one(i, &local);
      });
    for (auto& t : th) t.join();
  } else {
    for (size_t i = 0; i < n; ++i) solve_one(i, h);
  }
  // concatenate in the reference's order: thread t -> matches t, t+N, ...
  std::vector<DP> vdp;
  for (size_t i : stride_order(n, T)) if (solved[i]) vdp.push_back(res[i]);
  if (cull) {  // DepthProblemSolver::pointCulling, DepthProblemSolver.cpp:216-244
    const double var_thr = sq(h->prm.stdvar_vis_threshold);
    const double cost_thr = sq(h->prm.residual_vis_threshold) * (h->prm.patch_size_x * h->prm.patch_size_y);
    std::vector<DP> c;
    for (auto& d : vdp)
      if (d.variance <= var_thr && d.residual <= cost_thr && d.valid() && d.invDepth >= h->prm.invdepth_min &&
          d.invDepth <= h->prm.invdepth_max)
        c.push_back(d);
    vdp.swap(c);
  }
  for (size_t i = 0; i < vdp.size() && i < cap; ++i) dp_to_pod(vdp[i], out[i], (uint32_t)i);
  return vdp.size();
}

extern "C" void orc_mapper_push_frame(orc_mapper_handle h, const esvo_depth_point_t* pts, size_t n,
                                      const double* pose_T, size_t m) {
  Frame f;
  f.pts.reserve(n);
  for (size_t i = 0; i < n; ++i) f.pts.push_back(pod_to_dp(pts[i]));
  f.poses.resize(m);
  for (size_t i = 0; i < m; ++i) std::memcpy(f.poses[i].m, pose_T + 16 * i, sizeof(double) * 16);
  h->window.push_back(std::move(f));
  // window policy, esvo_Mapping.cpp:341-368
  if (h->prm.fusion_strategy == ESVO_FUSION_CONST_POINTS) {
    auto total = [&]() { size_t s = 0; for (auto& fr : h->window) s += fr.pts.size(); return s; };
    size_t np = total();
    while ((double)np > 1.5 * (double)h->prm.max_fusion_points) { h->window.pop_front(); np = total(); }
  } else {
    while (h->window.size() > (size_t)h->prm.max_fusion_frames) h->window.pop_front();
  }
}

extern "C" size_t orc_mapper_fuse(orc_mapper_handle h) {
  // new DepthFrame at the TS pose (esvo_Mapping.cpp:268-272) + fusion newest -> oldest (:372-377)
  h->map.init(h->W(), h->H());
  h->T_world_frame = h->T_world_obs;
  size_t numFusion = 0;
  for (auto it = h->window.rbegin(); it != h->window.rend(); ++it) numFusion += fusion_update(h, *it, h->prm.fusion_radius);
  const bool do_clean = h->prm.clean_requires_full_window ? (h->window.size() >= (size_t)h->prm.max_fusion_frames) : true;
  if (do_clean)  // esvo_Mapping.cpp:385-386 / esvo_MVStereo.cpp:496-497
    h->map.clean(sq(h->prm.stdvar_vis_threshold), h->prm.age_vis_threshold, h->prm.invdepth_max, h->prm.invdepth_min);
  if (h->prm.regularization) regularize(h);  // :390-395
  return numFusion;
}

extern "C" size_t orc_mapper_tick(orc_mapper_handle h, const esvo_event_t* ev, size_t n) {
  // esvo_Mapping::MappingAtTime, esvo_Mapping.cpp:261-431 (events already selected / denoised)
  std::vector<esvo_match_t> vEMP(n);
  size_t nm = orc_mapper_match(h, ev, n, vEMP.data(), n);
  std::vector<esvo_depth_point_t> vdp(nm ? nm : 1);
  size_t np = orc_mapper_refine(h, vEMP.data(), nm, 1, vdp.data(), nm, nullptr);
  std::vector<double> poses(h->pose_T.size() * 16);
  for (size_t i = 0; i < h->pose_T.size(); ++i) std::memcpy(&poses[16 * i], h->pose_T[i].m, sizeof(double) * 16);
  orc_mapper_push_frame(h, vdp.data(), np, poses.data(), h->pose_T.size());
  return orc_mapper_fuse(h);
}

extern "C" size_t orc_mapper_map_size(orc_mapper_handle h) { return h->map.size(); }
extern "C" size_t orc_mapper_get_map(orc_mapper_handle h, esvo_depth_point_t* out, size_t cap) {
  size_t k = 0;
  for (size_t i = 0; i < h->map.elems.size(); ++i) {
    if (!h->map.alive[i]) continue;
    if (k < cap) dp_to_pod(h->map.elems[i], out[k], (uint32_t)k);
    ++k;
  }
  return k;
}
extern "C" size_t orc_mapper_get_map_cells(orc_mapper_handle h, int32_t* out, size_t cap) {
  std::vector<int32_t> cell_of(h->map.elems.size(), -1);
  for (size_t c = 0; c < h->map.grid.size(); ++c) if (h->map.grid[c] >= 0) cell_of[h->map.grid[c]] = (int32_t)c;
  size_t k = 0;
  for (size_t i = 0; i < h->map.elems.size(); ++i) {
    if (!h->map.alive[i]) continue;
    if (k < cap) out[k] = cell_of[i];
    ++k;
  }
  return k;
}
extern "C" size_t orc_mapper_get_last_frame(orc_mapper_handle h, esvo_depth_point_t* out, size_t cap) {
  if (h->window.empty()) return 0;
  const Frame& f = h->window.back();
  for (size_t i = 0; i < f.pts.size() && i < cap; ++i) dp_to_pod(f.pts[i], out[i], (uint32_t)i);
  return f.pts.size();
}
extern "C" size_t orc_mapper_get_pointcloud_xyz(orc_mapper_handle h, float* out_xyz, size_t cap_points) {
  // publishPointCloud loop, esvo_Mapping.cpp:925-932
  const Mat4& T = h->T_world_frame;
  size_t k = 0;
  for (size_t i = 0; i < h->map.elems.size(); ++i) {
    if (!h->map.alive[i]) continue;
    const DP& d = h->map.elems[i];
    if (k < cap_points)
      for (int r = 0; r < 3; ++r)
        out_xyz[3 * k + r] = (float)(((T.m[r * 4 + 0] * d.p_cam[0] + T.m[r * 4 + 1] * d.p_cam[1]) + T.m[r * 4 + 2] * d.p_cam[2]) + T.m[r * 4 + 3]);
    ++k;
  }
  return k;
}
extern "C" void orc_mapper_counters(orc_mapper_handle h, uint64_t out[8]) {
  size_t np = 0;
  for (auto& f : h->window) np += f.pts.size();
  out[0] = h->window.size(); out[1] = np; out[2] = h->n_replace; out[3] = h->n_replace_displaced;
  out[4] = h->max_scale_iters; out[5] = h->n_evals; out[6] = out[7] = 0;
}

// =====================================================================================================
// SGM initialisation: esvo_Mapping::InitializationAtTime, esvo_Mapping.cpp:433-492
// =====================================================================================================
// cv::StereoSGBM::compute as the reference configures it (esvo_Mapping.cpp:102-108: minDisparity 0, numDisparities 48,
// blockSize 11, P1 = 8*11*11, P2 = 32*11*11, disp12MaxDiff -1, preFilterCap 0, uniquenessRatio 11, no speckle filter,
// MODE_SGBM).  OpenCV is a third-party dependency absent from /root/reference and from this image: "parity unpinned".
// Restated from the published algorithm of modules/calib3d/src/stereosgbm.cpp (OpenCV 4.x, scalar path):
//   * pixel cost: Birchfield-Tomasi on the clipped x-Sobel image (ftzero = max(preFilterCap, 15) | 1 = 15) plus a
//     quarter of the BT cost on the raw image; the first and last column of both planes are the constant tab[0] = 15;
//     only x in [numDisparities, W) is matched
//   * block cost C: blockSize x blockSize box sum of the pixel cost with replicated borders (inside the matched range)
//   * single-pass mode = 5 paths: from the left, up-left, up, up-right, and from the right; along each
//       L(p,d) = C(p,d) + min(L(q,d), L(q,d-1)+P1, L(q,d+1)+P1, min_k L(q,k)+P2) - (min_k L(q,k) + P2),  q = p - r,
//     with L(q,.) = 0 outside the matched range (OpenCV's zeroed border cells); S = sat16(sat16(L0+L1+L2+L3) + L4)
//   * winner = first minimum of S; rejected unless unique (S(d)*(100-u) >= minS*100 for all |d - best| > 1); parabola
//     sub-pixel step in 1/16; left-right check with tolerance 1 (disp12MaxDiff <= 0 selects 1); 3x3 median of the result.
// disp: W*H int16, disparity * 16, (minDisparity - 1) * 16 = -16 where no match.
extern "C" void orc_sgbm_compute(const uint8_t* left, const uint8_t* right, int W, int H, int num_disp, int block, int P1, int P2,
                                 int uniqueness, int16_t* disp) {
  const int D = num_disp, minD = 0, DISP_SCALE = 16, INVALID = (minD - 1) * DISP_SCALE, MAX_COST = 32767;
  const int ftzero = 15, SW2 = block / 2, SH2 = block / 2;
  const int minX1 = D, width1 = W - minX1;
  for (size_t i = 0; i < (size_t)W * H; ++i) disp[i] = (int16_t)INVALID;
  if (width1 <= 0) return;
  auto clip = [&](int v) { return std::min(std::max(v, -ftzero), ftzero) + ftzero; };
  // pre-filtered planes: [0] clipped Sobel-x, [1] raw; border columns = tab[0]
  std::vector<uint8_t> pl[2][2];
  for (int im = 0; im < 2; ++im) {
    const uint8_t* img = im ? right : left;
    pl[im][0].assign((size_t)W * H, (uint8_t)clip(0));
    pl[im][1].assign((size_t)W * H, (uint8_t)clip(0));
    for (int y = 0; y < H; ++y) {
      const uint8_t* r0 = img + (size_t)y * W;
      const uint8_t* rn = img + (size_t)(y > 0 ? y - 1 : y) * W;
      const uint8_t* rs = img + (size_t)(y < H - 1 ? y + 1 : y) * W;
      for (int x = 1; x < W - 1; ++x) {
        pl[im][0][(size_t)y * W + x] = (uint8_t)clip((r0[x + 1] - r0[x - 1]) * 2 + rn[x + 1] - rn[x - 1] + rs[x + 1] - rs[x - 1]);
        pl[im][1][(size_t)y * W + x] = r0[x];
      }
    }
  }
  // pixel cost (Birchfield-Tomasi), x' = x - minX1
  const size_t rowN = (size_t)width1 * D;
  std::vector<int16_t> pix(rowN * H, 0), hs(rowN * H), C(rowN * H);
  for (int y = 0; y < H; ++y)
    for (int c = 0; c < 2; ++c) {
      const uint8_t* p1 = pl[0][c].data() + (size_t)y * W;
      const uint8_t* p2 = pl[1][c].data() + (size_t)y * W;
      const int shift = c ? 2 : 0;
      for (int x = minX1; x < W; ++x) {
        const int u = p1[x];
        const int ul = x > 0 ? (u + p1[x - 1]) / 2 : u, ur = x < W - 1 ? (u + p1[x + 1]) / 2 : u;
        const int u0 = std::min(std::min(ul, ur), u), u1 = std::max(std::max(ul, ur), u);
        for (int d = 0; d < D; ++d) {
          const int xr = x - d - minD;
          const int v = p2[xr];
          const int vl = xr < W - 1 ? (v + p2[xr + 1]) / 2 : v;  // OpenCV walks the right row mirrored: its "x-1" is xr+1
          const int vr = xr > 0 ? (v + p2[xr - 1]) / 2 : v;
          const int v0 = std::min(std::min(vl, vr), v), v1 = std::max(std::max(vl, vr), v);
          const int c0 = std::max(std::max(0, u - v1), v0 - u);
          const int c1 = std::max(std::max(0, v - u1), u0 - v);
          int16_t& o = pix[(size_t)y * rowN + (size_t)(x - minX1) * D + d];
          o = (int16_t)(o + (std::min(c0, c1) >> shift));
        }
      }
    }
  // block cost: horizontal then vertical box sum, replicated borders
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < width1; ++x)
      for (int d = 0; d < D; ++d) {
        int sum = 0;
        for (int i = -SW2; i <= SW2; ++i) sum += pix[(size_t)y * rowN + (size_t)std::min(std::max(x + i, 0), width1 - 1) * D + d];
        hs[(size_t)y * rowN + (size_t)x * D + d] = (int16_t)sum;
      }
  for (int y = 0; y < H; ++y)
    for (size_t i = 0; i < rowN; ++i) {
      int sum = 0;
      for (int k = -SH2; k <= SH2; ++k) sum += hs[(size_t)std::min(std::max(y + k, 0), H - 1) * rowN + i];
      C[(size_t)y * rowN + i] = (int16_t)sum;
    }
  // the five paths
  const int dirs[5][2] = {{-1, 0}, {-1, -1}, {0, -1}, {1, -1}, {1, 0}};  // q = p + dir
  std::vector<int16_t> L[5];
  for (int r = 0; r < 5; ++r) {
    L[r].assign(rowN * H, 0);
    std::vector<int16_t>& Lr = L[r];
    const int dx = dirs[r][0];
    for (int y = 0; y < H; ++y)
      for (int xi = 0; xi < width1; ++xi) {
        const int x = dx > 0 ? width1 - 1 - xi : xi;  // the neighbour must have been computed already
        const int qx = x + dx, qy = y + dirs[r][1];
        const bool inside = qx >= 0 && qx < width1 && qy >= 0 && qy < H;
        const int16_t* Lq = inside ? &Lr[(size_t)qy * rowN + (size_t)qx * D] : nullptr;
        int minq = 0;
        if (inside) { minq = MAX_COST; for (int d = 0; d < D; ++d) minq = std::min(minq, (int)Lq[d]); }
        const int delta = minq + P2;
        for (int d = 0; d < D; ++d) {
          const int a = inside ? Lq[d] : 0;
          const int bm = d > 0 ? (inside ? Lq[d - 1] : 0) : MAX_COST;
          const int bp = d < D - 1 ? (inside ? Lq[d + 1] : 0) : MAX_COST;
          const int v = C[(size_t)y * rowN + (size_t)x * D + d] + std::min(a, std::min(bm + P1, std::min(bp + P1, delta))) - delta;
          Lr[(size_t)y * rowN + (size_t)x * D + d] = (int16_t)v;
        }
      }
  }
  auto sat16 = [](int v) { return v > 32767 ? 32767 : (v < -32768 ? -32768 : v); };
  std::vector<int> Sp(D);
  std::vector<int16_t> d1((size_t)W, 0), d2((size_t)W, 0);
  std::vector<int> d2cost((size_t)W, 0);
  std::vector<int16_t> raw((size_t)W * H, (int16_t)INVALID);
  for (int y = 0; y < H; ++y) {
    for (int x = 0; x < W; ++x) { d1[x] = d2[x] = (int16_t)INVALID; d2cost[x] = MAX_COST; }
    for (int x = width1 - 1; x >= 0; --x) {
      const size_t o = (size_t)y * rowN + (size_t)x * D;
      int minS = MAX_COST, best = -1;
      for (int d = 0; d < D; ++d) {
        const int s4 = sat16(L[0][o + d] + L[1][o + d] + L[2][o + d] + L[3][o + d]);
        Sp[d] = sat16(s4 + L[4][o + d]);
        if (Sp[d] < minS) { minS = Sp[d]; best = d; }
      }
      int d;
      for (d = 0; d < D; ++d)
        if (Sp[d] * (100 - uniqueness) < minS * 100 && std::abs(best - d) > 1) break;
      if (d < D) continue;
      d = best;
      const int x2 = x + minX1 - d - minD;
      if (d2cost[x2] > minS) { d2cost[x2] = minS; d2[x2] = (int16_t)(d + minD); }
      if (0 < d && d < D - 1) {
        const int denom2 = std::max(Sp[d - 1] + Sp[d + 1] - 2 * Sp[d], 1);
        d = d * DISP_SCALE + ((Sp[d - 1] - Sp[d + 1]) * DISP_SCALE + denom2) / (denom2 * 2);
      } else {
        d *= DISP_SCALE;
      }
      d1[x + minX1] = (int16_t)(d + minD * DISP_SCALE);
    }
    for (int x = minX1; x < W; ++x) {  // left-right consistency, tolerance 1
      const int v = d1[x];
      if (v == INVALID) continue;
      const int lo = v >> 4, hi = (v + DISP_SCALE - 1) >> 4;
      const int xa = x - lo, xb = x - hi;
      if (0 <= xa && xa < W && d2[xa] >= minD && std::abs(d2[xa] - lo) > 1 && 0 <= xb && xb < W && d2[xb] >= minD &&
          std::abs(d2[xb] - hi) > 1)
        d1[x] = (int16_t)INVALID;
    }
    for (int x = 0; x < W; ++x) raw[(size_t)y * W + x] = d1[x];
  }
  // medianBlur(disp, disp, 3): 3x3 median, replicated border
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      int16_t v[9];
      int k = 0;
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx)
          v[k++] = raw[(size_t)std::min(std::max(y + dy, 0), H - 1) * W + std::min(std::max(x + dx, 0), W - 1)];
      std::nth_element(v, v + 4, v + 9);
      disp[(size_t)y * W + x] = v[4];
    }
}
// ---- wire format ---------------------------------------------------------------------------------
// What rosbag::MessageInstance::instantiate<dvs_msgs::EventArray>() does to a message's bytes at
// events_repacking_helper/src/EventMessageEditor.cpp:104-119 (ros::serialization of the generated message class; rpg_dvs_ros
// dvs_msgs/EventArray.msg: Header header; uint32 height; uint32 width; Event[] events -- dvs_msgs/Event.msg: uint16 x; uint16 y;
// time ts; bool polarity -- std_msgs/Header: uint32 seq; time stamp; string frame_id; all little-endian, arrays and strings
// with a uint32 length prefix): the events in memory, one esvo_event_t (= in-memory dvs_msgs::Event) each.
// Returns the event count, or -1 when the buffer is not a complete EventArray.  out may be null (count only).
extern "C" long orc_decode_event_array(const uint8_t* msg, size_t n_bytes, esvo_event_t* out, size_t cap, uint32_t* height,
                                       uint32_t* width) {
  auto u32 = [&](size_t o) { return (uint32_t)msg[o] | ((uint32_t)msg[o + 1] << 8) | ((uint32_t)msg[o + 2] << 16) | ((uint32_t)msg[o + 3] << 24); };
  size_t o = 0;
  if (n_bytes < 16) return -1;
  o += 4;                      // header.seq
  o += 8;                      // header.stamp (secs, nsecs)
  const uint32_t id_len = u32(o);
  o += 4;
  if ((size_t)id_len > n_bytes - o) return -1;
  o += id_len;                 // header.frame_id
  if (n_bytes - o < 12) return -1;
  const uint32_t h = u32(o), w = u32(o + 4), count = u32(o + 8);
  o += 12;
  if ((n_bytes - o) / 13 < count || (size_t)count * 13 != n_bytes - o) return -1;
  if (height) *height = h;
  if (width) *width = w;
  for (uint32_t i = 0; i < count && out && i < cap; ++i, o += 13) {
    esvo_event_t e;
    std::memset(&e, 0, sizeof(e));
    e.x = (uint16_t)(msg[o] | (msg[o + 1] << 8));
    e.y = (uint16_t)(msg[o + 2] | (msg[o + 3] << 8));
    e.sec = u32(o + 4);
    e.nsec = u32(o + 8);
    e.polarity = msg[o + 12] ? 1 : 0;   // bool: any non-zero byte deserialises to true
    out[i] = e;
  }
  return (long)count;
}

// the SGM branch of dataTransferring (esvo_Mapping.cpp:537-552): events of the last 2 * BM_half_slice_thickness walking
// back from lower_bound(t), while size <= PROCESS_EVENT_NUM (i.e. up to PROCESS_EVENT_NUM + 1 events)
extern "C" size_t orc_select_events_sgm(const esvo_event_t* ev, size_t n, uint64_t t_ns, double half_slice, size_t max_num,
                                        uint32_t* out_idx, size_t cap) {
  const double t_end = ns_to_sec(t_ns);
  const double t_begin = ns_to_sec(ros_time_from_sec(std::max(0.0, t_end - 2 * half_slice)));
  auto lb = [&](double t) {
    size_t lo = 0, hi = n;
    while (lo < hi) { size_t mid = (lo + hi) / 2; if (ev_sec(ev[mid]) < t) lo = mid + 1; else hi = mid; }
    return lo;
  };
  size_t it_end = lb(t_end), it_begin = lb(t_begin), cnt = 0;
  while (it_end != it_begin && cnt <= max_num) {
    if (it_end < n) { if (cnt < cap) out_idx[cnt] = (uint32_t)it_end; cnt++; }  // end() is skipped, as in orc_select_events
    it_end--;
  }
  return cnt;
}
// DepthFusion::naive_propagation, DepthFusion.cpp:234-288: every point of the frame into the current DepthFrame (Gaussian
// variance propagation, 2 x 2 cells, new / closer-and-better-residual replace); used by InitializationAtTime and by
// esvo_MVStereo's PURE_BLOCK_MATCHING mode
static void naive_propagation(orc_mapper* h, const Frame& f) {
  const size_t W = h->W(), H = h->H();
  const Mat4 T_frame_world = rigid_inverse(h->T_world_frame);
  for (const DP& prior : f.pts) {
    const Mat4 T = mat4_mul(T_frame_world, f.poses[prior.pose_idx]);
    double pp[3];
    for (int r = 0; r < 3; ++r)
      pp[r] = ((T.m[r * 4 + 0] * prior.p_cam[0] + T.m[r * 4 + 1] * prior.p_cam[1]) + T.m[r * 4 + 2] * prior.p_cam[2]) + T.m[r * 4 + 3];
    double xp[2];
    h->camL.world2Cam(pp, xp);
    if (!boundaryCheck(xp[0], xp[1], W, H) || !(xp[0] == xp[0]) || !(xp[1] == xp[1])) continue;
    DP prop((size_t)std::floor(xp[1]), (size_t)std::floor(xp[0]));
    prop.x[0] = xp[0]; prop.x[1] = xp[1];
    double denominator = (T.m[8] * prior.p_cam[0] + T.m[9] * prior.p_cam[1]) + T.m[11];
    denominator /= prior.p_cam[2];
    denominator += T.m[10];
    const double J = T.m[10] / sq(denominator);
    prop.invDepth = 1.0 / pp[2];
    prop.variance = J * J * prior.variance;
    if (prop.variance < 1e-6) prop.variance = 1e-6;  // boundVariance
    prop.p_cam[0] = pp[0]; prop.p_cam[1] = pp[1]; prop.p_cam[2] = pp[2];
    prop.residual = prior.residual;
    prop.age = prior.age;
    for (int dy = 0; dy <= 1; dy++)
      for (int dx = 0; dx <= 1; dx++) {
        const size_t row = prop.row + dy, col = prop.col + dx;
        if (!boundaryCheck((double)col, (double)row, W, H)) continue;
        if (!h->map.exists(row, col)) {
          DP nw(row, col);
          nw.invDepth = prop.invDepth;
          nw.variance = prop.variance < 1e-6 ? 1e-6 : prop.variance;
          nw.residual = prop.residual;
          nw.age = prop.age;
          h->camL.cam2World(nw.x, prop.invDepth, nw.p_cam);
          h->map.set(row, col, nw);
        } else {
          DP& c = h->map.get(row, col);
          if (c.invDepth > prop.invDepth) continue;
          if (prop.residual < c.residual) c = prop;
        }
      }
  }
}

// InitializationAtTime on the current observation (set_observation: the UN-smoothed pair is used, :444) and the SGM event
// selection; returns the number of SGM depth points, or 0 when fewer than min_points (INIT_SGM_DP_NUM_THRESHOLD) were
// found -- then nothing is pushed.  On success the points open the fusion window (:485) and naive_propagation fills the
// DepthFrame (DepthFusion.cpp:234-288).  disp_out (nullable): the disparity image.
// esvo_MVStereo::MappingAtTime in PURE_BLOCK_MATCHING mode (esvo_MVStereo.cpp:383-432): block matching, vEMP2vDP
// (:1072-1094: a Gaussian DepthPoint per match with pseudo-variance 0 -> the 1e-6 bound, residual = ZNCC cost, age =
// age_vis_threshold), a window of maxNumFusionFrames frames, naive_propagation of every frame (newest first) into a new
// DepthFrame.  Returns the number of matches.
extern "C" size_t orc_mapper_tick_bm_only(orc_mapper_handle h, const esvo_event_t* ev, size_t n) {
  std::vector<esvo_match_t> vEMP(n ? n : 1);
  const size_t nm = orc_mapper_match(h, ev, n, vEMP.data(), n);
  Frame f;
  f.poses = h->pose_T;
  for (size_t i = 0; i < nm; ++i) {
    const esvo_match_t& m = vEMP[i];
    DP dp((size_t)std::floor(m.x_left[1]), (size_t)std::floor(m.x_left[0]));
    dp.x[0] = m.x_left[0]; dp.x[1] = m.x_left[1];
    h->camL.cam2World(m.x_left, m.inv_depth, dp.p_cam);
    dp.update(m.inv_depth, 0.0);
    dp.residual = m.cost;
    dp.age = (size_t)h->prm.age_vis_threshold;
    dp.pose_idx = m.pose_idx;
    f.pts.push_back(dp);
  }
  h->window.push_back(std::move(f));
  while (h->window.size() > (size_t)h->prm.max_fusion_frames) h->window.pop_front();
  h->map.init(h->W(), h->H());
  h->T_world_frame = h->T_world_obs;
  for (auto it = h->window.rbegin(); it != h->window.rend(); ++it) naive_propagation(h, *it);
  return nm;
}

extern "C" size_t orc_mapper_init_sgm(orc_mapper_handle h, const uint8_t* ts_left, const uint8_t* ts_right, const esvo_event_t* ev,
                                      size_t n, size_t min_points, int16_t* disp_out) {
  const int W = h->W(), H = h->H();
  std::vector<int16_t> disp((size_t)W * H);
  orc_sgbm_compute(ts_left, ts_right, W, H, 48, 11, 8 * 11 * 11, 32 * 11 * 11, 11, disp.data());
  if (disp_out) std::memcpy(disp_out, disp.data(), sizeof(int16_t) * disp.size());
  const esvo_params_t& p = h->prm;
  Frame f;
  f.poses.push_back(h->T_world_obs);
  const double var_SGM = 0.001 * 0.001;
  for (size_t i = 0; i < n; ++i) {  // createEdgeMask with radius 0 (:1000-1044) + the loop of :456-480
    const esvo_event_t& e = ev[i];
    if (e.x >= W || e.y >= H) continue;
    const double cx = h->camL.lut[2 * ((size_t)e.y * W + e.x)], cy = h->camL.lut[2 * ((size_t)e.y * W + e.x) + 1];
    const int xc = (int)std::floor(cx), yc = (int)std::floor(cy);
    if (xc < 0 || xc >= W || yc < 0 || yc >= H) continue;
    const double d = disp[(size_t)yc * W + xc] / 16.0;
    if (d < 0) continue;
    DP dp((size_t)xc, (size_t)yc);  // DepthPoint dp(x, y): the constructor takes (row, col) -- reproduced as written
    dp.x[0] = xc * 1.0; dp.x[1] = yc * 1.0;
    const double invDepth = d / (h->camL.P[0] * h->baseline);
    if (invDepth < p.invdepth_min || invDepth > p.invdepth_max) continue;
    h->camL.cam2World(dp.x, invDepth, dp.p_cam);
    dp.invDepth = invDepth;  // DepthPoint::update on a new point (DepthPoint.cpp:146-164) + boundVariance
    dp.variance = var_SGM < 1e-6 ? 1e-6 : var_SGM;
    dp.residual = 0.0;
    dp.age = (size_t)p.age_vis_threshold;
    dp.pose_idx = 0;
    f.pts.push_back(dp);
  }
  if (f.pts.size() < min_points) return 0;
  const size_t n_pts = f.pts.size();
  h->map.init(W, H);
  h->T_world_frame = h->T_world_obs;
  naive_propagation(h, f);
  h->window.push_back(std::move(f));  // dqvDepthPoints_.push_back(vdp_sgm): no window policy here (:485)
  return n_pts;
}

// =====================================================================================================
// Debug image publishers and the global-cloud voxel filter
// =====================================================================================================
// The 256 colours of Visualization::DrawPoint (Visualization.cpp:74-94,128-226) as the BGR bytes an 8-bit image stores:
// the reference's tables are jet on i / 255, i.e. 255 * channel = clamp(min(4 i + a, -4 i + b), 0, 255), rounded half
// to even (tests/golden/jet256.npy holds the bytes computed from the reference's own tables).
extern "C" void orc_jet_bgr(uint8_t out[768]) {
  const double ab[3][2] = {{127.5, 637.5}, {-127.5, 892.5}, {-382.5, 1147.5}};  // B, G, R
  for (int i = 0; i < 256; ++i)
    for (int c = 0; c < 3; ++c) {
      double v = std::min(4.0 * i + ab[c][0], -4.0 * i + ab[c][1]);
      v = v < 0 ? 0 : (v > 255 ? 255 : v);
      out[3 * i + c] = (uint8_t)std::nearbyint(v);
    }
}
// Visualization::plot_map (Visualization.cpp:13-72) with the arguments of publishMappingResults (esvo_Mapping.cpp:868-884);
// type: 0 InvDepthMap, 1 StdVarMap, 2 CostMap, 3 AgeMap.  bgr: H*W*3, zeroed here.
extern "C" void orc_mapper_debug_image(orc_mapper_handle h, int type, double age_max_range, uint8_t* bgr) {
  const int W = h->W(), H = h->H();
  std::memset(bgr, 0, (size_t)W * H * 3);
  uint8_t jet[768];
  orc_jet_bgr(jet);
  const esvo_params_t& p = h->prm;
  const double cost_thr = sq(p.residual_vis_threshold) * (p.patch_size_x * p.patch_size_y);  // esvo_Mapping.cpp:97
  double max_range = 0, min_range = 0, thr1 = 0, thr2 = 0;
  switch (type) {
    case 0: max_range = p.invdepth_max; min_range = p.invdepth_min; thr1 = p.stdvar_vis_threshold; thr2 = p.age_vis_threshold; break;
    case 1: max_range = p.stdvar_vis_threshold; min_range = 0; thr1 = p.stdvar_vis_threshold; break;
    case 2: max_range = cost_thr; min_range = 0; thr1 = cost_thr; break;
    default: max_range = age_max_range; min_range = 0; thr1 = p.age_vis_threshold; break;
  }
  for (size_t e = 0; e < h->map.elems.size(); ++e) {
    if (!h->map.alive[e]) continue;
    const DP& d = h->map.elems[e];
    if (!d.valid()) continue;
    double val;
    if (type == 0) { if (!(d.variance < sq(thr1) && (double)d.age >= (double)(int)thr2)) continue; val = d.invDepth; }
    else if (type == 1) { if (!(d.variance < sq(thr1))) continue; val = std::sqrt(d.variance); }
    else if (type == 2) { if (!(d.residual < thr1)) continue; val = d.residual; }
    else { if (!((double)d.age >= (double)(int)thr1)) continue; val = (double)d.age; }
    int index = (int)std::floor((val - min_range) / (max_range - min_range) * 255.0);  // DrawPoint, :82
    if (index > 255) index = 255;
    if (index < 0) index = 0;
    const int cx = (int)d.x[0], cy = (int)d.x[1];  // cv::Point from doubles: truncation
    // cv::circle(img, point, 1, color, cv::FILLED): the 5-pixel plus, clipped to the image
    const int px[5] = {cx, cx - 1, cx + 1, cx, cx}, py[5] = {cy, cy, cy, cy - 1, cy + 1};
    for (int k = 0; k < 5; ++k)
      if (px[k] >= 0 && px[k] < W && py[k] >= 0 && py[k] < H)
        for (int c = 0; c < 3; ++c) bgr[((size_t)py[k] * W + px[k]) * 3 + c] = jet[3 * index + c];
  }
}
// pc_near_ of publishPointCloud (esvo_Mapping.cpp:925-932): world points of the elements with |p_cam| < visualize_range
extern "C" size_t orc_mapper_get_pointcloud_near_xyz(orc_mapper_handle h, double visualize_range, float* out_xyz, size_t cap_points) {
  const Mat4& T = h->T_world_frame;
  size_t k = 0;
  for (size_t i = 0; i < h->map.elems.size(); ++i) {
    if (!h->map.alive[i]) continue;
    const DP& d = h->map.elems[i];
    const double nrm = std::sqrt((d.p_cam[0] * d.p_cam[0] + d.p_cam[1] * d.p_cam[1]) + d.p_cam[2] * d.p_cam[2]);
    if (!(nrm < visualize_range)) continue;
    if (k < cap_points)
      for (int r = 0; r < 3; ++r)
        out_xyz[3 * k + r] = (float)(((T.m[r * 4 + 0] * d.p_cam[0] + T.m[r * 4 + 1] * d.p_cam[1]) + T.m[r * 4 + 2] * d.p_cam[2]) + T.m[r * 4 + 3]);
    ++k;
  }
  return k;
}
// pcl::VoxelGrid<PointXYZ> with setLeafSize(leaf, leaf, leaf) as used at esvo_Mapping.cpp:960-964 (PCL is a third-party
// dependency absent from /root/reference and from this image: restated from its published algorithm -- bounding box of
// the finite points, voxel index = floor(p / leaf) - floor(min / leaf) linearised x-fastest, one centroid per occupied
// voxel in ascending index order, all in float; points of a voxel are summed in input order).  Returns the voxel count.
extern "C" size_t orc_voxel_filter(const float* xyz, size_t n, float leaf, float* out) {
  std::vector<size_t> fin;
  float mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
  for (size_t i = 0; i < n; ++i) {
    const float* p = xyz + 3 * i;
    if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
    if (fin.empty()) for (int c = 0; c < 3; ++c) mn[c] = mx[c] = p[c];
    for (int c = 0; c < 3; ++c) { mn[c] = std::min(mn[c], p[c]); mx[c] = std::max(mx[c], p[c]); }
    fin.push_back(i);
  }
  if (fin.empty()) return 0;
  const float inv = 1.0f / leaf;
  int minb[3], divb[3];
  for (int c = 0; c < 3; ++c) {
    minb[c] = (int)std::floor(mn[c] * inv);
    divb[c] = (int)std::floor(mx[c] * inv) - minb[c] + 1;
  }
  const long long mul[3] = {1, divb[0], (long long)divb[0] * divb[1]};
  std::vector<std::pair<long long, size_t>> idx;
  idx.reserve(fin.size());
  for (size_t i : fin) {
    const float* p = xyz + 3 * i;
    long long v = 0;
    for (int c = 0; c < 3; ++c) v += ((long long)((int)std::floor(p[c] * inv) - minb[c])) * mul[c];
    idx.emplace_back(v, i);
  }
  std::stable_sort(idx.begin(), idx.end(), [](const std::pair<long long, size_t>& a, const std::pair<long long, size_t>& b) { return a.first < b.first; });
  size_t k = 0;
  for (size_t a = 0; a < idx.size();) {
    size_t b = a;
    float c[3] = {0, 0, 0};
    while (b < idx.size() && idx[b].first == idx[a].first) {
      for (int d = 0; d < 3; ++d) c[d] += xyz[3 * idx[b].second + d];
      ++b;
    }
    for (int d = 0; d < 3; ++d) out[3 * k + d] = c[d] / (float)(b - a);
    ++k;
    a = b;
  }
  return k;
}

// residual vector of DepthProblem::operator() for one match at inverse depth rho (unit tests:
// comparison of the restated LM against scipy's MINPACK wrapper)
extern "C" int orc_mapper_eval_residual(orc_mapper_handle h, const double x_left[2], uint32_t pose_idx, double rho,
                                        double* fvec) {
  DepthProblem prob{h, h->prm.patch_size_x, h->prm.patch_size_y, {0, 0}, {0}};
  prob.setProblem(x_left, h->pose_T[pose_idx]);
  return prob(rho, fvec);
}
// ZNCC cost of two wy x wx patches (row-major doubles), literal and integer-moment forms
// =====================================================================================================
// Tracker: residual and Jacobian evaluation -- esvo_core/src/core/RegProblemLM.cpp with the
// shipped settings (patch 1x1, kernelSize 5, Huber, analytical Jacobian; cfg/tracking/*.yaml)
// =====================================================================================================
struct orc_tracker {
  Camera camL;
  int W = 0, H = 0;
  std::vector<double> neg;      // TS_negative_left_ (row-major)
  std::vector<double> du, dv;   // dTS_negative_du/dv_left_ (cv::Sobel, CV_64F, ksize 3, BORDER_REFLECT_101)
  std::vector<double> pts;      // ResItems_[i].p_ : 3 doubles per point, ref camera frame
};

namespace {
// RegProblemLM::isValidPatch (:366-385) for wx = wy = 1: (w-1)/2 == 0 in size_t arithmetic; Eigen index = truncation
bool trk_valid_patch(const orc_tracker* T, const double x[2]) {
  if (!std::isfinite(x[0]) || !std::isfinite(x[1])) return false;  // oracle definition (UB in the reference)
  if (x[0] < 0.0 || x[0] > (double)(T->W - 1) || x[1] < 0.0 || x[1] > (double)(T->H - 1)) return false;
  if (!T->camL.mask.empty() && T->camL.mask[(size_t)(int)x[1] * T->W + (int)x[0]] < 125) return false;
  return true;
}
// RegProblemLM::reprojection (:387-401)
bool trk_reproject(const orc_tracker* T, const double p[3], const double Tw[16], double x[2]) {
  double pl[3];
  for (int r = 0; r < 3; ++r) pl[r] = ((Tw[r * 4 + 0] * p[0] + Tw[r * 4 + 1] * p[1]) + Tw[r * 4 + 2] * p[2]) + Tw[r * 4 + 3];
  T->camL.world2Cam(pl, x);
  return trk_valid_patch(T, x);
}
// RegProblemLM::patchInterpolation (:403-483) for a 1x1 patch
bool trk_interp(const orc_tracker* T, const std::vector<double>& img, const double loc[2], double* out) {
  const int ux = (int)std::floor(loc[0]), uy = (int)std::floor(loc[1]);
  if (ux < 0 || uy < 0) return false;
  if (ux >= T->W || uy >= T->H) return false;
  const double q1 = (double)(ux + 1) - loc[0], q2 = loc[0] - (double)ux;
  const double q3 = (double)(uy + 1) - loc[1], q4 = loc[1] - (double)uy;
  if (uy + 1 >= T->H || ux + 1 >= T->W) return false;
  const double* s = img.data() + (size_t)uy * T->W + ux;
  const double r0 = q1 * s[0] + q2 * s[1];
  const double r1 = q1 * s[T->W] + q2 * s[T->W + 1];
  *out = q3 * r0 + q4 * r1;
  return true;
}
}  // namespace

extern "C" orc_tracker_handle orc_tracker_create(const esvo_calib_t* left) {
  orc_tracker* T = new orc_tracker();
  T->camL.init(left);
  T->W = left->width; T->H = left->height;
  return T;
}
extern "C" void orc_tracker_destroy(orc_tracker_handle T) { delete T; }

// TimeSurfaceObservation::getTimeSurfaceNegative(kernelSize) + computeTsNegativeGrad (TimeSurfaceObservation.h:118-147)
extern "C" int orc_tracker_set_current(orc_tracker_handle T, const uint8_t* ts_left, int kernel_size) {
  const int W = T->W, H = T->H;
  std::vector<uint8_t> blur((size_t)W * H);
  if (kernel_size == 0) std::memcpy(blur.data(), ts_left, blur.size());
  else if (kernel_size == 5) gaussian5_u8(ts_left, blur.data(), W, H);  // 8-bit GaussianBlur, sigma 0 (Appendix B.2)
  else return -1;
  T->neg.resize((size_t)W * H); T->du.resize(T->neg.size()); T->dv.resize(T->neg.size());
  for (size_t i = 0; i < T->neg.size(); ++i) T->neg[i] = 255.0 - (double)blur[i];
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      auto at = [&](int yy, int xx) { return T->neg[(size_t)reflect101(yy, H) * W + reflect101(xx, W)]; };
      // separable Sobel: derivative [-1 0 1] along one axis, smoothing [1 2 1] along the other (exact in f64)
      T->du[(size_t)y * W + x] = (at(y - 1, x + 1) - at(y - 1, x - 1)) + 2.0 * (at(y, x + 1) - at(y, x - 1)) + (at(y + 1, x + 1) - at(y + 1, x - 1));
      T->dv[(size_t)y * W + x] = (at(y + 1, x - 1) - at(y - 1, x - 1)) + 2.0 * (at(y + 1, x) - at(y - 1, x)) + (at(y + 1, x + 1) - at(y - 1, x + 1));
    }
  return 0;
}
extern "C" void orc_tracker_get_images(orc_tracker_handle T, uint8_t* neg, int16_t* du, int16_t* dv) {
  for (size_t i = 0; i < T->neg.size(); ++i) { neg[i] = (uint8_t)T->neg[i]; du[i] = (int16_t)T->du[i]; dv[i] = (int16_t)T->dv[i]; }
}

// the point loop of RegProblemLM::setProblem (:44-56): p_cam = R_world_ref^T (p - t_world_ref); the caller has
// already applied the stochastic swaps (:48-49, rand())
extern "C" void orc_tracker_set_reference(orc_tracker_handle T, const float* xyz_world, size_t n, const double T_world_ref[16]) {
  T->pts.resize(3 * n);
  for (size_t i = 0; i < n; ++i) {
    double d[3];
    for (int k = 0; k < 3; ++k) d[k] = (double)xyz_world[3 * i + k] - T_world_ref[k * 4 + 3];
    for (int r = 0; r < 3; ++r)  // row r of R^T = column r of R
      T->pts[3 * i + r] = (T_world_ref[0 * 4 + r] * d[0] + T_world_ref[1 * 4 + r] * d[1]) + T_world_ref[2 * 4 + r] * d[2];
  }
}

// RegProblemLM::operator() (:91-136) + thread() (:138-176) over ResItems [offset, offset + count)
extern "C" size_t orc_tracker_residuals(orc_tracker_handle T, const double T_left_ref[16], size_t offset, size_t count,
                                        int huber, double huber_threshold, double* fvec) {
  size_t n = 0;
  for (size_t i = offset; i < offset + count && 3 * i < T->pts.size(); ++i, ++n) {
    double x[2], r = 255.0, tau;
    if (trk_reproject(T, &T->pts[3 * i], T_left_ref, x) && trk_interp(T, T->neg, x, &tau)) r = tau;
    if (huber) {
      double w = 1.0;
      if (r > huber_threshold) w = huber_threshold / r;
      fvec[n] = std::sqrt(w) * r;
    } else {
      fvec[n] = r;
    }
  }
  return n;
}

// RegProblemLM::df at x = 0 (:178-269).  R, t are the problem's R_, t_ (T_ref_left); fjac is count x 6, column-major.
// Products are taken left to right as written at :236; J_G_0_ (computeJ_G at zero, :271-326) has the entries 0, +-2, 1.
extern "C" size_t orc_tracker_jacobian(orc_tracker_handle T, const double R[9], const double t[3], size_t offset, size_t count,
                                       double* fjac) {
  size_t m = 0;
  for (size_t i = offset; i < offset + count && 3 * i < T->pts.size(); ++i) ++m;
  const double* P = T->camL.P;
  const double P11 = P[0], P12 = P[1], P14 = P[3], P21 = P[4], P22 = P[5], P24 = P[7];
  double Tlr[16] = {0};  // T_left_ref = [R^T | -R^T t]
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) Tlr[r * 4 + c] = R[c * 3 + r];
    Tlr[r * 4 + 3] = (-R[0 * 3 + r] * t[0] + -R[1 * 3 + r] * t[1]) + -R[2 * 3 + r] * t[2];
  }
  Tlr[15] = 1.0;
  double Jc[3][2];  // J_constPart = R^T * diag(1/P11, 1/P22; 0)
  const double iP11 = 1.0 / P11, iP22 = 1.0 / P22;
  for (int r = 0; r < 3; ++r) { Jc[r][0] = R[0 * 3 + r] * iP11; Jc[r][1] = R[1 * 3 + r] * iP22; }
  for (size_t k = 0; k < m; ++k) {
    const double* p = &T->pts[3 * (offset + k)];
    double e[12];
    double x[2];
    if (!trk_reproject(T, p, Tlr, x)) {
      for (int j = 0; j < 12; ++j) e[j] = 0.0;
    } else {
      double gx = 0.0, gy = 0.0;  // the reference ignores patchInterpolation's failure here (border pixel): defined as 0
      trk_interp(T, T->du, x, &gx);
      trk_interp(T, T->dv, x, &gy);
      const double g0 = gx / 8, g1 = gy / 8;  // 8 = normalisation of the 3x3 Sobel filter
      const double z = p[2], z2 = z * z;
      double D[2][3];
      D[0][0] = P11 / z; D[0][1] = P12 / z; D[0][2] = -((P11 * p[0] + P12 * p[1]) + P14) / z2;
      D[1][0] = P21 / z; D[1][1] = P22 / z; D[1][2] = -((P21 * p[0] + P22 * p[1]) + P24) / z2;
      double a[3], b[2], c[3];
      for (int j = 0; j < 3; ++j) a[j] = g0 * D[0][j] + g1 * D[1][j];
      for (int j = 0; j < 2; ++j) b[j] = (a[0] * Jc[0][j] + a[1] * Jc[1][j]) + a[2] * Jc[2][j];
      for (int j = 0; j < 3; ++j) c[j] = b[0] * D[0][j] + b[1] * D[1][j];
      for (int j = 0; j < 3; ++j) {  // dT_dG = [x I, y I, z I, I], then the factor p_z
        e[j] = (c[j] * p[0]) * z; e[3 + j] = (c[j] * p[1]) * z; e[6 + j] = (c[j] * p[2]) * z; e[9 + j] = c[j] * z;
      }
    }
    // fjac = -fjacBlock * J_G_0_
    fjac[0 * m + k] = -(2.0 * e[5] - 2.0 * e[7]);
    fjac[1 * m + k] = -(2.0 * e[6] - 2.0 * e[2]);
    fjac[2 * m + k] = -(2.0 * e[1] - 2.0 * e[3]);
    fjac[3 * m + k] = -e[9];
    fjac[4 * m + k] = -e[10];
    fjac[5 * m + k] = -e[11];
  }
  return m;
}

// The normal equations of one tracker iteration (RegProblemSolverLM::solve_analytical, RegProblemSolverLM.cpp:148-215, consumes
// F(0) and df(0); the device hands back their products): out[0..20] = upper triangle of J^T J (row-major, i <= j),
// out[21..26] = J^T f, out[27] = |f|^2, summed in the device's order (kernels_track.hip, track_normal_kernel): partial sum t of
// 256 takes the points t, t + 256, ... in turn, then the tree s[t] += s[t + 128], ..., s[0] += s[1].
extern "C" size_t orc_tracker_normal_equations(orc_tracker_handle T, const double R[9], const double t[3], size_t offset, size_t count,
                                               int huber, double huber_threshold, double out[28]) {
  size_t m = 0;
  for (size_t i = offset; i < offset + count && 3 * i < T->pts.size(); ++i) ++m;
  for (int n = 0; n < 28; ++n) out[n] = 0.0;
  if (m == 0) return 0;
  std::vector<double> fjac(6 * m), fvec(m);
  double Tlr[16] = {0};  // T_left_ref = [R^T | -R^T t], as orc_tracker_jacobian forms it
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) Tlr[r * 4 + c] = R[c * 3 + r];
    Tlr[r * 4 + 3] = (-R[0 * 3 + r] * t[0] + -R[1 * 3 + r] * t[1]) + -R[2 * 3 + r] * t[2];
  }
  Tlr[15] = 1.0;
  orc_tracker_residuals(T, Tlr, offset, m, huber, huber_threshold, fvec.data());
  orc_tracker_jacobian(T, R, t, offset, m, fjac.data());
  const int NT = 256;
  std::vector<double> part((size_t)28 * NT, 0.0);
  for (int th = 0; th < NT; ++th) {
    double* acc = &part[(size_t)th * 28];
    for (size_t k = (size_t)th; k < m; k += NT) {
      double row[6];
      for (int j = 0; j < 6; ++j) row[j] = fjac[(size_t)j * m + k];
      const double f = fvec[k];
      int n = 0;
      for (int i = 0; i < 6; ++i)
        for (int j = i; j < 6; ++j) { acc[n] = acc[n] + row[i] * row[j]; ++n; }
      for (int i = 0; i < 6; ++i) acc[21 + i] = acc[21 + i] + row[i] * f;
      acc[27] = acc[27] + f * f;
    }
  }
  for (int s = NT / 2; s > 0; s >>= 1)
    for (int th = 0; th < s; ++th)
      for (int n = 0; n < 28; ++n) part[(size_t)th * 28 + n] = part[(size_t)th * 28 + n] + part[(size_t)(th + s) * 28 + n];
  for (int n = 0; n < 28; ++n) out[n] = part[n];
  return m;
}

extern "C" double orc_zncc_cost(const double* l, const double* r, int wx, int wy, int exact_int) {
  if (!exact_int) {
    std::vector<double> t1((size_t)wx * wy), t2((size_t)wx * wy);
    return zncc_cost(l, r, wx, wy, t1.data(), t2.data());
  }
  int64_t Sl = 0, Sll = 0, Sr = 0, Srr = 0, Slr = 0;
  for (int i = 0; i < wx * wy; ++i) {
    int64_t a = (int64_t)l[i], b = (int64_t)r[i];
    Sl += a; Sll += a * a; Sr += b; Srr += b * b; Slr += a * b;
  }
  return zncc_cost_int(Sl, Sll, Sr, Srr, Slr, wx * wy);
}
extern "C" void orc_abi_sizes(size_t out[8]) {
  out[0] = sizeof(esvo_event_t); out[1] = sizeof(esvo_calib_t); out[2] = sizeof(esvo_params_t);
  out[3] = sizeof(esvo_match_t); out[4] = sizeof(esvo_depth_point_t); out[5] = sizeof(esvo_stats_t);
  out[6] = 0; out[7] = 0;
}
