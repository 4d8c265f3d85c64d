// esvo_hip.hpp — C++ host layer over the C-ABI (esvo_hip.h), mirroring the reference's own
// class interface for the hot path so that ESVO's node code (and tests written like it) keeps
// its shape:
//
//   esvo_time_surface::TimeSurface::eventsCallback / createTimeSurfaceAtTime
//                                     (esvo_time_surface/src/TimeSurface.cpp:403-425, :52-152)
//   esvo_core::core::EventBM::resetParameters / createMatchProblem / match_all_HyperThread
//                                     (esvo_core/src/core/EventBM.cpp:34-78, :269-315)
//   esvo_core::core::DepthProblemSolver::solve / pointCulling
//                                     (esvo_core/src/core/DepthProblemSolver.cpp:28-78, :216-244)
//   esvo_core::core::DepthFusion (window push + update loop) / DepthMap::clean /
//   DepthRegularization::apply        (esvo_core/src/esvo_Mapping.cpp:341-395)
//   esvo_core::core::RegProblemLM::setProblem / operator() / df   (the tracker's evaluation side,
//                                     esvo_core/src/core/RegProblemLM.cpp:26-269)
//
// Header only, C++14, no ROS / Eigen / OpenCV types: poses are row-major 4x4 doubles, images are
// mono8 pointers, events are esvo_event_t (layout-identical to dvs_msgs::Event).  Every method
// throws esvo_hip::Error carrying esvo_last_error() — the reference's failure mode for these calls
// is exit(-1) or a silent return.  All compute happens in libesvo_hip.so on the GPU.
#ifndef ESVO_HIP_HPP
#define ESVO_HIP_HPP

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "esvo_hip.h"

namespace esvo_hip {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

using Event = esvo_event_t;
using EventMatchPair = esvo_match_t;   // x_left, invDepth, cost, disp, pose index (EventMatchPair.h:16-38)
using DepthPoint = esvo_depth_point_t; // every field of DepthPoint.h:70-88

// std::map<ros::Time, Transformation> st_map_ (esvo_Mapping.cpp:585-599), flattened
struct StampTransformationMap {
  std::vector<uint64_t> stamps_ns;
  std::vector<double> T_world_virtual;  // 16 per stamp, row-major
  size_t size() const { return stamps_ns.size(); }
  void emplace(uint64_t t_ns, const double T[16]) {
    stamps_ns.push_back(t_ns);
    T_world_virtual.insert(T_world_virtual.end(), T, T + 16);
  }
  void clear() { stamps_ns.clear(); T_world_virtual.clear(); }
};

// StampedTimeSurfaceObs (TimeSurfaceObservation.h:27-157): mono8 pair + pose.  left/right may be
// null: the pair rendered last on the device is used.
struct StampedTimeSurfaceObs {
  uint64_t t_ns = 0;
  const uint8_t* TS_left = nullptr;
  const uint8_t* TS_right = nullptr;
  double T_world_cam[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
};

// One GPU context shared by the TimeSurface nodes and the mapper (RAII over esvo_handle).
class Context {
 public:
  Context(const esvo_params_t& params, const esvo_calib_t& left, const esvo_calib_t& right, int device = 0)
      : params_(params), width_(left.width), height_(left.height) {
    int rc = esvo_create(&params_, &left, &right, device, &h_);
    if (rc != ESVO_OK) throw Error(rc, std::string("esvo_create: ") + esvo_last_error(nullptr));
  }
  ~Context() { if (h_) esvo_destroy(h_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  esvo_handle handle() const { return h_; }
  const esvo_params_t& params() const { return params_; }
  int width() const { return width_; }
  int height() const { return height_; }
  void check(int rc, const char* what) const {
    if (rc != ESVO_OK) throw Error(rc, std::string(what) + ": " + esvo_last_error(h_));
  }
  void setParams(const esvo_params_t& p) { check(esvo_set_params(h_, &p), "esvo_set_params"); params_ = p; }
  void reset() { check(esvo_reset(h_), "esvo_reset"); }  // esvo_Mapping::reset, esvo_Mapping.cpp:764-804

 private:
  esvo_params_t params_;
  int width_, height_;
  esvo_handle h_ = nullptr;
};
using ContextPtr = std::shared_ptr<Context>;

// esvo_time_surface::TimeSurface for one camera
class TimeSurface {
 public:
  TimeSurface(ContextPtr ctx, int cam) : ctx_(std::move(ctx)), cam_(cam) {}
  // TimeSurface::eventsCallback: events of one EventArray message, time sorted
  void eventsCallback(const Event* events, size_t n) {
    ctx_->check(esvo_ts_push_events(ctx_->handle(), cam_, events, n), "esvo_ts_push_events");
  }
  // TimeSurface::createTimeSurfaceAtTime: the rectified mono8 Time Surface at the sync time
  void createTimeSurfaceAtTime(uint64_t external_sync_time_ns, uint8_t* out_mono8 /*nullable*/) {
    ctx_->check(esvo_ts_render(ctx_->handle(), cam_, external_sync_time_ns, out_mono8), "esvo_ts_render");
  }

 private:
  ContextPtr ctx_;
  int cam_;
};

// esvo_core::core::EventBM
class EventBM {
 public:
  explicit EventBM(ContextPtr ctx) : ctx_(std::move(ctx)) {}
  void resetParameters(size_t patch_size_X, size_t patch_size_Y, size_t min_disparity, size_t max_disparity, size_t step,
                       double ZNCC_Threshold, bool bUpDownConfiguration) {
    esvo_params_t p = ctx_->params();
    p.patch_size_x = (int32_t)patch_size_X; p.patch_size_y = (int32_t)patch_size_Y;
    p.bm_min_disparity = (int32_t)min_disparity; p.bm_max_disparity = (int32_t)max_disparity;
    p.bm_step = (int32_t)step; p.bm_zncc_threshold = ZNCC_Threshold; p.bm_updown = bUpDownConfiguration;
    ctx_->setParams(p);
  }
  void createMatchProblem(const StampedTimeSurfaceObs* pStampedTsObs, const StampTransformationMap* pSt_map,
                          const std::vector<Event>* pvEvents) {
    obs_ = pStampedTsObs; st_map_ = pSt_map; events_ = pvEvents;
    ctx_->check(esvo_map_set_observation(ctx_->handle(), obs_->t_ns, obs_->TS_left, obs_->TS_right, obs_->T_world_cam),
                "esvo_map_set_observation");
  }
  void match_all_HyperThread(std::vector<EventMatchPair>& vEMP) {
    vEMP.resize(events_->size());
    size_t n = 0;
    ctx_->check(esvo_map_match(ctx_->handle(), events_->data(), events_->size(), st_map_->stamps_ns.data(),
                               st_map_->T_world_virtual.data(), st_map_->size(), vEMP.data(), vEMP.size(), &n),
                "esvo_map_match");
    vEMP.resize(n);
  }

 private:
  ContextPtr ctx_;
  const StampedTimeSurfaceObs* obs_ = nullptr;
  const StampTransformationMap* st_map_ = nullptr;
  const std::vector<Event>* events_ = nullptr;
};

// esvo_core::core::DepthProblemSolver (NUMERICAL problem type, Tdist norm)
class DepthProblemSolver {
 public:
  explicit DepthProblemSolver(ContextPtr ctx) : ctx_(std::move(ctx)) {}
  void solve(const std::vector<EventMatchPair>* pvEMP, const StampedTimeSurfaceObs* /*the observation set by EventBM*/,
             std::vector<DepthPoint>& vdp) {
    vdp.resize(pvEMP->size());
    size_t n = 0;
    ctx_->check(esvo_map_refine(ctx_->handle(), pvEMP->data(), pvEMP->size(), /*cull*/ 0, vdp.data(), vdp.size(), &n),
                "esvo_map_refine");
    vdp.resize(n);
  }
  // DepthProblemSolver::pointCulling (DepthProblemSolver.cpp:216-244): a stable host-side filter
  static void pointCulling(std::vector<DepthPoint>& vdp, double std_variance_threshold, double cost_threshold,
                           double invDepth_min_range, double invDepth_max_range) {
    std::vector<DepthPoint> kept;
    kept.reserve(vdp.size());
    for (const DepthPoint& d : vdp)
      if (d.variance <= std::pow(std_variance_threshold, 2) && d.residual <= cost_threshold && d.inv_depth > -1e-6 &&
          d.inv_depth >= invDepth_min_range && d.inv_depth <= invDepth_max_range)
        kept.push_back(d);
    vdp.swap(kept);
  }

 private:
  ContextPtr ctx_;
};

// The fusion stage of MappingAtTime (esvo_Mapping.cpp:341-395): dqvDepthPoints_.push_back(vdp) + window
// policy, then DepthFusion::update over the window (newest -> oldest) on a fresh DepthFrame,
// DepthMap::clean and DepthRegularization::apply.
class DepthFusion {
 public:
  explicit DepthFusion(ContextPtr ctx) : ctx_(std::move(ctx)) {}
  void pushFrame(const std::vector<DepthPoint>& vdp, const StampTransformationMap& st_map) {
    ctx_->check(esvo_map_push_frame(ctx_->handle(), vdp.data(), vdp.size(), st_map.T_world_virtual.data(), st_map.size()),
                "esvo_map_push_frame");
  }
  // esvo_MVStereo's PURE_BLOCK_MATCHING branch (esvo_MVStereo.cpp:411-423): vEMP2vDP, the window of maxNumFusionFrames_ frames
  // and DepthFusion::naive_propagation of every frame -- on the matches of EventBM::match_all_HyperThread
  void naivePropagation(const std::vector<EventMatchPair>& vEMP, const StampTransformationMap& st_map) {
    ctx_->check(esvo_map_fuse_matches_naive(ctx_->handle(), vEMP.data(), vEMP.size(), st_map.T_world_virtual.data(), st_map.size()),
                "esvo_map_fuse_matches_naive");
  }
  // returns numFusionCount
  size_t update() {
    size_t n = 0;
    ctx_->check(esvo_map_fuse(ctx_->handle(), &n), "esvo_map_fuse");
    return n;
  }
  // DepthMap iteration (SmartGrid::begin()/end()) in the reference's list order
  void getDepthMap(std::vector<DepthPoint>& out) {
    out.resize((size_t)ctx_->width() * ctx_->height());
    size_t n = 0;
    ctx_->check(esvo_map_get_depth_points(ctx_->handle(), out.data(), out.size(), &n), "esvo_map_get_depth_points");
    out.resize(n);
  }
  // publishPointCloud's payload (esvo_Mapping.cpp:925-932): float32 xyz in the world frame
  void getPointCloud(std::vector<float>& xyz) {
    xyz.resize((size_t)ctx_->width() * ctx_->height() * 3);
    size_t n = 0;
    ctx_->check(esvo_map_get_pointcloud_xyz(ctx_->handle(), xyz.data(), xyz.size() / 3, &n), "esvo_map_get_pointcloud_xyz");
    xyz.resize(n * 3);
  }

 private:
  ContextPtr ctx_;
};

// esvo_Mapping::createDenoisingMask + extractDenoisedEvents (esvo_Mapping.cpp:1046-1072) for the stage-wise
// path (esvo_map_tick applies them itself when params.denoising is set): binary map of the selected events
// at their RAW pixels -> 3x3 median (BORDER_REPLICATE) -> keep the events whose pixel is 255, in order.
inline void extractDenoisedEvents(const std::vector<Event>& vCloseEvents, std::vector<Event>& vEdgeEvents, int width, int height,
                                  size_t maxNum) {
  std::vector<uint8_t> map((size_t)width * height, 0);
  for (const Event& e : vCloseEvents)
    if (e.x < width && e.y < height) map[(size_t)e.y * width + e.x] = 1;
  vEdgeEvents.clear();
  for (const Event& e : vCloseEvents) {
    if (vEdgeEvents.size() >= maxNum) break;
    if (e.x >= width || e.y >= height) continue;
    int cnt = 0;
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) {
        const int yy = std::min(std::max((int)e.y + dy, 0), height - 1), xx = std::min(std::max((int)e.x + dx, 0), width - 1);
        cnt += map[(size_t)yy * width + xx];
      }
    if (cnt >= 5) vEdgeEvents.push_back(e);
  }
}

// esvo_Mapping::MappingAtTime on device-resident events / Time Surfaces (the fused path)
inline void MappingAtTime(Context& ctx, const StampedTimeSurfaceObs& obs, const StampTransformationMap& st_map) {
  ctx.check(esvo_map_set_observation(ctx.handle(), obs.t_ns, obs.TS_left, obs.TS_right, obs.T_world_cam),
            "esvo_map_set_observation");
  ctx.check(esvo_map_tick(ctx.handle(), obs.t_ns, st_map.stamps_ns.data(), st_map.T_world_virtual.data(), st_map.size()),
            "esvo_map_tick");
}

// esvo_MVStereo::MappingAtTime in MVStereoMode 1, PURE_BLOCK_MATCHING (esvo_MVStereo.cpp:383-432): block matching, vEMP2vDP and
// naive_propagation of the window instead of refinement and fusion
inline void MappingAtTimePureBlockMatching(Context& ctx, const StampedTimeSurfaceObs& obs, const StampTransformationMap& st_map) {
  ctx.check(esvo_map_set_observation(ctx.handle(), obs.t_ns, obs.TS_left, obs.TS_right, obs.T_world_cam),
            "esvo_map_set_observation");
  ctx.check(esvo_map_tick_bm_only(ctx.handle(), obs.t_ns, st_map.stamps_ns.data(), st_map.T_world_virtual.data(), st_map.size()),
            "esvo_map_tick_bm_only");
}

// ---- the tracker's optimiser: host C++ over the device's normal equations ----------------------------------------------------
// tools::cayley2rot (esvo_core/src/tools/cayley.cpp), row-major
inline void cayley2rot(const double c[3], double R[9]) {
  const double c0 = c[0], c1 = c[1], c2 = c[2];
  const double s = 1.0 + ((c0 * c0 + c1 * c1) + c2 * c2);
  const double M[9] = {1 + c0 * c0 - c1 * c1 - c2 * c2, 2 * (c0 * c1 - c2), 2 * (c0 * c2 + c1),
                       2 * (c0 * c1 + c2), 1 - c0 * c0 + c1 * c1 - c2 * c2, 2 * (c1 * c2 - c0),
                       2 * (c0 * c2 - c1), 2 * (c1 * c2 + c0), 1 - c0 * c0 - c1 * c1 + c2 * c2};
  for (int i = 0; i < 9; ++i) R[i] = M[i] / s;
}
// U V^T of the SVD of a (nearly orthonormal) 3x3 matrix = its orthogonal polar factor -- what addMotionUpdate's JacobiSVD
// re-orthonormalisation (RegProblemLM.cpp:355-357) returns; Newton's iteration X <- (X + X^-T) / 2 converges quadratically
inline void orthonormalize3(double X[9]) {
  for (int it = 0; it < 20; ++it) {
    const double* a = X;
    const double c00 = a[4] * a[8] - a[5] * a[7], c01 = a[5] * a[6] - a[3] * a[8], c02 = a[3] * a[7] - a[4] * a[6];
    const double c10 = a[2] * a[7] - a[1] * a[8], c11 = a[0] * a[8] - a[2] * a[6], c12 = a[1] * a[6] - a[0] * a[7];
    const double c20 = a[1] * a[5] - a[2] * a[4], c21 = a[2] * a[3] - a[0] * a[5], c22 = a[0] * a[4] - a[1] * a[3];
    const double det = (a[0] * c00 + a[1] * c01) + a[2] * c02;
    const double invT[9] = {c00 / det, c01 / det, c02 / det, c10 / det, c11 / det, c12 / det, c20 / det, c21 / det, c22 / det};  // X^-T
    double d = 0;
    for (int i = 0; i < 9; ++i) { const double n = 0.5 * (X[i] + invT[i]); d = std::max(d, std::fabs(n - X[i])); X[i] = n; }
    if (d < 1e-16) break;
  }
}
// solves A x = rhs (6 x 6, row-major) by Gaussian elimination with partial pivoting; false if singular
inline bool solve6(const double A_in[36], const double rhs[6], double x[6]) {
  double A[6][7];
  for (int i = 0; i < 6; ++i) { for (int j = 0; j < 6; ++j) A[i][j] = A_in[i * 6 + j]; A[i][6] = rhs[i]; }
  for (int c = 0; c < 6; ++c) {
    int piv = c;
    for (int r = c + 1; r < 6; ++r) if (std::fabs(A[r][c]) > std::fabs(A[piv][c])) piv = r;
    if (A[piv][c] == 0.0 || !std::isfinite(A[piv][c])) return false;
    if (piv != c) for (int j = 0; j < 7; ++j) std::swap(A[piv][j], A[c][j]);
    for (int r = c + 1; r < 6; ++r) {
      const double f = A[r][c] / A[c][c];
      for (int j = c; j < 7; ++j) A[r][j] -= f * A[c][j];
    }
  }
  for (int i = 5; i >= 0; --i) {
    double s = A[i][6];
    for (int j = i + 1; j < 6; ++j) s -= A[i][j] * x[j];
    x[i] = s / A[i][i];
  }
  return true;
}
struct Registration { double R[9]; double t[3]; double rms = 0; int iterations = 0; bool ok = true; };
// Levenberg-damped Gauss-Newton on x = (Cayley parameters, translation), linearised at x = 0 in every iteration exactly as
// RegProblemSolverLM::solve_analytical does (RegProblemSolverLM.cpp:160-183: x.fill(0), minimizeInit, ONE minimizeOneStep,
// addMotionUpdate):
//   (H + lambda diag(H) + 1e-9 I) dx = -b,   R' = orth(cayley2rot(dx_c) R),   t' = dx_t + cayley2rot(dx_c) t
// Like Eigen's minimizeOneStep, a trial step is ACCEPTED only if it pays: the cost is evaluated at the trial pose (same
// batch) and the step taken iff actual reduction / predicted reduction >= 1e-4, with
//   predicted = |f|^2 - |f + J dx|^2 = -(2 b.dx + dx.H dx);
// otherwise the damping is raised tenfold and the step recomputed (up to 6 times) -- the driver never moves to a pose with
// a higher cost.  After an accepted step the damping is lowered tenfold (never below `damping`) and carried into the next
// iteration (Marquardt's schedule; Eigen's minimizeInit restarts its trust region at every outer iteration and spends functor
// evaluations shrinking it again -- on the Time Surface's piecewise-constant image the useful damping is 1 .. 10, three
// rejections away from 1e-3, every time).
// The trials lambda, 10 lambda, 100 lambda of an iteration are SPECULATIVE: all three poses go to `normal_eq` in one call
// (one kernel launch, one workgroup per pose, one read-back on the device) and the first acceptable one in that order is
// taken -- the result is that of trying them one after the other, at the latency of one evaluation.
// It stops after max_iterations, when an accepted step is shorter than 1e-6, or when no damping up to 1e5 x the current one
// yields an acceptable step (the state in which Eigen's trust radius has shrunk below xtol |x|: its status 2 / 3, on which
// the reference's loop breaks, :181-182).  Stated deviation (DESIGN.md "Deviations"): the step is Levenberg's diagonal
// damping, not MINPACK's lmpar trust-region solve, so single steps differ from Eigen's while the fixed point is the same
// (bounded against the reference's loop on recorded cases: tests/test_track_normal.py, tests/golden/ref_track_solve.npz).
// `normal_eq(it, k, R[k][9], t[k][3], H[k][36], b[k][6], cost[k], &n)` evaluates H = J^T J, b = J^T f, cost = |f|^2 at k poses
// (1 <= k <= 3) on the batch of outer iteration `it`: esvo_track_normal_equations_batch on the device, or the CPU oracle's
// restatement in the tests.  same_batch: the batch does not depend on `it`, so the evaluation at an accepted trial pose IS
// the next iteration's linearisation.
constexpr int kRegisterTrials = 3;
template <class NormalEq>
Registration gauss_newton_register(NormalEq&& normal_eq, const double R0[9], const double t0[3], int max_iterations = 12,
                                   double damping = 1e-3, bool same_batch = true) {
  constexpr int K = kRegisterTrials;
  Registration g;
  for (int i = 0; i < 9; ++i) g.R[i] = R0[i];
  for (int i = 0; i < 3; ++i) g.t[i] = t0[i];
  double H[36], b[6], cost = 0, lambda = damping;
  size_t n = 0;
  bool have = false;
  for (int it = 0; it < max_iterations; ++it) {
    if (!have && !normal_eq(it, 1, g.R, g.t, H, b, &cost, &n)) { g.ok = false; return g; }
    have = false;
    g.iterations = it + 1;
    g.rms = n ? std::sqrt(cost / (double)n) : 0.0;
    double dx[K][6], Rn[K][9], tn[K][3], Ht[K][36], bt[K][6], cost_t[K];
    size_t nt = 0;
    int pick = -1;
    for (int round = 0; round < 2 && pick < 0; ++round) {  // dampings lambda 10^0..2, then lambda 10^3..5
      int k_eff = 0;
      double lam = lambda;
      for (int k = 0; k < K; ++k, lam *= 10.0) {
        double A[36], rhs[6];
        for (int i = 0; i < 36; ++i) A[i] = H[i];
        for (int i = 0; i < 6; ++i) { A[i * 6 + i] = (H[i * 6 + i] + lam * H[i * 6 + i]) + 1e-9; rhs[i] = -b[i]; }
        if (!solve6(A, rhs, dx[k])) break;
        double dR[9];
        cayley2rot(dx[k], dR);
        for (int r = 0; r < 3; ++r)
          for (int c = 0; c < 3; ++c)
            Rn[k][r * 3 + c] = (dR[r * 3 + 0] * g.R[0 * 3 + c] + dR[r * 3 + 1] * g.R[1 * 3 + c]) + dR[r * 3 + 2] * g.R[2 * 3 + c];
        orthonormalize3(Rn[k]);
        for (int r = 0; r < 3; ++r) tn[k][r] = dx[k][3 + r] + ((dR[r * 3 + 0] * g.t[0] + dR[r * 3 + 1] * g.t[1]) + dR[r * 3 + 2] * g.t[2]);
        k_eff = k + 1;
      }
      if (k_eff == 0 || !normal_eq(it, k_eff, &Rn[0][0], &tn[0][0], &Ht[0][0], &bt[0][0], cost_t, &nt)) { g.ok = false; return g; }
      lam = lambda;
      for (int k = 0; k < k_eff && pick < 0; ++k, lam *= 10.0) {
        double pred = 0;  // -(2 b.dx + dx.H dx)
        for (int i = 0; i < 6; ++i) {
          double hd = 0;
          for (int j = 0; j < 6; ++j) hd += H[i * 6 + j] * dx[k][j];
          pred -= dx[k][i] * (2.0 * b[i] + hd);
        }
        if (pred > 0 && (cost - cost_t[k]) >= 1e-4 * pred) { pick = k; lambda = lam; }
      }
      if (pick < 0) {
        if (k_eff < K) { g.ok = false; return g; }  // a singular damped system among the trials, none before it acceptable
        lambda = lam;  // = lambda x 10 x 10 x 10, the product a one-by-one loop arrives at
      }
    }
    if (pick < 0) break;  // no step pays any more: (R, t) stays the last accepted pose
    for (int i = 0; i < 9; ++i) g.R[i] = Rn[pick][i];
    for (int i = 0; i < 3; ++i) g.t[i] = tn[pick][i];
    lambda = lambda / 10.0 > damping ? lambda / 10.0 : damping;
    if (same_batch) {
      for (int i = 0; i < 36; ++i) H[i] = Ht[pick][i];
      for (int i = 0; i < 6; ++i) b[i] = bt[pick][i];
      cost = cost_t[pick]; n = nt; have = true;
      g.rms = n ? std::sqrt(cost / (double)n) : 0.0;
    }
    double nrm = 0;
    for (int i = 0; i < 6; ++i) nrm += dx[pick][i] * dx[pick][i];
    if (std::sqrt(nrm) < 1e-6) break;
  }
  return g;
}

// esvo_core::core::RegProblemLM's evaluation side (esvo_core/src/core/RegProblemLM.cpp): the per-point loops of
// setProblem (:44-56), operator() (:91-136) and df (:178-269) on the device.  The 6-DoF LM driver, the Cayley update and
// the SVD re-orthonormalisation (getWarpingTransformation / addMotionUpdate, :328-364) stay with the caller, who passes
// the resulting 4x4 warp (for operator()) or R_, t_ (for df) in row-major doubles.
struct RegProblemConfig {       // the fields of RegProblemConfig the evaluation reads (cfg/tracking/*.yaml)
  int kernelSize = 5;
  bool huber = true;            // LSnorm: "Huber" | "l2"
  double huber_threshold = 50.0;
  size_t BATCH_SIZE = 300;
  size_t MAX_REGISTRATION_POINTS = 2000;
};
class RegProblemLM {
 public:
  RegProblemLM(ContextPtr ctx, const RegProblemConfig& cfg) : ctx_(std::move(ctx)), cfg_(cfg) {}
  // setProblem: ref's point cloud (already in the stochastic order of :48-49) + the current frame's left Time Surface
  // (nullptr = the device-resident one of the last createTimeSurfaceAtTime of the left camera)
  void setProblem(const float* ref_xyz_world, size_t n_points, const double T_world_ref[16], const uint8_t* cur_TS_left) {
    numPoints_ = std::min(n_points, cfg_.MAX_REGISTRATION_POINTS);
    ctx_->check(esvo_track_set_reference(ctx_->handle(), ref_xyz_world, numPoints_, T_world_ref), "esvo_track_set_reference");
    ctx_->check(esvo_track_set_current(ctx_->handle(), cur_TS_left, cfg_.kernelSize), "esvo_track_set_current");
    numBatches_ = std::max(numPoints_ / cfg_.BATCH_SIZE, (size_t)1);
    setStochasticSampling(0, numPoints_);
  }
  void setStochasticSampling(size_t offset, size_t N) { offset_ = offset; count_ = N; }
  // operator(): fvec for the warp T_left_ref the caller derived from x; returns the number of values
  size_t operator()(const double T_left_ref[16], std::vector<double>& fvec) const {
    fvec.resize(count_);
    size_t n = 0;
    ctx_->check(esvo_track_residuals(ctx_->handle(), T_left_ref, offset_, count_, cfg_.huber ? ESVO_TRACK_HUBER : ESVO_TRACK_L2,
                                     cfg_.huber_threshold, fvec.data(), &n), "esvo_track_residuals");
    fvec.resize(n);
    return n;
  }
  // df at x = 0: fjac is n x 6, column-major (Eigen::MatrixXd layout)
  size_t df(const double R[9], const double t[3], std::vector<double>& fjac) const {
    fjac.resize(6 * count_);
    size_t n = 0;
    ctx_->check(esvo_track_jacobian(ctx_->handle(), R, t, offset_, count_, fjac.data(), &n), "esvo_track_jacobian");
    fjac.resize(6 * n);
    return n;
  }
  // F(0), df(0) and their products J^T J, J^T F, |F|^2 on the current batch in one device call
  size_t normalEquations(const double R[9], const double t[3], double H[36], double b[6], double* cost) const {
    size_t n = 0;
    ctx_->check(esvo_track_normal_equations(ctx_->handle(), R, t, offset_, count_, cfg_.huber ? ESVO_TRACK_HUBER : ESVO_TRACK_L2,
                                            cfg_.huber_threshold, H, b, cost, &n), "esvo_track_normal_equations");
    return n;
  }
  // ... at k poses (R: k x 9, t: k x 3, H: k x 36, b: k x 6, cost: k) in one launch
  size_t normalEquations(int k, const double* R, const double* t, double* H, double* b, double* cost) const {
    size_t n = 0;
    ctx_->check(esvo_track_normal_equations_batch(ctx_->handle(), k, R, t, offset_, count_, cfg_.huber ? ESVO_TRACK_HUBER : ESVO_TRACK_L2,
                                                  cfg_.huber_threshold, H, b, cost, &n), "esvo_track_normal_equations_batch");
    return n;
  }
  // RegProblemSolverLM::solve_analytical's loop (RegProblemSolverLM.cpp:148-215) with gauss_newton_register as the step:
  // the batch advances with the iteration as setStochasticSampling does there (:167-168)
  Registration solve(const double R0[9], const double t0[3], int MAX_ITERATION = 12, double damping = 1e-3) {
    const bool batches = cfg_.BATCH_SIZE < numPoints_;
    auto ne = [&](int it, int k, const double* R, const double* t, double* H, double* b, double* cost, size_t* n) {
      if (batches) setStochasticSampling(((size_t)it % numBatches_) * cfg_.BATCH_SIZE, cfg_.BATCH_SIZE);
      *n = normalEquations(k, R, t, H, b, cost);
      return true;
    };
    return gauss_newton_register(ne, R0, t0, MAX_ITERATION, damping, !batches);
  }
  size_t numBatches_ = 1, numPoints_ = 0;

 private:
  ContextPtr ctx_;
  RegProblemConfig cfg_;
  size_t offset_ = 0, count_ = 0;
};

}  // namespace esvo_hip
#endif  // ESVO_HIP_HPP
