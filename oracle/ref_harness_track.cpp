// ref_harness_track.cpp -- C entry points around the REFERENCE's own tracker functor (TEST INFRASTRUCTURE).
//
// oracle/_ref/libesvo_ref.so also contains the unmodified reference sources esvo_core/src/core/RegProblemLM.cpp,
// esvo_core/src/tools/cayley.cpp and esvo_core/src/container/ResidualItem.cpp, compiled where they lie under
// /root/reference against oracle/ref_shim/ (see ref_harness.cpp).  This file is glue only: it builds a RefFrame / CurFrame
// pair from plain arrays, calls RegProblemLM::setProblem / setStochasticSampling / operator() / df /
// getWarpingTransformation (RegProblemLM.cpp:26-351) and hands the results back, so that oracle/esvo_oracle.cpp's
// restatement of the tracker evaluation can be pinned to reference source
// (tests/golden/make_ref_fixtures.py, tests/test_ref_pin.py).
//
// What is NOT reference code here: the negated blurred Time Surface and its Sobel derivatives are OpenCV products
// (cv::GaussianBlur, cv::Sobel: absent); they are injected -- TS_left_ = 255 - negative with kernelSize 0 makes
// getTimeSurfaceNegative (TimeSurfaceObservation.h:118-135) reproduce the negative image exactly, the two derivative
// images are written into dTS_negative_du/dv_left_ directly.
#include <esvo_core/core/RegProblemLM.h>
#include <unsupported/Eigen/NonLinearOptimization>

#include <cstdlib>
#include <cstring>
#include <memory>

#include "../include/esvo_hip.h"

using namespace esvo_core;
using namespace esvo_core::core;
using namespace esvo_core::container;

namespace esvo_ref_shim {
struct Injected;
}

struct ref_tracker {
  int W = 0, H = 0;
  CameraSystem::Ptr camSys;
  RegProblemConfig::Ptr cfg;
  std::unique_ptr<RegProblemLM> prob;
  std::vector<pcl::PointXYZ> pts;
  RefFrame ref;
  CurFrame cur;
  TimeSurfaceObservation obs;
};

extern "C" {
// the CameraSystem comes from the mapper handle's factory (same injected calibration): see ref_harness.cpp
void* ref_make_camera_system(const char* calib_dir, const esvo_calib_t* left, const esvo_calib_t* right);

ref_tracker* ref_tracker_create(const char* calib_dir, const esvo_calib_t* left, const esvo_calib_t* right, int huber,
                                double huber_threshold, size_t max_points) {
  ref_tracker* h = new ref_tracker;
  h->W = left->width; h->H = left->height;
  h->camSys = *reinterpret_cast<CameraSystem::Ptr*>(ref_make_camera_system(calib_dir, left, right));
  // patch 1x1, kernelSize 0 (the negative image is injected, see the header), every shipped tracking yaml otherwise
  h->cfg = std::make_shared<RegProblemConfig>(1, 1, 0, huber ? "Huber" : "l2", huber_threshold, 0.2, 2.0, 1000, max_points, 200, 10);
  h->prob.reset(new RegProblemLM(h->camSys, h->cfg, 1));
  return h;
}
void ref_tracker_destroy(ref_tracker* h) { delete h; }

// neg: W*H u8 (TS_negative_left_), du / dv: W*H i16 (its Sobel derivatives); xyz_world: n points; poses 4x4 row-major.
// RegProblemLM::setProblem shuffles the points with rand() (:46-49): seed makes it repeatable, order_out (n) receives
// the index of the input point that ends up at each position.
void ref_tracker_set_problem(ref_tracker* h, const uint8_t* neg, const int16_t* du, const int16_t* dv, const float* xyz_world,
                             size_t n, const double T_world_ref[16], const double T_world_left[16], unsigned seed,
                             uint32_t* order_out) {
  const int W = h->W, H = h->H;
  h->obs = TimeSurfaceObservation();
  h->obs.TS_left_.resize(H, W);
  h->obs.TS_right_.resize(H, W);
  h->obs.dTS_negative_du_left_.resize(H, W);
  h->obs.dTS_negative_dv_left_.resize(H, W);
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      h->obs.TS_left_(y, x) = 255.0 - (double)neg[y * W + x];
      h->obs.TS_right_(y, x) = 0.0;
      h->obs.dTS_negative_du_left_(y, x) = (double)du[y * W + x];
      h->obs.dTS_negative_dv_left_(y, x) = (double)dv[y * W + x];
    }
  auto toT = [](const double* T) {
    Eigen::Matrix<double, 4, 4> M;
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) M(i, j) = T[i * 4 + j];
    return Transformation(M);
  };
  h->pts.resize(n);
  h->ref.vPointXYZPtr_.resize(n);
  for (size_t i = 0; i < n; ++i) {
    h->pts[i].x = xyz_world[3 * i]; h->pts[i].y = xyz_world[3 * i + 1]; h->pts[i].z = xyz_world[3 * i + 2];
    h->ref.vPointXYZPtr_[i] = &h->pts[i];
  }
  h->ref.tr_ = toT(T_world_ref);
  h->cur.tr_ = toT(T_world_left);
  h->cur.pTsObs_ = &h->obs;
  srand(seed);
  h->prob->setProblem(&h->ref, &h->cur, false);
  for (size_t i = 0; i < n; ++i) order_out[i] = (uint32_t)(h->ref.vPointXYZPtr_[i] - h->pts.data());
}
size_t ref_tracker_num_points(ref_tracker* h) { return h->prob->ResItems_.size(); }
void ref_tracker_relative_pose(ref_tracker* h, double R[9], double t[3]) {
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) R[i * 3 + j] = h->prob->R_(i, j);
    t[i] = h->prob->t_(i);
  }
}
// setStochasticSampling(offset, N) (:71-89), then operator() at x (:91-136); T_warping_out: getWarpingTransformation(x)
size_t ref_tracker_residuals(ref_tracker* h, size_t offset, size_t count, const double x[6], double* fvec_out,
                             double T_warping_out[16]) {
  h->prob->setStochasticSampling(offset, count);
  Eigen::Matrix<double, 6, 1> xv;
  for (int i = 0; i < 6; ++i) xv(i) = x[i];
  Eigen::VectorXd fvec(h->prob->values());
  (*h->prob)(xv, fvec);
  for (int i = 0; i < fvec.size(); ++i) fvec_out[i] = fvec(i);
  Eigen::Matrix4d Tw = Eigen::Matrix4d::Identity();
  h->prob->getWarpingTransformation(Tw, xv);
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) T_warping_out[i * 4 + j] = Tw(i, j);
  return (size_t)fvec.size();
}
// df at x = 0 (:178-269); out: column-major n x 6, as Eigen stores fjac
size_t ref_tracker_jacobian(ref_tracker* h, size_t offset, size_t count, double* fjac_out) {
  h->prob->setStochasticSampling(offset, count);
  Eigen::MatrixXd J;
  h->prob->df(Eigen::Matrix<double, 6, 1>::Zero(), J);
  for (int j = 0; j < 6; ++j)
    for (int i = 0; i < J.rows(); ++i) fjac_out[(size_t)j * J.rows() + i] = J(i, j);
  return (size_t)J.rows();
}
// The reference's tracker loop, RegProblemSolverLM::solve_analytical (RegProblemSolverLM.cpp:148-178, 212): the solver CLASS is not
// compiled here (it drags the visualisation publishers in), so its twenty lines are spelled out again around the reference's own
// functor -- LevenbergMarquardt with ftol = xtol = 1e-3 and maxfev = 8 MAX_ITERATION, and per outer iteration: the next
// batch (setStochasticSampling), x = 0, minimizeInit, ONE minimizeOneStep, addMotionUpdate(x); it leaves on status 2 / 3.
// The LM class is ref_shim's restatement of MINPACK lmder / lmpar (unsupported/Eigen/NonLinearOptimization there), i.e. Eigen's
// trust-region semantics, not the library.  Call after ref_tracker_set_problem; batch_size / max_iteration are the yaml's
// BATCH_SIZE / MAX_ITERATION.  Outputs: R_, t_ after the loop (what setPose composes with T_world_ref), the outer iterations,
// the functor evaluations, the last status.
int ref_tracker_solve(ref_tracker* h, size_t batch_size, size_t max_iteration, double R[9], double t[3], size_t* iterations,
                      size_t* nfev_out, int* last_status) {
  RegProblemLM& prob = *h->prob;
  h->cfg->BATCH_SIZE_ = batch_size;
  h->cfg->MAX_ITERATION_ = max_iteration;
  prob.numBatches_ = std::max(prob.ResItems_.size() / batch_size, (size_t)1);
  Eigen::LevenbergMarquardt<RegProblemLM, double> lm(prob);
  lm.resetParameters();
  lm.parameters.ftol = 1e-3;
  lm.parameters.xtol = 1e-3;
  lm.parameters.maxfev = max_iteration * 8;
  size_t iteration = 0, nfev = 0;
  int status = -2;
  while (iteration < max_iteration) {
    prob.setStochasticSampling((iteration % prob.numBatches_) * batch_size, batch_size);
    Eigen::VectorXd x(6);
    x.setZero();  // x.fill(0.0) upstream
    if (lm.minimizeInit(x) == Eigen::LevenbergMarquardtSpace::ImproperInputParameters) return -1;
    status = (int)lm.minimizeOneStep(x);
    prob.addMotionUpdate(x);
    iteration++;
    nfev += lm.nfev;
    if (status == 2 || status == 3) break;
  }
  prob.setPose();
  ref_tracker_relative_pose(h, R, t);
  *iterations = iteration;
  *nfev_out = nfev;
  *last_status = status;
  return 0;
}
}  // extern "C"
