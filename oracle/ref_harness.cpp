// ref_harness.cpp -- C entry points around the REFERENCE's own mapper classes (TEST INFRASTRUCTURE).
//
// oracle/_ref/libesvo_ref.so = this file + the unmodified reference sources
//   esvo_core/src/container/{DepthPoint,CameraSystem}.cpp, esvo_core/src/core/{EventBM,DepthProblem,
//   DepthProblemSolver,DepthFusion,DepthRegularization}.cpp, esvo_core/src/tools/sobel.cpp (+ SmartGrid.h,
//   DepthMap.h, utils.h, TimeSurfaceObservation.h, ... through their includes)
// compiled where they lie under /root/reference against the stand-in headers of oracle/ref_shim/ (Eigen, OpenCV,
// ROS, kindr, glog, yaml-cpp, pcl are absent from this image).  Nothing of the reference is copied into this
// repository; the library exists only in this container (oracle/_ref/ is git-ignored) and is used to pin
// oracle/esvo_oracle.cpp: tests/golden/make_ref_fixtures.py records its outputs as fixtures and
// tests/test_ref_pin.py checks the oracle against them (and against the live library when it is present).
//
// What this file itself contains is glue only: POD <-> reference types, and the call sequence of
// esvo_Mapping::MappingAtTime (esvo_Mapping.cpp:261-431) around the reference's classes -- the node file itself
// needs a ROS runtime and is not compiled.  One definition is applied where the reference is undefined
// (SURVEY.md Appendix A-7): after DepthMap::clean, grid cells that still point at an erased element are set to NULL
// ("the cell reads empty", as the oracle defines it) and counted; ref_mapper_counters reports how often it happened.
#define private public  // SmartGrid's grid is private; the harness needs to see dangling cells (layout unchanged)
#include <esvo_core/container/SmartGrid.h>
#undef private
#include <esvo_core/container/CameraSystem.h>
#include <esvo_core/container/DepthMap.h>
#include <esvo_core/core/DepthFusion.h>
#include <esvo_core/core/DepthProblem.h>
#include <esvo_core/core/DepthProblemSolver.h>
#include <esvo_core/core/DepthRegularization.h>
#include <esvo_core/core/EventBM.h>

#include <cstdio>
#include <deque>
#include <set>
#include <unordered_map>

#include "../include/esvo_hip.h"

using namespace esvo_core;
using namespace esvo_core::core;
using namespace esvo_core::container;

namespace {
Eigen::Matrix<double, 4, 4> mat4(const double* T) {
  Eigen::Matrix<double, 4, 4> M;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) M(i, j) = T[i * 4 + j];
  return M;
}
}  // namespace

struct ref_mapper {
  esvo_params_t prm;
  CameraSystem::Ptr camSys;
  std::shared_ptr<DepthProblemConfig> dpConfig;
  std::unique_ptr<EventBM> ebm;
  std::unique_ptr<DepthProblemSolver> solver;
  std::unique_ptr<DepthFusion> fusor;
  std::unique_ptr<DepthRegularization> regularizor;
  // TS_obs_ and st_map_
  StampedTimeSurfaceObs obs;
  StampTransformationMap st_map;
  std::vector<uint64_t> pose_t;
  std::vector<Eigen::Matrix<double, 4, 4>> pose_T;
  // dqvDepthPoints_ and depthFramePtr_
  std::deque<std::vector<DepthPoint>> window;
  DepthFrame::Ptr frame;
  std::vector<DepthPoint> last_vdp;
  uint64_t n_dangling = 0;
  int W = 0, H = 0;

  void build() {
    const char* norm = prm.ls_norm == ESVO_LSNORM_TDIST ? "Tdist" : "l2";
    dpConfig = std::make_shared<DepthProblemConfig>(prm.patch_size_x, prm.patch_size_y, norm, prm.td_nu, prm.td_scale,
                                                    prm.lm_max_iteration, prm.reg_radius, prm.reg_min_neighbours,
                                                    prm.reg_min_close_neighbours);
    // esvo_Mapping.cpp:44-52,127-128: EventBM(camSys, NUM_THREAD_MAPPING, bSmoothTS, ...) + resetParameters.
    // bSmoothTS is false here: the harness receives the already-smoothed observation (OpenCV's GaussianBlur
    // is not part of this build), see ref_mapper_set_observation.
    ebm.reset(new EventBM(camSys, (size_t)prm.num_threads, false, prm.patch_size_x, prm.patch_size_y,
                          prm.bm_min_disparity, prm.bm_max_disparity, prm.bm_step, prm.bm_zncc_threshold,
                          prm.bm_updown != 0));
    solver.reset(new DepthProblemSolver(camSys, dpConfig, NUMERICAL, (size_t)prm.num_threads));
    fusor.reset(new DepthFusion(camSys, dpConfig));
    regularizor.reset(new DepthRegularization(dpConfig));
  }
  uint32_t pose_index(const Eigen::Matrix<double, 4, 4>& T) const {
    for (size_t k = 0; k < pose_T.size(); ++k) {
      bool same = true;
      for (int i = 0; i < 4 && same; ++i)
        for (int j = 0; j < 4; ++j)
          if (pose_T[k](i, j) != T(i, j)) { same = false; break; }
      if (same) return (uint32_t)k;
    }
    return 0xffffffffu;
  }
  void to_pod(const DepthPoint& d, esvo_depth_point_t& o, uint32_t seq) const {
    o.row = (uint32_t)d.row(); o.col = (uint32_t)d.col();
    o.x[0] = d.x()(0); o.x[1] = d.x()(1);
    o.inv_depth = d.invDepth(); o.scale2 = d.scaleSquared(); o.nu = d.nu(); o.variance = d.variance();
    o.residual = d.residual(); o.age = d.age();
    for (int k = 0; k < 3; ++k) o.p_cam[k] = d.p_cam()(k);
    o.pose_idx = pose_index(d.T_world_cam());
    o.seq = seq;
  }
};

extern "C" {

// calib_dir holds left.yaml / right.yaml in the reference's format (CameraSystem::loadCalibInfo reads them);
// left/right carry the OpenCV-derived products (LUT, mask) that the stand-in cv:: functions hand to
// PerspectiveCamera::preComputeRectifiedCoordinate.
ref_mapper* ref_mapper_create(const char* calib_dir, const esvo_params_t* p, const esvo_calib_t* left,
                              const esvo_calib_t* right) {
  ref_mapper* h = new ref_mapper;
  h->prm = *p;
  h->W = left->width; h->H = left->height;
  esvo_ref_shim::Injected& g = esvo_ref_shim::inject();
  g.W = h->W; g.H = h->H; g.cam = 0;
  g.lut[0] = left->rect_lut; g.lut[1] = right->rect_lut;
  g.mask[0] = left->rect_mask; g.mask[1] = right->rect_mask;
  h->camSys = std::make_shared<CameraSystem>(std::string(calib_dir), false);
  h->build();
  h->frame = std::make_shared<DepthFrame>(h->H, h->W);
  return h;
}
// the same CameraSystem factory for the tracker harness (ref_harness_track.cpp); the returned pointer owns a shared_ptr
void* ref_make_camera_system(const char* calib_dir, const esvo_calib_t* left, const esvo_calib_t* right) {
  esvo_ref_shim::Injected& g = esvo_ref_shim::inject();
  g.W = left->width; g.H = left->height; g.cam = 0;
  g.lut[0] = left->rect_lut; g.lut[1] = right->rect_lut;
  g.mask[0] = left->rect_mask; g.mask[1] = right->rect_mask;
  return new CameraSystem::Ptr(std::make_shared<CameraSystem>(std::string(calib_dir), false));
}
void ref_mapper_destroy(ref_mapper* h) { delete h; }
void ref_mapper_reset(ref_mapper* h) { h->window.clear(); h->frame = std::make_shared<DepthFrame>(h->H, h->W); }
void ref_mapper_set_params(ref_mapper* h, const esvo_params_t* p) { h->prm = *p; h->build(); }
double ref_mapper_baseline(ref_mapper* h) { return h->camSys->baseline_; }

void ref_cam2world(ref_mapper* h, const double x[2], double inv_depth, double p[3]) {
  Eigen::Vector3d out;
  h->camSys->cam_left_ptr_->cam2World(Eigen::Vector2d(x[0], x[1]), inv_depth, out);
  for (int k = 0; k < 3; ++k) p[k] = out(k);
}
void ref_world2cam(ref_mapper* h, int right, const double p[3], double x[2]) {
  Eigen::Vector2d out;
  (right ? h->camSys->cam_right_ptr_ : h->camSys->cam_left_ptr_)->world2Cam(Eigen::Vector3d(p[0], p[1], p[2]), out);
  x[0] = out(0); x[1] = out(1);
}
// the calibration products as the reference's PerspectiveCamera stores them (round trip of the injection)
void ref_get_lut_mask(ref_mapper* h, double* lut_xy, int32_t* mask) {
  auto& cam = *h->camSys->cam_left_ptr_;
  for (int y = 0; y < h->H; ++y)
    for (int x = 0; x < h->W; ++x) {
      Eigen::Matrix<double, 2, 1> v = cam.getRectifiedUndistortedCoordinate(x, y);
      lut_xy[2 * (y * h->W + x)] = v(0);
      lut_xy[2 * (y * h->W + x) + 1] = v(1);
      mask[y * h->W + x] = cam.UndistortRectify_mask_(y, x);
    }
}

// TS_obs_ (TimeSurfaceObservation ctor, TimeSurfaceObservation.h:29-56): mono8 images -> MatrixXd
void ref_mapper_set_observation(ref_mapper* h, uint64_t t_ns, const uint8_t* ts_left, const uint8_t* ts_right,
                                const double T_world_cam[16]) {
  cv_bridge::CvImagePtr l = std::make_shared<cv_bridge::CvImage>(), r = std::make_shared<cv_bridge::CvImage>();
  l->image = cv::Mat(h->H, h->W, CV_8U);
  r->image = cv::Mat(h->H, h->W, CV_8U);
  for (size_t i = 0; i < (size_t)h->W * h->H; ++i) { l->image.v[i] = ts_left[i]; r->image.v[i] = ts_right[i]; }
  Transformation tr(mat4(T_world_cam));
  h->obs = StampedTimeSurfaceObs(ros::Time((uint32_t)(t_ns / 1000000000ull), (uint32_t)(t_ns % 1000000000ull)),
                                 TimeSurfaceObservation(l, r, tr, 0, false));
}
void ref_mapper_set_poses(ref_mapper* h, const uint64_t* t_ns, const double* T, size_t m) {
  h->st_map.clear();
  h->pose_t.assign(t_ns, t_ns + m);
  h->pose_T.clear();
  for (size_t i = 0; i < m; ++i) {
    h->pose_T.push_back(mat4(T + 16 * i));
    h->st_map.emplace(ros::Time((uint32_t)(t_ns[i] / 1000000000ull), (uint32_t)(t_ns[i] % 1000000000ull)),
                      Transformation(h->pose_T.back()));
  }
}

static void run_match(ref_mapper* h, const esvo_event_t* ev, size_t n, std::vector<EventMatchPair>& vEMP) {
  std::vector<dvs_msgs::Event> events(n);
  std::vector<dvs_msgs::Event*> ptrs(n);
  for (size_t i = 0; i < n; ++i) {
    events[i].x = ev[i].x; events[i].y = ev[i].y;
    events[i].ts = ros::Time(ev[i].sec, ev[i].nsec);
    events[i].polarity = ev[i].polarity;
    ptrs[i] = &events[i];
  }
  h->ebm->createMatchProblem(&h->obs, &h->st_map, &ptrs);  // esvo_Mapping.cpp:307
  h->ebm->match_all_HyperThread(vEMP);                     // :308
}

size_t ref_mapper_match(ref_mapper* h, const esvo_event_t* ev, size_t n, esvo_match_t* out, size_t cap) {
  std::vector<EventMatchPair> vEMP;
  run_match(h, ev, n, vEMP);
  // event index: first event with the same raw pixel and stamp (EventMatchPair keeps no index)
  std::unordered_map<uint64_t, std::vector<uint32_t>> by_key;
  auto key = [](uint32_t x, uint32_t y, uint64_t t) { return (t * 1315423911ull) ^ ((uint64_t)x << 16) ^ y; };
  for (size_t i = 0; i < n; ++i)
    by_key[key(ev[i].x, ev[i].y, (uint64_t)ev[i].sec * 1000000000ull + ev[i].nsec)].push_back((uint32_t)i);
  for (size_t k = 0; k < vEMP.size() && k < cap; ++k) {
    const EventMatchPair& m = vEMP[k];
    esvo_match_t& o = out[k];
    o.x_left[0] = m.x_left_(0); o.x_left[1] = m.x_left_(1);
    o.inv_depth = m.invDepth_; o.cost = m.cost_; o.disp = m.disp_;
    o.event_idx = 0xffffffffu;
    auto it = by_key.find(key((uint32_t)m.x_left_raw_(0), (uint32_t)m.x_left_raw_(1), m.t_.toNSec()));
    if (it != by_key.end())
      for (uint32_t i : it->second)
        if (ev[i].x == (uint16_t)m.x_left_raw_(0) && ev[i].y == (uint16_t)m.x_left_raw_(1) && ev[i].sec == m.t_.sec &&
            ev[i].nsec == m.t_.nsec) { o.event_idx = i; break; }
    o.pose_idx = h->pose_index(m.trans_.getTransformationMatrix());
  }
  return vEMP.size();
}

static void run_refine(ref_mapper* h, const esvo_match_t* matches, size_t n, int cull, std::vector<DepthPoint>& vdp) {
  std::vector<EventMatchPair> vEMP(n);
  for (size_t i = 0; i < n; ++i) {
    vEMP[i].x_left_ = Eigen::Vector2d(matches[i].x_left[0], matches[i].x_left[1]);
    vEMP[i].trans_ = Transformation(h->pose_T[matches[i].pose_idx]);
    vEMP[i].invDepth_ = matches[i].inv_depth;
    vEMP[i].cost_ = matches[i].cost;
    vEMP[i].disp_ = matches[i].disp;
  }
  h->solver->solve(&vEMP, &h->obs, vdp);  // esvo_Mapping.cpp:329
  if (cull)                                // :333-334, cost threshold esvo_Mapping.cpp:97
    h->solver->pointCulling(vdp, h->prm.stdvar_vis_threshold,
                            pow(h->prm.residual_vis_threshold, 2) * h->prm.patch_size_x * h->prm.patch_size_y,
                            h->prm.invdepth_min, h->prm.invdepth_max);
}

size_t ref_mapper_refine(ref_mapper* h, const esvo_match_t* matches, size_t n, int cull, esvo_depth_point_t* out,
                         size_t cap) {
  std::vector<DepthPoint> vdp;
  run_refine(h, matches, n, cull, vdp);
  for (size_t i = 0; i < vdp.size() && i < cap; ++i) h->to_pod(vdp[i], out[i], (uint32_t)i);
  return vdp.size();
}

// DepthProblem::operator() at one inverse depth (DepthProblem.cpp:34-160)
int ref_mapper_eval_residual(ref_mapper* h, const double x_left[2], uint32_t pose_idx, double rho, double* fvec) {
  DepthProblem prob(h->dpConfig, h->camSys);
  Eigen::Vector2d coor(x_left[0], x_left[1]);
  Eigen::Matrix<double, 4, 4> T = h->pose_T[pose_idx];
  prob.setProblem(coor, T, &h->obs);
  Eigen::VectorXd x(1), f(prob.values());
  x << rho;
  int ret = prob(x, f);
  for (int i = 0; i < prob.values(); ++i) fvec[i] = f[i];
  return ret;
}
// DepthProblemSolver::solve_single_problem_numerical on one match (DepthProblemSolver.cpp:138-214):
// result = {inverse depth, variance, residual}; returns 1 when the problem counts as solved
int ref_mapper_solve_single(ref_mapper* h, const double x_left[2], uint32_t pose_idx, double d_init, double result[3]) {
  auto prob = std::make_shared<Eigen::NumericalDiff<DepthProblem>>(h->dpConfig, h->camSys);
  Eigen::Vector2d coor(x_left[0], x_left[1]);
  Eigen::Matrix<double, 4, 4> T = h->pose_T[pose_idx];
  prob->setProblem(coor, T, &h->obs);
  return h->solver->solve_single_problem_numerical(d_init, prob, result) ? 1 : 0;
}
// EventBM::zncc_cost on two wy x wx patches (row-major doubles), EventBM.cpp:317-333
double ref_zncc_cost(const double* l, const double* r, int wx, int wy) {
  Eigen::MatrixXd L(wy, wx), R(wy, wx);
  for (int y = 0; y < wy; ++y)
    for (int x = 0; x < wx; ++x) { L(y, x) = l[y * wx + x]; R(y, x) = r[y * wx + x]; }
  return EventBM::zncc_cost(L, R, false);
}

static std::vector<DepthPoint> from_pod(ref_mapper* h, const esvo_depth_point_t* pts, size_t n, const double* pose_T,
                                        size_t m) {
  std::vector<DepthPoint> v;
  v.reserve(n);
  for (size_t i = 0; i < n; ++i) {
    DepthPoint dp(pts[i].row, pts[i].col);
    dp.update_x(Eigen::Vector2d(pts[i].x[0], pts[i].x[1]));
    dp.invDepth() = pts[i].inv_depth; dp.scaleSquared() = pts[i].scale2; dp.nu() = pts[i].nu;
    dp.variance() = pts[i].variance; dp.residual() = pts[i].residual; dp.age() = (size_t)pts[i].age;
    dp.update_p_cam(Eigen::Vector3d(pts[i].p_cam[0], pts[i].p_cam[1], pts[i].p_cam[2]));
    Eigen::Matrix<double, 4, 4> T = mat4(pose_T + 16 * (pts[i].pose_idx < m ? pts[i].pose_idx : 0));
    dp.updatePose(T);
    v.push_back(dp);
  }
  (void)h;
  return v;
}

static void window_policy(ref_mapper* h, std::vector<DepthPoint>& vdp) {  // esvo_Mapping.cpp:341-368
  if (h->prm.fusion_strategy == ESVO_FUSION_CONST_POINTS) {
    size_t numFusionPoints = 0;
    h->window.push_back(vdp);
    for (size_t n = 0; n < h->window.size(); n++) numFusionPoints += h->window[n].size();
    while (numFusionPoints > 1.5 * h->prm.max_fusion_points) {
      h->window.pop_front();
      numFusionPoints = 0;
      for (size_t n = 0; n < h->window.size(); n++) numFusionPoints += h->window[n].size();
    }
  } else {
    h->window.push_back(vdp);
    while (h->window.size() > (size_t)h->prm.max_fusion_frames) h->window.pop_front();
  }
}

void ref_mapper_push_frame(ref_mapper* h, const esvo_depth_point_t* pts, size_t n, const double* pose_T, size_t m) {
  std::vector<DepthPoint> vdp = from_pod(h, pts, n, pose_T, m);
  h->last_vdp = vdp;
  window_policy(h, vdp);
}

size_t ref_mapper_fuse(ref_mapper* h) {
  // esvo_Mapping.cpp:268-272: a new DepthFrame at the TS pose
  h->frame = std::make_shared<DepthFrame>(h->H, h->W);
  h->frame->setTransformation(h->obs.second.tr_);
  size_t numFusionCount = 0;
  for (auto it = h->window.rbegin(); it != h->window.rend(); it++)  // :372-377
    numFusionCount += h->fusor->update(*it, h->frame, h->prm.fusion_radius);
  const bool do_clean = h->prm.clean_requires_full_window ? h->window.size() >= (size_t)h->prm.max_fusion_frames : true;
  if (do_clean) {  // :385-386 (esvo_MVStereo.cpp:496-497 cleans always)
    h->frame->dMap_->clean(pow(h->prm.stdvar_vis_threshold, 2), h->prm.age_vis_threshold, h->prm.invdepth_max,
                           h->prm.invdepth_min);
    // Appendix A-7: cells that still point at an erased element are undefined behaviour upstream; define: empty
    DepthMap& dm = *h->frame->dMap_;
    std::set<const DepthPoint*> live;
    for (auto it = dm.begin(); it != dm.end(); ++it) live.insert(&*it);
    for (size_t r = 0; r < dm.rows(); ++r)
      for (size_t c = 0; c < dm.cols(); ++c) {
        DepthPoint*& cell = (*dm._grid[r])[c];
        if (cell && !live.count(cell)) { cell = NULL; h->n_dangling++; }
      }
  }
  if (h->prm.regularization) h->regularizor->apply(h->frame->dMap_);  // :390-395
  return numFusionCount;
}

// InitializationAtTime after the StereoSGBM call (esvo_Mapping.cpp:433-492): the disparity image is an OpenCV product and
// comes in (disp16: W*H int16, disparity * 16); xy: the (x, y) pairs createEdgeMask (:1000-1044, radius 0) derives from
// the SGM events.  Glue for :455-480 (DepthPoint(x, y) with the arguments in the order the node passes them, update_x,
// cam2World, the Gaussian update, residual, age, pose), then the reference's DepthFusion::naive_propagation
// (DepthFusion.cpp:234-288).  Returns the number of SGM depth points; 0 (and no change) below min_points.
size_t ref_mapper_init_from_disparity(ref_mapper* h, const int16_t* disp16, const uint32_t* xy, size_t n, size_t min_points) {
  h->frame = std::make_shared<DepthFrame>(h->H, h->W);
  h->frame->setTransformation(h->obs.second.tr_);
  std::vector<DepthPoint> vdp_sgm;
  vdp_sgm.reserve(n);
  const double var_SGM = pow(0.001, 2);
  for (size_t i = 0; i < n; i++) {
    const size_t x = xy[2 * i], y = xy[2 * i + 1];
    const double disp = disp16[y * (size_t)h->W + x] / 16.0;
    if (disp < 0) continue;
    DepthPoint dp(x, y);
    Eigen::Vector2d p_img(x * 1.0, y * 1.0);
    dp.update_x(p_img);
    const double invDepth = disp / (h->camSys->cam_left_ptr_->P_(0, 0) * h->camSys->baseline_);
    if (invDepth < h->prm.invdepth_min || invDepth > h->prm.invdepth_max) continue;
    Eigen::Vector3d p_cam;
    h->camSys->cam_left_ptr_->cam2World(p_img, invDepth, p_cam);
    dp.update_p_cam(p_cam);
    dp.update(invDepth, var_SGM);
    dp.residual() = 0.0;
    dp.age() = h->prm.age_vis_threshold;
    Eigen::Matrix<double, 4, 4> T_world_cam = h->obs.second.tr_.getTransformationMatrix();
    dp.updatePose(T_world_cam);
    vdp_sgm.push_back(dp);
  }
  if (vdp_sgm.size() < min_points) return 0;
  h->window.push_back(vdp_sgm);
  h->last_vdp = vdp_sgm;
  h->fusor->naive_propagation(vdp_sgm, h->frame);
  return vdp_sgm.size();
}

// MappingAtTime on already selected (and denoised) events, esvo_Mapping.cpp:261-431
size_t ref_mapper_tick(ref_mapper* h, const esvo_event_t* ev, size_t n) {
  std::vector<EventMatchPair> vEMP;
  run_match(h, ev, n, vEMP);
  std::vector<DepthPoint> vdp;
  vdp.reserve(vEMP.size());
  h->solver->solve(&vEMP, &h->obs, vdp);
  h->solver->pointCulling(vdp, h->prm.stdvar_vis_threshold,
                          pow(h->prm.residual_vis_threshold, 2) * h->prm.patch_size_x * h->prm.patch_size_y,
                          h->prm.invdepth_min, h->prm.invdepth_max);
  h->last_vdp = vdp;
  window_policy(h, vdp);
  return ref_mapper_fuse(h);
}

size_t ref_mapper_map_size(ref_mapper* h) { return h->frame->dMap_->size(); }
size_t ref_mapper_get_map(ref_mapper* h, esvo_depth_point_t* out, size_t cap) {
  size_t k = 0;
  for (auto it = h->frame->dMap_->begin(); it != h->frame->dMap_->end(); ++it, ++k)
    if (k < cap) h->to_pod(*it, out[k], (uint32_t)k);
  return k;
}
// true grid cell (row*W+col) of each element in list order, -1 when no cell points at it
size_t ref_mapper_get_map_cells(ref_mapper* h, int32_t* out, size_t cap) {
  DepthMap& dm = *h->frame->dMap_;
  std::unordered_map<const DepthPoint*, int32_t> cell_of;
  for (size_t r = 0; r < dm.rows(); ++r)
    for (size_t c = 0; c < dm.cols(); ++c)
      if ((*dm._grid[r])[c]) cell_of[(*dm._grid[r])[c]] = (int32_t)(r * dm.cols() + c);
  size_t k = 0;
  for (auto it = dm.begin(); it != dm.end(); ++it, ++k)
    if (k < cap) {
      auto f = cell_of.find(&*it);
      out[k] = f == cell_of.end() ? -1 : f->second;
    }
  return k;
}
size_t ref_mapper_get_last_frame(ref_mapper* h, esvo_depth_point_t* out, size_t cap) {
  for (size_t i = 0; i < h->last_vdp.size() && i < cap; ++i) h->to_pod(h->last_vdp[i], out[i], (uint32_t)i);
  return h->last_vdp.size();
}
void ref_mapper_counters(ref_mapper* h, uint64_t out[8]) {
  size_t np = 0;
  for (auto& f : h->window) np += f.size();
  out[0] = h->window.size(); out[1] = np; out[2] = h->n_dangling;
  for (int i = 3; i < 8; ++i) out[i] = 0;
}
// unit hooks on DepthPoint (DepthPoint.cpp:146-188)
void ref_update_student_t(double state[5], double inv_depth, double scale2, double variance, double nu) {
  // state = {invDepth, scale2, nu, variance, age}
  DepthPoint dp(0, 0);
  if (state[0] > -1e-6) {
    dp.invDepth() = state[0]; dp.scaleSquared() = state[1]; dp.nu() = state[2]; dp.variance() = state[3];
    dp.age() = (size_t)state[4];
  }
  dp.update_studentT(inv_depth, scale2, variance, nu);
  state[0] = dp.invDepth(); state[1] = dp.scaleSquared(); state[2] = dp.nu(); state[3] = dp.variance();
  state[4] = (double)dp.age();
}
}  // extern "C"
