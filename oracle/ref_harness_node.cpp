// ref_harness_node.cpp -- C entry points around the REFERENCE's own mapper NODE class (TEST INFRASTRUCTURE).
//
// oracle/_ref/libesvo_ref_node.so = this file + the unmodified esvo_core/src/esvo_Mapping.cpp and the mapper sources it
// drives (container/{DepthPoint,CameraSystem}.cpp, core/{EventBM,DepthProblem,DepthProblemSolver,DepthFusion,
// DepthRegularization}.cpp, tools/sobel.cpp), compiled where they lie under /root/reference against the stand-in headers of
// oracle/ref_shim_node/ (ROS, tf, message_filters, image_transport, dynamic_reconfigure, cv_bridge, pcl) and ref_shim/.
// What runs is the node's own eventsCallback, timeSurfaceCallback, dataTransferring (event selection, the table of virtual
// views), MappingAtTime (denoising, block matching, refinement, culling, the window policy, fusion, clean, regularisation)
// and InitializationAtTime -- esvo_Mapping.cpp:261-600,669-760,1000-1072.  Stand-ins, not reference code: tf (every
// lookup is answered by the harness's pose function: tf's interpolation is third-party), OpenCV (the calibration
// products and the StereoSGBM disparity are injected, medianBlur(3) on the denoising mask is a 3x3 median), the
// publishers (nobody subscribes), Visualization::plot_eventMap (restated below: Visualization.cpp draws with OpenCV).
// With -DREF_NODE_MVSTEREO the same entry points wrap esvo_core/src/esvo_MVStereo.cpp (+ core/EventMatcher.cpp, linked but
// unused) in BM_PLUS_ESTIMATION mode -> oracle/_ref/libesvo_ref_mvstereo.so: dataTransferring :567-668, MappingAtTime :244-565.
// The node's MappingLoop thread leaves at once (ros::ok() is false); the harness calls the stages itself.
//
// One known difference from ref_harness.cpp: between SmartGrid::clean and the regulariser the node dereferences grid cells
// that point at erased elements (undefined behaviour upstream); ref_harness.cpp nulls them, the node
// cannot.  The fixture (tests/golden/ref_node.npz) is therefore recorded with Regularization off; with it on the node's
// maps equal ref_harness.cpp's in everything but the inverse depths around those cells (asserted when the fixture is made).
#include <chrono>
#include <cstring>
#include <deque>
#include <future>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>
#include <ros/ros.h>
#include <Eigen/Eigen>
#define private public
#define protected public
#ifdef REF_NODE_MVSTEREO   // -DREF_NODE_MVSTEREO: the same entry points around esvo_MVStereo (BM_PLUS_ESTIMATION mode)
#include <esvo_core/esvo_MVStereo.h>
#else
#include <esvo_core/esvo_Mapping.h>
#endif
#undef private
#undef protected

#include <cstring>
#include <memory>
#include <sstream>

#include "../include/esvo_hip.h"
#ifdef REF_NODE_WITH_HIP  // + the product's reference-side binding (include/esvo_hip_mapping_node.hpp), linked with libesvo_hip.so
#include "../include/esvo_hip_mapping_node.hpp"
#endif

using namespace esvo_core;
#ifdef REF_NODE_MVSTEREO
typedef esvo_MVStereo NodeClass;
#else
typedef esvo_Mapping NodeClass;
#endif

// ---- Visualization (tools/Visualization.cpp is OpenCV drawing code): only plot_eventMap matters to the mapper ----
namespace esvo_core {
namespace tools {
Visualization::Visualization() {}
Visualization::~Visualization() {}
void Visualization::plot_map(DepthMap::Ptr&, VisMapType, cv::Mat&, double, double, double, double) {}
void Visualization::plot_events(std::vector<Eigen::Matrix<double, 2, 1>, Eigen::aligned_allocator<Eigen::Matrix<double, 2, 1>>>&,
                                cv::Mat&, size_t, size_t) {}
void Visualization::DrawPoint(double, double, double, const Eigen::Vector2d&, cv::Mat&) {}
// Visualization.cpp:96-104
void Visualization::plot_eventMap(std::vector<dvs_msgs::Event*>& vEventPtr, cv::Mat& eventMap, size_t row, size_t col) {
  eventMap = cv::Mat(cv::Size(col, row), CV_8UC1, cv::Scalar(0));
  for (size_t i = 0; i < vEventPtr.size(); i++) eventMap.at<uchar>(vEventPtr[i]->y, vEventPtr[i]->x) = 255;
}
}  // namespace tools
}  // namespace esvo_core

struct ref_node {
  ros::NodeHandle nh, pnh;
  std::unique_ptr<NodeClass> node;
  std::vector<short> disparity;
  int W = 0, H = 0;
#ifdef REF_NODE_WITH_HIP
  std::unique_ptr<esvo_hip::MappingNodeHip<NodeClass, cv::Mat>> hip;
#endif
};

namespace {
template <class T> void setp(const char* name, const T& v) {
  std::ostringstream os;
  os.precision(17);
  os << v;
  esvo_node_shim::params()[name] = os.str();
}
}  // namespace

namespace { std::map<std::string, std::string>& overrides() { static std::map<std::string, std::string> m; return m; } }
extern "C" {
// a private-namespace parameter the POD has no field for (e.g. INIT_SGM_DP_NUM_THRESHOLD); applies to the next ref_node_create
void ref_node_preset_param(const char* name, const char* value) { overrides()[name] = value; }
// the private-namespace parameters of cfg/mapping/*.yaml, from the POD the C-ABI uses
ref_node* ref_node_create(const char* calib_dir, const esvo_params_t* p, const esvo_calib_t* left, const esvo_calib_t* right) {
  ref_node* h = new ref_node;
  h->W = left->width; h->H = left->height;
  esvo_ref_shim::Injected& g = esvo_ref_shim::inject();
  g.W = h->W; g.H = h->H; g.cam = 0;
  g.lut[0] = left->rect_lut; g.lut[1] = right->rect_lut;
  g.mask[0] = left->rect_mask; g.mask[1] = right->rect_mask;
  esvo_node_shim::params().clear();
  setp("calibInfoDir", std::string(calib_dir));
  setp("patch_size_X", p->patch_size_x); setp("patch_size_Y", p->patch_size_y);
  setp("LSnorm", std::string("Tdist"));
  setp("Tdist_nu", p->td_nu); setp("Tdist_scale", p->td_scale);
  setp("ITERATION_OPTIMIZATION", p->lm_max_iteration);
  setp("RegularizationRadius", p->reg_radius); setp("RegularizationMinNeighbours", p->reg_min_neighbours);
  setp("RegularizationMinCloseNeighbours", p->reg_min_close_neighbours);
  setp("SmoothTimeSurface", 0);  // the harness hands over observations that are smoothed already (GaussianBlur is OpenCV)
  setp("invDepth_min_range", p->invdepth_min); setp("invDepth_max_range", p->invdepth_max);
  setp("residual_vis_threshold", p->residual_vis_threshold); setp("stdVar_vis_threshold", p->stdvar_vis_threshold);
  setp("age_max_range", 5); setp("age_vis_threshold", p->age_vis_threshold);
  setp("fusion_radius", p->fusion_radius); setp("maxNumFusionFrames", p->max_fusion_frames);
  setp("FUSION_STRATEGY", std::string(p->fusion_strategy == ESVO_FUSION_CONST_POINTS ? "CONST_POINTS" : "CONST_FRAMES"));
  setp("maxNumFusionPoints", p->max_fusion_points);
  setp("Denoising", p->denoising ? 1 : 0); setp("Regularization", p->regularization ? 1 : 0);
  setp("PROCESS_EVENT_NUM", p->process_event_num); setp("TS_HISTORY_LENGTH", 100);
  setp("BM_half_slice_thickness", p->bm_half_slice_thickness);
  setp("BM_min_disparity", p->bm_min_disparity); setp("BM_max_disparity", p->bm_max_disparity);
  setp("BM_step", p->bm_step); setp("BM_ZNCC_Threshold", p->bm_zncc_threshold);
  setp("BM_bUpDownConfiguration", p->bm_updown ? 1 : 0);
#ifdef REF_NODE_MVSTEREO
  setp("MVStereoMode", 3);  // BM_PLUS_ESTIMATION (esvo_MVStereo.h:43-50), the mode of cfg/mvstereo/*.yaml (1: PURE_BLOCK_MATCHING)
#endif
  for (auto& kv : overrides()) esvo_node_shim::params()[kv.first] = kv.second;
  overrides().clear();
  h->node.reset(new NodeClass(h->nh, h->pnh));
#ifndef REF_NODE_MVSTEREO
  h->node->ESVO_System_Status_ = "WORKING";
  h->nh.setParam("/ESVO_SYSTEM_STATUS", std::string("WORKING"));
#endif
  return h;
}
void ref_node_destroy(ref_node* h) { delete h; }
void ref_node_set_pose_fn(esvo_node_shim::pose_fn f) { esvo_node_shim::pose_cb() = f; }
void ref_node_set_status(ref_node* h, const char* s) {
#ifndef REF_NODE_MVSTEREO
  h->node->ESVO_System_Status_ = s;
#endif
  h->nh.setParam("/ESVO_SYSTEM_STATUS", std::string(s));
}
// esvo_Mapping::eventsCallback (:669-703) on the left queue
void ref_node_push_events(ref_node* h, const esvo_event_t* ev, size_t n) {
  if (!n) return;
  auto msg = std::make_shared<dvs_msgs::EventArray>();
  msg->width = (uint32_t)h->W; msg->height = (uint32_t)h->H;
  msg->events.resize(n);
  for (size_t i = 0; i < n; ++i) {
    msg->events[i].x = ev[i].x; msg->events[i].y = ev[i].y;
    msg->events[i].ts = ros::Time(ev[i].sec, ev[i].nsec);
    msg->events[i].polarity = ev[i].polarity;
  }
  h->node->eventsCallback(msg, h->node->events_left_);
}
// esvo_Mapping::timeSurfaceCallback (:718-760) with one stereo pair of mono8 images
void ref_node_push_time_surfaces(ref_node* h, uint64_t t_ns, const uint8_t* left, const uint8_t* right) {
  auto mk = [&](const uint8_t* img) {
    auto m = std::make_shared<sensor_msgs::Image>();
    m->header.stamp = ros::Time((uint32_t)(t_ns / 1000000000ull), (uint32_t)(t_ns % 1000000000ull));
    m->width = (unsigned)h->W; m->height = (unsigned)h->H;
    m->data.assign(img, img + (size_t)h->W * h->H);
    return m;
  };
  h->node->timeSurfaceCallback(mk(left), mk(right));
}
// dataTransferring (:494-600): returns 1 when an observation, its events and its virtual views were loaded
int ref_node_data_transferring(ref_node* h) { return h->node->dataTransferring() ? 1 : 0; }
uint64_t ref_node_obs_time(ref_node* h) { return h->node->TS_obs_.first.toNSec(); }
// the events dataTransferring selected (as indices into the left queue, newest first, the order the node keeps them in)
namespace {
size_t index_events(ref_node* h, std::vector<dvs_msgs::Event*>& v, uint32_t* idx_out, size_t cap) {
  // position of every event in the left deque (not contiguous: pointer -> index through a map built once per call)
  std::map<const dvs_msgs::Event*, uint32_t> where;
  for (size_t j = 0; j < h->node->events_left_.size(); ++j) where[&h->node->events_left_[j]] = (uint32_t)j;
  for (size_t i = 0; i < v.size() && i < cap; ++i) {
    auto it = where.find(v[i]);
    idx_out[i] = it == where.end() ? 0xffffffffu : it->second;
  }
  return v.size();
}
}  // namespace
// the events dataTransferring selected (indices into the left queue, in the order the node keeps them)
size_t ref_node_selected_events(ref_node* h, uint32_t* idx_out, size_t cap) {
  return index_events(h, h->node->vCloseEventsPtr_left_, idx_out, cap);
}
// the events dataTransferring keeps for the SGM bootstrap while the status is INITIALIZATION (:538-552)
size_t ref_node_sgm_events(ref_node* h, uint32_t* idx_out, size_t cap) {
  return index_events(h, h->node->vEventsPtr_left_SGM_, idx_out, cap);
}
// the events MappingAtTime handed to the block matcher (:286-306: denoised, or the first PROCESS_EVENT_NUM close ones)
size_t ref_node_matched_events(ref_node* h, uint32_t* idx_out, size_t cap) {
  return index_events(h, h->node->vDenoisedEventsPtr_left_, idx_out, cap);
}
size_t ref_node_pose_table(ref_node* h, uint64_t* stamps, double* poses, size_t cap) {
  size_t k = 0;
  for (auto& kv : h->node->st_map_) {
    if (k < cap) {
      stamps[k] = kv.first.toNSec();
      Eigen::Matrix<double, 4, 4> T = kv.second.getTransformationMatrix();
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) poses[k * 16 + i * 4 + j] = T(i, j);
    }
    ++k;
  }
  return k;
}
// MappingAtTime (:261-431) on what dataTransferring loaded
void ref_node_mapping_at_time(ref_node* h) {
  h->node->MappingAtTime(h->node->TS_obs_.first);
  // MappingAtTime hands the DepthMap (a shared_ptr, by value) to a detached publisher thread (:400-403): wait until that
  // thread has let go of it, so that what the harness reads next -- the map, the clouds -- is not being written
  for (int spin = 0; spin < 20000 && h->node->depthFramePtr_->dMap_.use_count() > 1; ++spin)
    std::this_thread::sleep_for(std::chrono::microseconds(100));
}
#ifndef REF_NODE_MVSTEREO
// publishPointCloud (:909-978) on the current DepthFrame: the cloud handed to the tracker (which = 0) or the points
// closer than visualize_range (which = 1), float32 xyz in the world frame
size_t ref_node_pointcloud(ref_node* h, int which, float* out_xyz, size_t cap_points) {
  ros::Time t = h->node->TS_obs_.first;
  h->node->publishPointCloud(h->node->depthFramePtr_->dMap_, h->node->depthFramePtr_->T_world_frame_, t);
  auto& pc = which ? *h->node->pc_near_ : *h->node->pc_;
  size_t k = 0;
  for (auto& p : pc.points) {
    if (k < cap_points) { out_xyz[3 * k] = p.x; out_xyz[3 * k + 1] = p.y; out_xyz[3 * k + 2] = p.z; }
    ++k;
  }
  return k;
}
#endif
// InitializationAtTime (:433-492) with the given disparity image standing in for StereoSGBM::compute
int ref_node_initialization_at_time(ref_node* h, const int16_t* disp16) {
  h->disparity.assign(disp16, disp16 + (size_t)h->W * h->H);
  esvo_ref_shim::inject().disparity = h->disparity.data();
#ifndef REF_NODE_MVSTEREO
  return h->node->InitializationAtTime(h->node->TS_obs_.first) ? 1 : 0;
#else
  return 0;  // esvo_MVStereo has no bootstrap
#endif
}
size_t ref_node_window(ref_node* h, uint32_t* sizes, size_t cap) {
  size_t k = 0;
  for (auto& f : h->node->dqvDepthPoints_) { if (k < cap) sizes[k] = (uint32_t)f.size(); ++k; }
  return k;
}
// the newest frame of the fusion window (the DepthPoints this tick's optimisation kept, :327-339), pose as an index into
// the tick's virtual-view table
size_t ref_node_newest_frame(ref_node* h, esvo_depth_point_t* out, size_t cap) {
  if (h->node->dqvDepthPoints_.empty()) return 0;
  auto& f = h->node->dqvDepthPoints_.back();
  std::vector<Eigen::Matrix<double, 4, 4>> table;
  for (auto& kv : h->node->st_map_) table.push_back(kv.second.getTransformationMatrix());
  for (size_t k = 0; k < f.size() && k < cap; ++k) {
    const DepthPoint& d = f[k];
    esvo_depth_point_t& o = out[k];
    o.row = (uint32_t)d.row(); o.col = (uint32_t)d.col();
    o.x[0] = d.x()(0); o.x[1] = d.x()(1);
    o.inv_depth = d.invDepth(); o.scale2 = d.scaleSquared(); o.nu = d.nu(); o.variance = d.variance();
    o.residual = d.residual(); o.age = d.age();
    for (int q = 0; q < 3; ++q) o.p_cam[q] = d.p_cam()(q);
    o.pose_idx = 0xffffffffu; o.seq = (uint32_t)k;
    for (size_t m = 0; m < table.size(); ++m) {
      bool same = true;
      for (int i = 0; i < 4 && same; ++i)
        for (int j = 0; j < 4; ++j)
          if (table[m](i, j) != d.T_world_cam()(i, j)) { same = false; break; }
      if (same) { o.pose_idx = (uint32_t)m; break; }
    }
  }
  return f.size();
}
size_t ref_node_get_map(ref_node* h, esvo_depth_point_t* out, size_t cap) {
  size_t k = 0;
  DepthMap& dm = *h->node->depthFramePtr_->dMap_;
  for (auto it = dm.begin(); it != dm.end(); ++it, ++k)
    if (k < cap) {
      const DepthPoint& d = *it;
      esvo_depth_point_t& o = out[k];
      o.row = (uint32_t)d.row(); o.col = (uint32_t)d.col();
      o.x[0] = d.x()(0); o.x[1] = d.x()(1);
      o.inv_depth = d.invDepth(); o.scale2 = d.scaleSquared(); o.nu = d.nu(); o.variance = d.variance();
      o.residual = d.residual(); o.age = d.age();
      for (int q = 0; q < 3; ++q) o.p_cam[q] = d.p_cam()(q);
      o.pose_idx = 0; o.seq = (uint32_t)k;
    }
  return k;
}

#ifdef REF_NODE_WITH_HIP
// ---- the same node object with MappingAtTime replaced by the device path (include/esvo_hip_mapping_node.hpp) ----
static std::string& hip_error() { static std::string e; return e; }
const char* ref_node_hip_error() { return hip_error().c_str(); }
int ref_node_hip_attach(ref_node* h, const esvo_params_t* p, const esvo_calib_t* left, const esvo_calib_t* right, int device) {
  try {
    h->hip.reset(new esvo_hip::MappingNodeHip<NodeClass, cv::Mat>(*h->node, *p, *left, *right, device));
  } catch (const esvo_hip::Error& e) { hip_error() = e.what(); return e.code; }
  return 0;
}
// what MappingLoop would call in place of MappingAtTime(TS_obs_.first)
int ref_node_hip_mapping_at_time(ref_node* h) {
  try {
    h->hip->MappingAtTime();
  } catch (const esvo_hip::Error& e) { hip_error() = e.what(); return e.code; }
  return 0;
}
size_t ref_node_hip_matched_events(ref_node* h, uint32_t* idx_out, size_t cap) {
  return index_events(h, h->node->vDenoisedEventsPtr_left_, idx_out, cap);
}
size_t ref_node_hip_newest_frame(ref_node* h, esvo_depth_point_t* out, size_t cap) {
  const auto& f = h->hip->newestFrame();
  for (size_t i = 0; i < f.size() && i < cap; ++i) out[i] = f[i];
  return f.size();
}
size_t ref_node_hip_get_map(ref_node* h, esvo_depth_point_t* out, size_t cap) {
  std::vector<esvo_depth_point_t> m;
  try {
    h->hip->getDepthMap(m);
  } catch (const esvo_hip::Error& e) { hip_error() = e.what(); return 0; }
  for (size_t i = 0; i < m.size() && i < cap; ++i) out[i] = m[i];
  return m.size();
}
#endif
}  // extern "C"
