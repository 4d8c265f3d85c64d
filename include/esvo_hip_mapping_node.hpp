// esvo_hip_mapping_node.hpp -- the reference-side binding of INTEGRATION.md section 3 as code: what
// esvo_Mapping::MappingAtTime (esvo_core/src/esvo_Mapping.cpp:261-431) -- or esvo_MVStereo::MappingAtTime in its
// BM_PLUS_ESTIMATION mode (esvo_MVStereo.cpp:244-565) -- becomes when the node is built against libesvo_hip.so.
//
// The node keeps its own callbacks, its tf lookups and dataTransferring (event selection, the table of virtual views);
// the adapter takes what dataTransferring loaded straight out of the node's members and replaces the body of
// MappingAtTime: denoising stays the node's own host code, block matching, refinement, culling, the window policy,
// fusion, clean and regularisation run on the device through the classes of esvo_hip.hpp, which carry the reference's
// names.  A template over the node class, so this header pulls in no ROS / Eigen / OpenCV itself: it is compiled against
// the reference's real esvo_Mapping.h wherever that header can be compiled (oracle/ref_harness_node.cpp does, against
// stand-in ROS / OpenCV headers; tests/test_gpu_node_dropin.py then runs the reference node object with and without it).
// Inside the class the members are reachable as written; from outside they are private (the harness opens them).
#ifndef ESVO_HIP_MAPPING_NODE_HPP
#define ESVO_HIP_MAPPING_NODE_HPP

#include <algorithm>
#include <cstdint>
#include <memory>
#include <vector>

#include "esvo_hip.hpp"

namespace esvo_hip {

// Node: esvo_core::esvo_Mapping or esvo_core::esvo_MVStereo; MaskImage: the image type of the node's denoising mask (cv::Mat)
template <class Node, class MaskImage>
class MappingNodeHip {
 public:
  // params: the node's yaml values as esvo_params_t (INTEGRATION.md section 1 lists the member behind every field);
  // left / right: the calibration products the node's CameraSystem holds
  MappingNodeHip(Node& node, const esvo_params_t& params, const esvo_calib_t& left, const esvo_calib_t& right, int device = 0)
      : node_(node), ctx_(std::make_shared<Context>(params, left, right, device)), ebm_(ctx_), solver_(ctx_), fusion_(ctx_) {}

  // the replacement of MappingAtTime(t): call it where MappingLoop calls the original (:243), after dataTransferring
  void MappingAtTime() {
    const int W = ctx_->width(), H = ctx_->height();
    // TS_obs_: the observation pair as mono8 (TS_left_ holds the image values, TimeSurfaceObservation.h:38-39; with the
    // real cv_bridge image, cvImagePtr_left_->image.data is the same bytes without the conversion) and its pose
    obs_.t_ns = node_.TS_obs_.first.toNSec();
    ts_[0].resize((size_t)W * H);
    ts_[1].resize((size_t)W * H);
    for (int r = 0; r < H; ++r)
      for (int c = 0; c < W; ++c) {
        ts_[0][(size_t)r * W + c] = (uint8_t)node_.TS_obs_.second.TS_left_(r, c);
        ts_[1][(size_t)r * W + c] = (uint8_t)node_.TS_obs_.second.TS_right_(r, c);
      }
    obs_.TS_left = ts_[0].data();
    obs_.TS_right = ts_[1].data();
    row_major(node_.TS_obs_.second.tr_.getTransformationMatrix(), obs_.T_world_cam);
    // st_map_: the virtual views of the slice (:585-599)
    st_map_.clear();
    for (auto& kv : node_.st_map_) {
      double T[16];
      row_major(kv.second.getTransformationMatrix(), T);
      st_map_.emplace(kv.first.toNSec(), T);
    }
    // :281-306 -- the node's own denoising (OpenCV median on the event map) or the first PROCESS_EVENT_NUM events
    auto& vDenoised = node_.vDenoisedEventsPtr_left_;
    vDenoised.clear();
    if (node_.bDenoising_) {
      MaskImage denoising_mask;
      node_.createDenoisingMask(node_.vALLEventsPtr_left_, denoising_mask, (size_t)H, (size_t)W);
      node_.extractDenoisedEvents(node_.vCloseEventsPtr_left_, vDenoised, denoising_mask, node_.PROCESS_EVENT_NUM_);
      node_.totalNumCount_ = vDenoised.size();
    } else {
      vDenoised.insert(vDenoised.end(), node_.vCloseEventsPtr_left_.begin(),
                       node_.vCloseEventsPtr_left_.begin() + std::min(node_.vCloseEventsPtr_left_.size(), node_.PROCESS_EVENT_NUM_));
    }
    events_.resize(vDenoised.size());
    for (size_t i = 0; i < vDenoised.size(); ++i) {
      Event& e = events_[i];
      e.x = vDenoised[i]->x; e.y = vDenoised[i]->y;
      e.sec = vDenoised[i]->ts.sec; e.nsec = vDenoised[i]->ts.nsec;
      e.polarity = vDenoised[i]->polarity ? 1 : 0;
      e._pad[0] = e._pad[1] = e._pad[2] = 0;
    }
    // :307-339 -- block matching, refinement, culling
    ebm_.createMatchProblem(&obs_, &st_map_, &events_);
    ebm_.match_all_HyperThread(vEMP_);
    if (pureBlockMatching(node_, 0)) {  // esvo_MVStereo in MVStereoMode 1 (esvo_MVStereo.cpp:411-432): no refinement, no fusion
      vdp_.clear();
      fusion_.naivePropagation(vEMP_, st_map_);
      numFusionCount_ = 0;
      return;
    }
    solver_.solve(&vEMP_, &obs_, vdp_);
    solver_.pointCulling(vdp_, node_.stdVar_vis_threshold_, node_.cost_vis_threshold_, node_.invDepth_min_range_,
                         node_.invDepth_max_range_);
    // :341-395 -- window policy, fusion over the window, clean, regularisation
    fusion_.pushFrame(vdp_, st_map_);
    numFusionCount_ = fusion_.update();
  }

  const std::vector<EventMatchPair>& matches() const { return vEMP_; }
  const std::vector<DepthPoint>& newestFrame() const { return vdp_; }
  size_t numFusionCount() const { return numFusionCount_; }
  void getDepthMap(std::vector<DepthPoint>& out) { fusion_.getDepthMap(out); }       // depthFramePtr_->dMap_
  void getPointCloud(std::vector<float>& xyz) { fusion_.getPointCloud(xyz); }         // publishPointCloud's payload
  Context& context() { return *ctx_; }

 private:
  // msm_ == PURE_BLOCK_MATCHING where the node class has that member (esvo_MVStereo.h:43-50,155); esvo_Mapping has none
  template <class N> static auto pureBlockMatching(const N& n, int) -> decltype((void)n.msm_, bool()) { return (int)n.msm_ == 1; }
  template <class N> static bool pureBlockMatching(const N&, long) { return false; }
  template <class M> static void row_major(const M& T, double out[16]) {
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) out[i * 4 + j] = T(i, j);
  }

  Node& node_;
  ContextPtr ctx_;
  EventBM ebm_;
  DepthProblemSolver solver_;
  DepthFusion fusion_;
  StampedTimeSurfaceObs obs_;
  StampTransformationMap st_map_;
  std::vector<uint8_t> ts_[2];
  std::vector<Event> events_;
  std::vector<EventMatchPair> vEMP_;
  std::vector<DepthPoint> vdp_;
  size_t numFusionCount_ = 0;
};

}  // namespace esvo_hip
#endif
