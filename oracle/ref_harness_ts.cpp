// ref_harness_ts.cpp -- C entry points around the REFERENCE's own Time-Surface node class (TEST INFRASTRUCTURE).
//
// oracle/_ref/libesvo_ref_ts.so = this file + the unmodified esvo_time_surface/src/TimeSurface.cpp (with
// include/esvo_time_surface/TimeSurface.h: EventQueueMat), compiled where it lies under /root/reference against the
// stand-in headers of oracle/ref_shim_ts/ (ROS, image_transport, cv_bridge, a cv::Mat that only stores) and ref_shim/
// (Eigen, dvs_msgs, glog).  What runs is the reference's eventsCallback (:403-425: sorted insertion, per-pixel queues of
// max_event_queue_len events) and createTimeSurfaceAtTime (:52-152: getMostRecentEventBeforeT, dt, exp(-dt / decay),
// polarity, the scale to [0, 255]); what does not exist here is OpenCV -- convertTo's rounding, medianBlur, remap -- so the
// f64 image handed to convertTo is what comes back.  Pins oracle/esvo_oracle.cpp's raster (tests/test_ref_pin.py).
#define private public  // the node's members and callbacks are private; layout unchanged
#include <esvo_time_surface/TimeSurface.h>
#undef private

#include <cstring>
#include <memory>

#include "../include/esvo_hip.h"

using namespace esvo_time_surface;

struct ref_ts {
  ros::NodeHandle nh;
  std::unique_ptr<TimeSurface> node;
  int W = 0, H = 0;
};

extern "C" {
ref_ts* ref_ts_create(int width, int height, double decay_ms, int ignore_polarity, int queue_len) {
  ref_ts* h = new ref_ts;
  h->W = width; h->H = height;
  h->node.reset(new TimeSurface(h->nh, h->nh));
  h->node->decay_ms_ = decay_ms;
  h->node->ignore_polarity_ = ignore_polarity != 0;
  h->node->median_blur_kernel_size_ = 0;  // OpenCV: not part of this build
  h->node->max_event_queue_length_ = queue_len;
  h->node->time_surface_mode_ = TimeSurface::BACKWARD;
  h->node->bCamInfoAvailable_ = true;     // cameraInfoCallback only prepares OpenCV's undistortion maps
  h->node->init(width, height);
  return h;
}
void ref_ts_destroy(ref_ts* h) { delete h; }
// FORWARD mode (TimeSurface.cpp:85-116): rect_lut = what cameraInfoCallback's cv::undistortPoints would have produced
// (:374-399, an OpenCV product: injected), 2 floats per raw pixel
void ref_ts_set_forward(ref_ts* h, const float* rect_lut) {
  const size_t n = (size_t)h->W * h->H;
  h->node->precomputed_rectified_points_ = Eigen::Matrix2Xd(2, n);
  for (size_t i = 0; i < n; ++i)
    h->node->precomputed_rectified_points_.col(i) = Eigen::Matrix<double, 2, 1>(rect_lut[2 * i], rect_lut[2 * i + 1]);
  h->node->time_surface_mode_ = TimeSurface::FORWARD;
}
// eventsCallback with one EventArray message
void ref_ts_push(ref_ts* h, const esvo_event_t* ev, size_t n) {
  auto msg = std::make_shared<dvs_msgs::EventArray>();
  msg->width = (uint32_t)h->W; msg->height = (uint32_t)h->H;
  msg->events.resize(n);
  for (size_t i = 0; i < n; ++i) {
    msg->events[i].x = ev[i].x; msg->events[i].y = ev[i].y;
    msg->events[i].ts = ros::Time(ev[i].sec, ev[i].nsec);
    msg->events[i].polarity = ev[i].polarity;
  }
  h->node->eventsCallback(msg);
}
// createTimeSurfaceAtTime(t): out = the W*H f64 image the node hands to cv::Mat::convertTo(CV_8U)
void ref_ts_render(ref_ts* h, uint64_t t_ns, double* out) {
  esvo_ts_shim::captured().clear();
  h->node->createTimeSurfaceAtTime(ros::Time((uint32_t)(t_ns / 1000000000ull), (uint32_t)(t_ns % 1000000000ull)));
  const std::vector<double>& c = esvo_ts_shim::captured();
  std::memcpy(out, c.data(), sizeof(double) * c.size());
}
}  // extern "C"
