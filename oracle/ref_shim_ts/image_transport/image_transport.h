// oracle/_ref build shim (TEST INFRASTRUCTURE): nobody subscribes, so the node never publishes
#ifndef ESVO_REF_SHIM_TS_IMAGE_TRANSPORT
#define ESVO_REF_SHIM_TS_IMAGE_TRANSPORT
#include <ros/ros.h>
#include <sensor_msgs/Image.h>
#include <string>
namespace image_transport {
struct Publisher {
  void shutdown() {}
  unsigned getNumSubscribers() const { return 0; }
  void publish(const sensor_msgs::ImagePtr&) {}
};
struct ImageTransport {
  explicit ImageTransport(ros::NodeHandle&) {}
  Publisher advertise(const std::string&, unsigned) { return Publisher(); }
};
}
#endif
