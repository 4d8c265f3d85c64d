// oracle/_ref build shim for the mapper NODE (esvo_Mapping.cpp) (TEST INFRASTRUCTURE): enough of the ROS API for the node to be
// constructed and for dataTransferring / MappingAtTime / InitializationAtTime to run; nothing is published, no thread loops.
#ifndef ESVO_REF_SHIM_NODE_IT
#define ESVO_REF_SHIM_NODE_IT
#include <ros/ros.h>
#include <sensor_msgs/Image.h>
namespace image_transport {
struct Publisher { void shutdown() {} unsigned getNumSubscribers() const { return 0; } void publish(const sensor_msgs::ImagePtr&) const {} };
struct ImageTransport { explicit ImageTransport(const ros::NodeHandle&) {} Publisher advertise(const std::string&, unsigned) { return Publisher(); } };
}
#endif
