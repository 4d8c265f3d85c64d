// oracle/_ref build shim for the mapper NODE (esvo_Mapping.cpp) (TEST INFRASTRUCTURE): enough of the ROS API for the node to be
// constructed and for dataTransferring / MappingAtTime / InitializationAtTime to run; nothing is published, no thread loops.
// setCallback does NOT fire the callback (the real server does, with the .cfg defaults): the node keeps its yaml parameters
#ifndef ESVO_REF_SHIM_NODE_DYNRECONF
#define ESVO_REF_SHIM_NODE_DYNRECONF
#include <ros/ros.h>
#include <cstdint>
namespace dynamic_reconfigure {
template <class C> struct Server {
  typedef std::function<void(C&, uint32_t)> CallbackType;
  explicit Server(const ros::NodeHandle&) {}
  void setCallback(const CallbackType&) {}
};
}
#endif
