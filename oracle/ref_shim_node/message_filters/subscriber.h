// oracle/_ref build shim for the mapper NODE (esvo_Mapping.cpp) (TEST INFRASTRUCTURE): enough of the ROS API for the node to be
// constructed and for dataTransferring / MappingAtTime / InitializationAtTime to run; nothing is published, no thread loops.
#ifndef ESVO_REF_SHIM_NODE_MF
#define ESVO_REF_SHIM_NODE_MF
#include <ros/ros.h>
namespace message_filters {
template <class M> struct Subscriber { Subscriber(ros::NodeHandle&, const std::string&, unsigned) {} };
namespace sync_policies { template <class A, class B> struct ExactTime { explicit ExactTime(unsigned) {} }; }
template <class P> struct Synchronizer {
  template <class A, class B> Synchronizer(const P&, A&, B&) {}
  template <class F> void registerCallback(F) {}
};
}
#endif
