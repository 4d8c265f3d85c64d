// oracle/_ref build shim for the Time-Surface node (TEST INFRASTRUCTURE): the node's constructor only needs these to exist.
#ifndef ESVO_REF_SHIM_TS_ROS
#define ESVO_REF_SHIM_TS_ROS
#include <ros/time.h>
#include <cstdio>
#include <string>
namespace ros {
struct Subscriber {};
struct NodeHandle {
  template <class M, class T> Subscriber subscribe(const std::string&, unsigned, void (T::*)(M), T*) { return Subscriber(); }
  template <class V> void param(const std::string&, V& var, const V& def) const { var = def; }
  template <class V, class D> void param(const std::string&, V& var, const D& def) const { var = (V)def; }
};
}  // namespace ros
#define ROS_INFO(...) do {} while (0)
#define ROS_ERROR_ONCE(...) do {} while (0)
#endif
