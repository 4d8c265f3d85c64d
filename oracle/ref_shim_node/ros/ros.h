// oracle/_ref build shim for the mapper NODE (esvo_Mapping.cpp) (TEST INFRASTRUCTURE): enough of the ROS API for the node to be
// constructed and for dataTransferring / MappingAtTime / InitializationAtTime to run; nothing is published, no thread loops.
#ifndef ESVO_REF_SHIM_NODE_ROS
#define ESVO_REF_SHIM_NODE_ROS
#include <ros/time.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <cstdio>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <sstream>
#include <string>
namespace esvo_node_shim {
inline std::map<std::string, std::string>& params() { static std::map<std::string, std::string> p; return p; }
template <class T> inline T parse(const std::string& s) { T v{}; std::istringstream is(s); is >> v; return v; }
template <> inline std::string parse<std::string>(const std::string& s) { return s; }
template <> inline bool parse<bool>(const std::string& s) { return s == "1" || s == "true" || s == "True"; }
}
namespace ros {
inline bool ok() { return false; }  // the node's MappingLoop thread leaves at once: the harness calls its stages
struct Subscriber {};
struct Publisher {
  void shutdown() {}
  unsigned getNumSubscribers() const { return 0; }
  template <class M> void publish(const M&) const {}
};
struct NodeHandle {
  bool hasParam(const std::string& n) const { return esvo_node_shim::params().count(n) != 0; }
  template <class T> bool getParam(const std::string& n, T& v) const {
    auto it = esvo_node_shim::params().find(n);
    if (it == esvo_node_shim::params().end()) return false;
    v = esvo_node_shim::parse<T>(it->second);
    return true;
  }
  template <class T> void param(const std::string& n, T& v, const T& def) const { if (!getParam(n, v)) v = def; }
  template <class T> void setParam(const std::string& n, const T& v) const { std::ostringstream os; os << v; esvo_node_shim::params()[n] = os.str(); }
  template <class M, class T> Subscriber subscribe(const std::string&, unsigned, void (T::*)(M), T*) { return Subscriber(); }
  template <class M, class F> Subscriber subscribe(const std::string&, unsigned, F) { return Subscriber(); }
  template <class M> Publisher advertise(const std::string&, unsigned) { return Publisher(); }
};
}  // namespace ros
// the node writes boost::bind(..., _1, boost::ref(x)) (through ROS's own includes)
#include <functional>
namespace boost { using std::bind; using std::ref; template <class T> using function = std::function<T>; template <class T> using shared_ptr = std::shared_ptr<T>; }
using namespace std::placeholders;
#define ROS_INFO(...) do {} while (0)
#define ROS_ERROR(...) do {} while (0)
#define ROS_ERROR_ONCE(...) do {} while (0)
#define ROS_INFO_STREAM(x) do {} while (0)
#define ROS_WARN_STREAM(x) do {} while (0)
#define ROS_ERROR_STREAM(x) do {} while (0)
#endif
