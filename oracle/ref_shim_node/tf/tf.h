// oracle/_ref build shim for the mapper NODE (esvo_Mapping.cpp) (TEST INFRASTRUCTURE): enough of the ROS API for the node to be
// constructed and for dataTransferring / MappingAtTime / InitializationAtTime to run; nothing is published, no thread loops.
// tf::Transformer answers every lookup from ONE function the harness installs (esvo_node_shim::pose_cb): the pose of the
// left camera at a stamp.  tf's own interpolation between stamped poses is third-party and not restated.
#ifndef ESVO_REF_SHIM_NODE_TF
#define ESVO_REF_SHIM_NODE_TF
#include <ros/time.h>
#include <geometry_msgs/PoseStamped.h>
#include <string>
namespace esvo_node_shim {
typedef int (*pose_fn)(unsigned long long t_ns, double T_world_cam[16]);  // 0: no pose at that stamp
inline pose_fn& pose_cb() { static pose_fn f = nullptr; return f; }
}
namespace tf {
struct Quaternion { double x, y, z, w; Quaternion(double x_ = 0, double y_ = 0, double z_ = 0, double w_ = 1) : x(x_), y(y_), z(z_), w(w_) {} };
struct Vector3 { double x, y, z; Vector3(double x_ = 0, double y_ = 0, double z_ = 0) : x(x_), y(y_), z(z_) {} };
struct Transform { Quaternion q; Vector3 t; Transform() {} Transform(const Quaternion& q_, const Vector3& t_) : q(q_), t(t_) {} };
struct StampedTransform : Transform {
  ros::Time stamp_;
  double T[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};  // what the shim's Transformer hands to transformTFToKindr
  StampedTransform() {}
  StampedTransform(const Transform& tr, const ros::Time& s, const std::string&, const std::string&) : Transform(tr), stamp_(s) {}
};
struct Transformer {
  Transformer(bool = true, ros::Duration = ros::Duration()) {}
  bool setTransform(const StampedTransform&) { return true; }
  void clear() {}
  int getLatestCommonTime(const std::string&, const std::string&, ros::Time& t, std::string*) const { t = ros::Time(); return 0; }
  bool canTransform(const std::string&, const std::string&, const ros::Time& t, std::string* = nullptr) const {
    double T[16];
    return esvo_node_shim::pose_cb() && esvo_node_shim::pose_cb()(t.toNSec(), T) != 0;
  }
  void lookupTransform(const std::string&, const std::string&, const ros::Time& t, StampedTransform& st) const {
    esvo_node_shim::pose_cb()(t.toNSec(), st.T);
    st.stamp_ = t;
  }
};
}  // namespace tf
#endif
