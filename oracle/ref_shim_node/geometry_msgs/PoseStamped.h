// oracle/_ref build shim for the mapper NODE (esvo_Mapping.cpp) (TEST INFRASTRUCTURE): enough of the ROS API for the node to be
// constructed and for dataTransferring / MappingAtTime / InitializationAtTime to run; nothing is published, no thread loops.
#ifndef ESVO_REF_SHIM_NODE_POSESTAMPED
#define ESVO_REF_SHIM_NODE_POSESTAMPED
#include <std_msgs/Header.h>
#include <memory>
namespace geometry_msgs {
struct Point { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };
struct Pose { Point position; Quaternion orientation; };
struct PoseStamped { std_msgs::Header header; Pose pose; };
typedef std::shared_ptr<const PoseStamped> PoseStampedConstPtr;
typedef std::shared_ptr<PoseStamped> PoseStampedPtr;
struct TransformStamped {};
}
#endif
