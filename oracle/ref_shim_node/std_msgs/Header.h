// oracle/_ref build shim for the mapper NODE (esvo_Mapping.cpp) (TEST INFRASTRUCTURE): enough of the ROS API for the node to be
// constructed and for dataTransferring / MappingAtTime / InitializationAtTime to run; nothing is published, no thread loops.
#ifndef ESVO_REF_SHIM_NODE_HEADER
#define ESVO_REF_SHIM_NODE_HEADER
#include <ros/time.h>
#include <string>
namespace std_msgs { struct Header { ros::Time stamp; std::string frame_id; unsigned seq = 0; }; }
#endif
