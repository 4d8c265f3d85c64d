// oracle/_ref build shim for the mapper NODE (esvo_Mapping.cpp) (TEST INFRASTRUCTURE): enough of the ROS API for the node to be
// constructed and for dataTransferring / MappingAtTime / InitializationAtTime to run; nothing is published, no thread loops.
#ifndef ESVO_REF_SHIM_NODE_TF2
#define ESVO_REF_SHIM_NODE_TF2
namespace tf2_ros { struct TransformBroadcaster {}; }
#endif
