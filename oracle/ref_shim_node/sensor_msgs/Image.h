// oracle/_ref build shim for the mapper NODE (esvo_Mapping.cpp) (TEST INFRASTRUCTURE): enough of the ROS API for the node to be
// constructed and for dataTransferring / MappingAtTime / InitializationAtTime to run; nothing is published, no thread loops.
#ifndef ESVO_REF_SHIM_NODE_SENSOR_IMAGE
#define ESVO_REF_SHIM_NODE_SENSOR_IMAGE
#include <std_msgs/Header.h>
#include <memory>
#include <vector>
namespace sensor_msgs {
struct Image { std_msgs::Header header; unsigned width = 0, height = 0; std::vector<unsigned char> data; };
typedef std::shared_ptr<Image> ImagePtr;
typedef std::shared_ptr<const Image> ImageConstPtr;
struct PointCloud2 { typedef std::shared_ptr<PointCloud2> Ptr; std_msgs::Header header; };
namespace image_encodings { static const char* const MONO8 = "mono8"; static const char* const BGR8 = "bgr8"; }
}
#endif
