// oracle/_ref build shim (TEST INFRASTRUCTURE)
#ifndef ESVO_REF_SHIM_TS_CV_BRIDGE
#define ESVO_REF_SHIM_TS_CV_BRIDGE
#include <memory>
#include <string>
#include <opencv2/cv_stub.hpp>
#include <ros/time.h>
#include <sensor_msgs/Image.h>
namespace cv_bridge {
struct Header { ros::Time stamp; };
struct CvImage {
  Header header;
  std::string encoding;
  cv::Mat image;
  sensor_msgs::ImagePtr toImageMsg() const { return std::make_shared<sensor_msgs::Image>(); }
};
}
#endif
