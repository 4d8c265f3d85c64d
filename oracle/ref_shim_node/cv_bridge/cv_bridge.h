// oracle/_ref build shim for the mapper NODE (TEST INFRASTRUCTURE)
#ifndef ESVO_REF_SHIM_CV_BRIDGE
#define ESVO_REF_SHIM_CV_BRIDGE
#include <opencv2/opencv.hpp>
#include <sensor_msgs/Image.h>
#include <memory>
#include <stdexcept>
#include <string>
namespace cv_bridge {
struct Exception : std::runtime_error { Exception() : std::runtime_error("cv_bridge") {} };
struct CvImage {
  std_msgs::Header header;
  std::string encoding;
  cv::Mat image;
  CvImage() {}
  CvImage(const std_msgs::Header& h, const std::string& e, const cv::Mat& m) : header(h), encoding(e), image(m) {}
  sensor_msgs::ImagePtr toImageMsg() const { return std::make_shared<sensor_msgs::Image>(); }
};
typedef std::shared_ptr<CvImage> CvImagePtr;
inline CvImagePtr toCvCopy(const sensor_msgs::ImageConstPtr& msg, const std::string& enc) {
  CvImagePtr p = std::make_shared<CvImage>();
  p->header = msg->header; p->encoding = enc;
  p->image = cv::Mat((int)msg->height, (int)msg->width, CV_8U);
  for (size_t i = 0; i < msg->data.size(); ++i) p->image.v[i] = (double)msg->data[i];
  return p;
}
}  // namespace cv_bridge
#endif
