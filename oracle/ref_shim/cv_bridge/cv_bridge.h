// oracle/_ref build shim (TEST INFRASTRUCTURE): cv_bridge::CvImagePtr as TimeSurfaceObservation.h uses it.
#ifndef ESVO_REF_SHIM_CV_BRIDGE
#define ESVO_REF_SHIM_CV_BRIDGE
#include <opencv2/opencv.hpp>
#include <memory>
namespace cv_bridge {
struct CvImage { cv::Mat image; };
typedef std::shared_ptr<CvImage> CvImagePtr;
}  // namespace cv_bridge
#endif
