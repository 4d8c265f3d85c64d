// oracle/_ref build shim (TEST INFRASTRUCTURE)
#ifndef ESVO_REF_SHIM_TS_SENSOR_CAMINFO
#define ESVO_REF_SHIM_TS_SENSOR_CAMINFO
#include <memory>
#include <string>
#include <vector>
namespace sensor_msgs {
struct CameraInfo {
  typedef std::shared_ptr<const CameraInfo> ConstPtr;
  unsigned width = 0, height = 0;
  std::string distortion_model;
  std::vector<double> D;
  double K[9] = {0}, R[9] = {0}, P[12] = {0};
};
}
#endif
