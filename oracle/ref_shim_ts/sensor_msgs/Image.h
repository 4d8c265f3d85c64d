// oracle/_ref build shim (TEST INFRASTRUCTURE)
#ifndef ESVO_REF_SHIM_TS_SENSOR_IMAGE
#define ESVO_REF_SHIM_TS_SENSOR_IMAGE
#include <memory>
namespace sensor_msgs { struct Image {}; typedef std::shared_ptr<Image> ImagePtr; }
#endif
