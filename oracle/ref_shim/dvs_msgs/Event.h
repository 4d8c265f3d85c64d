// oracle/_ref build shim (TEST INFRASTRUCTURE): in-memory layout of dvs_msgs/Event (rpg_dvs_ros).
#ifndef ESVO_REF_SHIM_DVS_EVENT
#define ESVO_REF_SHIM_DVS_EVENT
#include <ros/time.h>
#include <cstdint>
#include <vector>
namespace dvs_msgs {
struct Event {
  uint16_t x = 0, y = 0;
  ros::Time ts;
  uint8_t polarity = 0;
};
}  // namespace dvs_msgs
#endif
