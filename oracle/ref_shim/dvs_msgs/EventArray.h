#ifndef ESVO_REF_SHIM_DVS_EVENTARRAY
#define ESVO_REF_SHIM_DVS_EVENTARRAY
#include <dvs_msgs/Event.h>
#include <memory>
namespace dvs_msgs {
struct EventArray {
  uint32_t height = 0, width = 0;
  std::vector<Event> events;
  typedef std::shared_ptr<const EventArray> ConstPtr;
};
}  // namespace dvs_msgs
#endif
