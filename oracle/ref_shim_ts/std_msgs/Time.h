// oracle/_ref build shim (TEST INFRASTRUCTURE)
#ifndef ESVO_REF_SHIM_TS_STDMSGS_TIME
#define ESVO_REF_SHIM_TS_STDMSGS_TIME
#include <memory>
#include <ros/time.h>
namespace std_msgs {
struct Time { ros::Time data; };
typedef std::shared_ptr<const Time> TimeConstPtr;
}
#endif
