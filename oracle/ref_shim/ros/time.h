// oracle/_ref build shim (TEST INFRASTRUCTURE): the slice of ros::Time the mapper sources touch.
#ifndef ESVO_REF_SHIM_ROS_TIME
#define ESVO_REF_SHIM_ROS_TIME
#include <cmath>
#include <cstdint>
namespace ros {
struct Time {
  uint32_t sec = 0, nsec = 0;
  Time() {}
  Time(uint32_t s, uint32_t ns) : sec(s), nsec(ns) {}
  explicit Time(double t) { sec = (uint32_t)std::floor(t); nsec = (uint32_t)std::round((t - sec) * 1e9); sec += nsec / 1000000000u; nsec %= 1000000000u; }
  double toSec() const { return (double)sec + 1e-9 * (double)nsec; }
  uint64_t toNSec() const { return (uint64_t)sec * 1000000000ull + (uint64_t)nsec; }
  bool operator<(const Time& o) const { return sec < o.sec || (sec == o.sec && nsec < o.nsec); }
  bool operator==(const Time& o) const { return sec == o.sec && nsec == o.nsec; }
};
}  // namespace ros
#endif
