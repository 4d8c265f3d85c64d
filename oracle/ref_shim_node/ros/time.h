// oracle/_ref build shim for the Time-Surface node (TEST INFRASTRUCTURE): ros::Time / ros::Duration as roscpp defines them
// (time.h / duration.h: sec + nsec pairs, normalised differences, toSec() = sec + 1e-9 * nsec).
#ifndef ESVO_REF_SHIM_TS_ROS_TIME
#define ESVO_REF_SHIM_TS_ROS_TIME
#define ESVO_REF_SHIM_ROS_TIME  // supersedes ref_shim/ros/time.h
#include <cmath>
#include <cstdint>
namespace ros {
struct Duration {
  int32_t sec = 0, nsec = 0;
  Duration() {}
  explicit Duration(double) {}
  Duration(int32_t s, int32_t ns) {  // normalizeSecNSecSigned
    int64_t s64 = s, n64 = ns;
    while (n64 >= 1000000000ll) { n64 -= 1000000000ll; ++s64; }
    while (n64 < 0) { n64 += 1000000000ll; --s64; }
    sec = (int32_t)s64; nsec = (int32_t)n64;
  }
  double toSec() const { return (double)sec + 1e-9 * (double)nsec; }
};
struct Time {
  uint32_t sec = 0, nsec = 0;
  Time() {}
  Time(uint32_t s, uint32_t ns) : sec(s), nsec(ns) {}
  explicit Time(double t) { sec = (uint32_t)std::floor(t); nsec = (uint32_t)std::round((t - sec) * 1e9); sec += nsec / 1000000000u; nsec %= 1000000000u; }
  static Time now() { return Time(); }
  double toSec() const { return (double)sec + 1e-9 * (double)nsec; }
  uint64_t toNSec() const { return (uint64_t)sec * 1000000000ull + (uint64_t)nsec; }
  bool operator<(const Time& o) const { return sec < o.sec || (sec == o.sec && nsec < o.nsec); }
  bool operator>(const Time& o) const { return o < *this; }
  bool operator==(const Time& o) const { return sec == o.sec && nsec == o.nsec; }
  Duration operator-(const Time& o) const { return Duration((int32_t)sec - (int32_t)o.sec, (int32_t)nsec - (int32_t)o.nsec); }
};
struct Rate { explicit Rate(double) {} void sleep() {} };
}  // namespace ros
#endif
