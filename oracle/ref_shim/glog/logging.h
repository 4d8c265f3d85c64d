// oracle/_ref build shim (TEST INFRASTRUCTURE): LOG(x) << ... is swallowed.
#ifndef ESVO_REF_SHIM_GLOG
#define ESVO_REF_SHIM_GLOG
#include <iostream>
namespace esvo_ref_shim {
struct NullLog {
  template <class T> NullLog& operator<<(const T&) { return *this; }
  NullLog& operator<<(std::ostream& (*)(std::ostream&)) { return *this; }
};
}  // namespace esvo_ref_shim
#define LOG(severity) ::esvo_ref_shim::NullLog()
#endif
