// oracle/_ref build shim (TEST INFRASTRUCTURE): utils.h only names these types.
#ifndef ESVO_REF_SHIM_PCL
#define ESVO_REF_SHIM_PCL
#include <memory>
#include <vector>
namespace pcl {
struct PointXYZ { float x = 0, y = 0, z = 0; };
template <class P> struct PointCloud {
  typedef std::shared_ptr<PointCloud<P>> Ptr;
  std::vector<P> points;
};
}  // namespace pcl
#endif
