// oracle/_ref build shim for the mapper NODE (TEST INFRASTRUCTURE)
#ifndef ESVO_REF_SHIM_PCL
#define ESVO_REF_SHIM_PCL
#include <std_msgs/Header.h>
#include <memory>
#include <vector>
namespace pcl {
struct PointXYZ { float x = 0, y = 0, z = 0; PointXYZ() {} PointXYZ(float a, float b, float c) : x(a), y(b), z(c) {} };
struct PCLHeader { std::string frame_id; unsigned long long stamp = 0; };
template <class P> struct PointCloud {
  typedef std::shared_ptr<PointCloud<P>> Ptr;
  typedef typename std::vector<P>::iterator iterator;
  PCLHeader header;
  std::vector<P> points;
  void clear() { points.clear(); }
  void reserve(size_t n) { points.reserve(n); }
  void push_back(const P& p) { points.push_back(p); }
  size_t size() const { return points.size(); }
  bool empty() const { return points.empty(); }
  iterator begin() { return points.begin(); }
  iterator end() { return points.end(); }
  template <class It> void insert(iterator pos, It a, It b) { points.insert(pos, a, b); }
};
template <class C> void toROSMsg(const C&, sensor_msgs::PointCloud2&) {}
// the global-cloud voxel filter is a publisher detail (PCL: third-party); the stand-in passes the cloud through
template <class P> struct VoxelGrid {
  typename PointCloud<P>::Ptr in;
  void setInputCloud(const typename PointCloud<P>::Ptr& c) { in = c; }
  void setLeafSize(float, float, float) {}
  void filter(PointCloud<P>& out) { out = *in; }
};
}  // namespace pcl
#endif
