// stand-in for <pcl/point_cloud.h> (syntax-only test infrastructure)
#pragma once
#include <memory>
#include <vector>
namespace pcl {
template <class P> struct PointCloud {
  std::vector<P> points;
  unsigned width = 0, height = 0;
  using Ptr = std::shared_ptr<PointCloud<P>>;
  size_t size() const { return points.size(); }
};
}  // namespace pcl
// PCL pulls Eigen in; utils/utils.h names Eigen::Vector3d / Eigen::aligned_allocator in one declaration
namespace Eigen {
struct Vector3d { double v[3]; };
template <class T> using aligned_allocator = std::allocator<T>;
}  // namespace Eigen
