// stand-in for <pcl/point_types.h> (syntax-only test infrastructure)
#pragma once
namespace pcl {
struct PointXYZ { float x, y, z; };
struct PointXYZI { float x, y, z, intensity; };
struct PointXYZRGB { float x, y, z; unsigned char r, g, b; };
}  // namespace pcl
