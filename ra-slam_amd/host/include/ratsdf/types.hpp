// types.hpp -- host-side value types with the reference's names and meaning, free of Eigen/OpenCV.
//
//   CameraIntrinsics<T>   utils/cuda/camera.cuh:13-52
//   SE3<T>                utils/cuda/lie_group.cuh:8-45   (Eigen quaternion + translation restated)
//   BoundingCube<T>       utils/tsdf/voxel_tsdf.cuh:19-34
//   VoxelSpatialTSDF(SEGM) utils/tsdf/voxel_types.cuh:46-70
//   Image                 stands in for the cv::Mat arguments (continuous CV_8UC3 / CV_32FC1 data)
//
// The reference's own cv::Mat / Eigen signatures are provided on top of these by
// compat_opencv_eigen.hpp where those libraries exist.
#pragma once
#include <cmath>
#include <cstdint>

#include "../../../../include/ratsdf.h"

namespace ratsdf {

template <typename T>
struct CameraIntrinsics {
  T fx, fy, cx, cy;
  CameraIntrinsics(const T& fx_, const T& fy_, const T& cx_, const T& cy_)
      : fx(fx_), fy(fy_), cx(cx_), cy(cy_) {}
  CameraIntrinsics<T> Inverse() const {  // camera.cuh:35-39
    const T fx_inv = 1 / fx;
    const T fy_inv = 1 / fy;
    return CameraIntrinsics<T>(fx_inv, fy_inv, -cx * fx_inv, -cy * fy_inv);
  }
};

struct CameraParams {  // camera.cuh:54-68
  CameraIntrinsics<float> intrinsics;
  CameraIntrinsics<float> intrinsics_inv;
  int img_h, img_w;
  CameraParams(const CameraIntrinsics<float>& k, int h, int w)
      : intrinsics(k), intrinsics_inv(k.Inverse()), img_h(h), img_w(w) {}
};

template <typename T>
struct Quaternion {
  T x, y, z, w;
};
template <typename T>
struct Vector3 {
  T x, y, z;
};

namespace detail {
template <typename T>
inline Vector3<T> cross(const Vector3<T>& a, const Vector3<T>& b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// Eigen QuaternionBase::_transformVector
template <typename T>
inline Vector3<T> rotate(const Quaternion<T>& q, const Vector3<T>& v) {
  const Vector3<T> qv{q.x, q.y, q.z};
  Vector3<T> uv = cross(qv, v);
  uv.x += uv.x;
  uv.y += uv.y;
  uv.z += uv.z;
  const Vector3<T> c = cross(qv, uv);
  return {(v.x + q.w * uv.x) + c.x, (v.y + q.w * uv.y) + c.y, (v.z + q.w * uv.z) + c.z};
}
// Eigen quaternion product (non-vectorised form)
template <typename T>
inline Quaternion<T> mul(const Quaternion<T>& a, const Quaternion<T>& b) {
  return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
          a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
}  // namespace detail

template <typename T>
class SE3 {
 public:
  SE3() : R_{0, 0, 0, 1}, t_{0, 0, 0} {}
  SE3(const Quaternion<T>& rot, const Vector3<T>& trans) : R_(rot), t_(trans) {}
  // row-major 4x4 (or the top 3x4 of it); Eigen Quaternion(Matrix3) trace method, lie_group.cuh:15-19
  explicit SE3(const T* m, int row_stride = 4) {
    auto M = [&](int r, int c) { return m[r * row_stride + c]; };
    T q[4];
    T t = M(0, 0) + M(1, 1) + M(2, 2);
    if (t > T(0)) {
      t = std::sqrt(t + T(1.0));
      q[3] = T(0.5) * t;
      t = T(0.5) / t;
      q[0] = (M(2, 1) - M(1, 2)) * t;
      q[1] = (M(0, 2) - M(2, 0)) * t;
      q[2] = (M(1, 0) - M(0, 1)) * t;
    } else {
      int i = 0;
      if (M(1, 1) > M(0, 0)) i = 1;
      if (M(2, 2) > M(i, i)) i = 2;
      const int j = (i + 1) % 3, k = (j + 1) % 3;
      t = std::sqrt(M(i, i) - M(j, j) - M(k, k) + T(1.0));
      q[i] = T(0.5) * t;
      t = T(0.5) / t;
      q[3] = (M(k, j) - M(j, k)) * t;
      q[j] = (M(j, i) + M(i, j)) * t;
      q[k] = (M(k, i) + M(i, k)) * t;
    }
    R_ = {q[0], q[1], q[2], q[3]};
    t_ = {M(0, 3), M(1, 3), M(2, 3)};
  }
  static SE3<T> Identity() { return SE3<T>(); }
  SE3<T> Inverse() const {  // lie_group.cuh:25-27
    const T n2 = (R_.x * R_.x + R_.y * R_.y) + (R_.z * R_.z + R_.w * R_.w);
    Quaternion<T> qi{0, 0, 0, 0};
    if (n2 > T(0)) qi = {(-R_.x) / n2, (-R_.y) / n2, (-R_.z) / n2, R_.w / n2};
    return SE3<T>(qi, detail::rotate(qi, Vector3<T>{-t_.x, -t_.y, -t_.z}));
  }
  Quaternion<T> GetR() const { return R_; }
  Vector3<T> GetT() const { return t_; }
  Vector3<T> Apply(const Vector3<T>& v) const {  // lie_group.cuh:33-36
    const Vector3<T> r = detail::rotate(R_, v);
    return {r.x + t_.x, r.y + t_.y, r.z + t_.z};
  }
  SE3<T> operator*(const SE3<T>& o) const {  // lie_group.cuh:38-40
    const Vector3<T> r = detail::rotate(R_, o.t_);
    return SE3<T>(detail::mul(R_, o.R_), Vector3<T>{r.x + t_.x, r.y + t_.y, r.z + t_.z});
  }
  ratsdf_pose abi() const { return ratsdf_pose{R_.x, R_.y, R_.z, R_.w, t_.x, t_.y, t_.z}; }

 private:
  Quaternion<T> R_;
  Vector3<T> t_;
};

template <typename T>
struct BoundingCube {
  T xmin, xmax, ymin, ymax, zmin, zmax;
};

using VoxelSpatialTSDF = ratsdf_voxel_tsdf;          // 16-byte record
using VoxelSpatialTSDFSEGM = ratsdf_voxel_segm;      // 20-byte record

// Continuous image memory, the part of cv::Mat the path uses (voxel_tsdf.cu:419-440).
enum ImageType { kU8C3 = 0, kF32C1 = 1 };
struct Image {
  const void* data = nullptr;
  int rows = 0, cols = 0;
  ImageType type = kF32C1;
  bool empty() const { return data == nullptr || rows <= 0 || cols <= 0; }
  size_t bytes() const { return (size_t)rows * cols * (type == kU8C3 ? 3 : 4); }
};

}  // namespace ratsdf
