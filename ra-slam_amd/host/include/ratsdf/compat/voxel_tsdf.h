// compat/voxel_tsdf.h -- reference-side binding: `TSDFGrid` with the reference's own signatures
// (utils/tsdf/voxel_tsdf.cuh:19-106: cv::Mat images, the reference's SE3<float> /
// CameraIntrinsics<float> / CameraParams, GLImage8UC4 render targets, Eigen mesh buffers) on top of
// ratsdf::TSDFGrid, i.e. of the C ABI in include/ratsdf.h.
//
// It replaces utils/tsdf/voxel_tsdf.cuh in the RA-SLAM tree, so that the direct TSDFGrid callers
// (examples/tsdf/offline.cc:90,169-208, examples/scannet_evaluation/eval_one.cc:33,75,82) compile
// unchanged.  Needs what those callers already need: OpenCV core, Eigen, and the reference's
// utils/cuda/{camera,lie_group}.cuh, utils/gl/image.h, utils/tsdf/voxel_types.cuh.  None of them
// exists in the build container, where this header is only type-checked against minimal stand-in
// declarations (tests/test_compat_shim.py).
//
// Rendering: the reference writes the two ray-cast images into OpenGL textures through CUDA-GL
// interop (GLImageBase::LoadCuda, utils/gl/image.cc:108-119).  Here they are ray cast into host
// buffers and handed to RATSDF_GL_UPLOAD(image, rgba_bytes), by default image->LoadHost(bytes): the
// ~5-line method a maintainer adds next to LoadCuda (glBindTexture + glTexSubImage2D with
// GL_RGBA / GL_UNSIGNED_BYTE).
#pragma once
#include <Eigen/Dense>
#include <opencv2/core.hpp>

#include <cassert>
#include <cstdint>
#include <cstring>
#include <vector>

#include "utils/cuda/camera.cuh"
#include "utils/cuda/lie_group.cuh"
#include "utils/gl/image.h"
#include "utils/tsdf/voxel_types.cuh"

#include "ratsdf/tsdf_grid.hpp"

#ifndef RATSDF_GL_UPLOAD
#define RATSDF_GL_UPLOAD(image, bytes) (image)->LoadHost(bytes)
#endif

// voxel_tsdf.cuh:19-34 (this header takes that file's place)
template <typename T>
struct BoundingCube {
  T xmin;
  T xmax;
  T ymin;
  T ymax;
  T zmin;
  T zmax;

  template <typename Tout = T>
  BoundingCube<Tout> Scale(T scale) const {
    return BoundingCube<Tout>({static_cast<Tout>(xmin * scale), static_cast<Tout>(xmax * scale),
                               static_cast<Tout>(ymin * scale), static_cast<Tout>(ymax * scale),
                               static_cast<Tout>(zmin * scale), static_cast<Tout>(zmax * scale)});
  }
};

namespace ratsdf {
namespace compat {

static_assert(sizeof(::VoxelSpatialTSDF) == sizeof(ratsdf_voxel_tsdf), "16-byte record");
static_assert(sizeof(::VoxelSpatialTSDFSEGM) == sizeof(ratsdf_voxel_segm), "20-byte record");

inline ratsdf::SE3<float> pose(const ::SE3<float>& T) {  // quaternion + translation, as they are
  const auto q = T.GetR();
  const auto t = T.GetT();
  return ratsdf::SE3<float>({q.x(), q.y(), q.z(), q.w()}, {t[0], t[1], t[2]});
}
inline ratsdf::CameraIntrinsics<float> intrinsics(const ::CameraIntrinsics<float>& K) {
  return ratsdf::CameraIntrinsics<float>(K.fx, K.fy, K.cx, K.cy);
}
inline ratsdf::CameraParams camera(const ::CameraParams& c) {
  return ratsdf::CameraParams(intrinsics(c.intrinsics), c.img_h, c.img_w);
}
// the asserts of TSDFGrid::Integrate (voxel_tsdf.cu:419-428): type and continuity
inline ratsdf::Image view(const cv::Mat& m, ratsdf::ImageType t) {
  if (m.empty()) return ratsdf::Image{};
  assert(m.isContinuous());
  assert(m.type() == (t == ratsdf::kU8C3 ? CV_8UC3 : CV_32FC1));
  return ratsdf::Image{m.data, m.rows, m.cols, t};
}
template <typename Rec, typename Abi>
inline std::vector<Rec> records(const std::vector<Abi>& v) {  // same bytes, the reference's class
  std::vector<Rec> out(v.size());
  if (!v.empty()) std::memcpy(static_cast<void*>(out.data()), v.data(), v.size() * sizeof(Abi));
  return out;
}

}  // namespace compat
}  // namespace ratsdf

class TSDFGrid {
 public:
  TSDFGrid(float voxel_size, float truncation) : impl_(voxel_size, truncation) {}

  void Integrate(const cv::Mat& img_rgb, const cv::Mat& img_depth, const cv::Mat& img_ht,
                 const cv::Mat& img_lt, float max_depth, const CameraIntrinsics<float>& intrinsics,
                 const SE3<float>& cam_T_world) {
    namespace c = ratsdf::compat;
    assert(img_rgb.rows == img_depth.rows && img_rgb.cols == img_depth.cols);
    impl_.Integrate(c::view(img_rgb, ratsdf::kU8C3), c::view(img_depth, ratsdf::kF32C1),
                    c::view(img_ht, ratsdf::kF32C1), c::view(img_lt, ratsdf::kF32C1), max_depth,
                    c::intrinsics(intrinsics), c::pose(cam_T_world));
  }

  void RayCast(float max_depth, const CameraParams& virtual_cam, const SE3<float>& cam_T_world,
               GLImage8UC4* tsdf_rgba = NULL, GLImage8UC4* tsdf_normal = NULL) {
    namespace c = ratsdf::compat;
    const size_t bytes = (size_t)virtual_cam.img_h * virtual_cam.img_w * 4;
    if (tsdf_rgba) rgba_.resize(bytes);
    if (tsdf_normal) normal_.resize(bytes);
    impl_.RayCast(max_depth, c::camera(virtual_cam), c::pose(cam_T_world),
                  tsdf_rgba ? rgba_.data() : nullptr, tsdf_normal ? normal_.data() : nullptr);
    if (tsdf_rgba) RATSDF_GL_UPLOAD(tsdf_rgba, rgba_.data());
    if (tsdf_normal) RATSDF_GL_UPLOAD(tsdf_normal, normal_.data());
  }

  std::vector<VoxelSpatialTSDF> GatherValid() {
    return ratsdf::compat::records<VoxelSpatialTSDF>(impl_.GatherValid());
  }
  std::vector<VoxelSpatialTSDFSEGM> GatherValidSemantic() {
    return ratsdf::compat::records<VoxelSpatialTSDFSEGM>(impl_.GatherValidSemantic());
  }
  std::vector<VoxelSpatialTSDF> GatherVoxels(const BoundingCube<float>& v) {
    return ratsdf::compat::records<VoxelSpatialTSDF>(
        impl_.GatherVoxels({v.xmin, v.xmax, v.ymin, v.ymax, v.zmin, v.zmax}));
  }
  void GatherValidMesh(std::vector<Eigen::Vector3f>* vertex_buffer,
                       std::vector<Eigen::Vector3i>* index_buffer,
                       std::vector<float>* vertex_prob_buffer) {
    std::vector<float> v;
    std::vector<int32_t> i;
    impl_.GatherValidMesh(&v, &i, vertex_prob_buffer);
    vertex_buffer->resize(v.size() / 3);
    index_buffer->resize(i.size() / 3);
    for (size_t k = 0; k < vertex_buffer->size(); ++k)
      (*vertex_buffer)[k] = Eigen::Vector3f(v[3 * k], v[3 * k + 1], v[3 * k + 2]);
    for (size_t k = 0; k < index_buffer->size(); ++k)
      (*index_buffer)[k] = Eigen::Vector3i(i[3 * k], i[3 * k + 1], i[3 * k + 2]);
  }

  ratsdf::TSDFGrid& engine() { return impl_; }  // beyond the reference: the ratsdf object underneath

 private:
  ratsdf::TSDFGrid impl_;
  std::vector<uint8_t> rgba_, normal_;
};
