// compat/tsdf_module.h -- reference-side binding: `TSDFSystem` with the reference's own signatures
// (modules/tsdf_module.h:37-165) on top of ratsdf::TSDFSystem.  It takes the place of
// modules/tsdf_module.h in the RA-SLAM tree; main/offline_eval.cc, disinfect_slam/disinfect_slam.cc
// and modules/renderer_module.cc (Render(..., &tsdf_rgba_, &tsdf_normal_, depth), :56) keep their
// calls unchanged.  See compat/voxel_tsdf.h for the dependencies and for how the two rendered images
// reach their GLImage8UC4.
#pragma once
#include <string>
#include <vector>

#include "ratsdf/compat/voxel_tsdf.h"
#include "ratsdf/tsdf_system.hpp"

class TSDFSystem {
 public:
  TSDFSystem(float voxel_size, float truncation, float max_depth,
             const CameraIntrinsics<float>& intrinsics,
             const SE3<float>& extrinsics = SE3<float>::Identity())
      : impl_(voxel_size, truncation, max_depth, ratsdf::compat::intrinsics(intrinsics),
              ratsdf::compat::pose(extrinsics)),
        max_depth_(max_depth) {}

  void Integrate(const SE3<float>& posecam_T_world, const cv::Mat& img_rgb, const cv::Mat& img_depth,
                 const cv::Mat& img_ht = {}, const cv::Mat& img_lt = {}) {
    namespace c = ratsdf::compat;
    impl_.Integrate(c::pose(posecam_T_world), c::view(img_rgb, ratsdf::kU8C3),
                    c::view(img_depth, ratsdf::kF32C1), c::view(img_ht, ratsdf::kF32C1),
                    c::view(img_lt, ratsdf::kF32C1));
  }

  std::vector<VoxelSpatialTSDF> Query(const BoundingCube<float>& v) {
    return ratsdf::compat::records<VoxelSpatialTSDF>(
        impl_.Query({v.xmin, v.xmax, v.ymin, v.ymax, v.zmin, v.zmax}));
  }

  // tsdf_module.h:87-88 / tsdf_module.cc:45-49: ray cast to twice the integration depth
  void Render(const CameraParams& virtual_cam, const SE3<float> cam_T_world, GLImage8UC4* img_rgba,
              GLImage8UC4* img_normal) {
    Render(virtual_cam, cam_T_world, img_rgba, img_normal, max_depth_ * 2);
  }
  // tsdf_module.h:98-99
  void Render(const CameraParams& virtual_cam, const SE3<float> cam_T_world, GLImage8UC4* img_rgba,
              GLImage8UC4* img_normal, float max_depth) {
    namespace c = ratsdf::compat;
    const size_t bytes = (size_t)virtual_cam.img_h * virtual_cam.img_w * 4;
    if (img_rgba) rgba_.resize(bytes);
    if (img_normal) normal_.resize(bytes);
    impl_.Render(c::camera(virtual_cam), c::pose(cam_T_world), img_rgba ? rgba_.data() : nullptr,
                 img_normal ? normal_.data() : nullptr, max_depth);
    if (img_rgba) RATSDF_GL_UPLOAD(img_rgba, rgba_.data());      // GL calls: the caller's (GUI) thread,
    if (img_normal) RATSDF_GL_UPLOAD(img_normal, normal_.data());  // as with the reference's LoadCuda
  }

  void DownloadAll(const std::string& file_path) { impl_.DownloadAll(file_path); }
  void DownloadAllMesh(const std::string& vertices_path, const std::string& indices_path,
                       const std::string& prob_path) {
    impl_.DownloadAllMesh(vertices_path, indices_path, prob_path);
  }
  bool is_terminated() { return impl_.is_terminated(); }
  void terminate() { impl_.terminate(); }
  void SetPause(bool pause) { impl_.SetPause(pause); }

  ratsdf::TSDFSystem& engine() { return impl_; }  // beyond the reference: Flush(), QueueSize(), ...

 private:
  ratsdf::TSDFSystem impl_;
  float max_depth_;
  std::vector<uint8_t> rgba_, normal_;  // Render is serialised by its caller (one GUI thread)
};
