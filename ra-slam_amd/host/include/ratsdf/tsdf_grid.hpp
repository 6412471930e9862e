// tsdf_grid.hpp -- TSDFGrid with the reference's method names (utils/tsdf/voxel_tsdf.cuh:39-145),
// forwarding to the C ABI of include/ratsdf.h.  The ABI library is bound at run time (dlopen), so
// this host layer builds with g++ alone; the default binding is the in-tree HIP engine.
#pragma once
#include <string>
#include <vector>

#include "types.hpp"

#ifndef RATSDF_ABI_PREFIX
#define RATSDF_ABI_PREFIX "ratsdf_"
#endif

namespace ratsdf {

// Entry points of include/ratsdf.h resolved from one shared library.
struct Api {
  int (*create)(float, float, int, ratsdf_engine**) = nullptr;
  int (*destroy)(ratsdf_engine*) = nullptr;
  int (*integrate)(ratsdf_engine*, const uint8_t*, const float*, const float*, const float*, int,
                   int, float, const ratsdf_intrinsics*, const ratsdf_pose*) = nullptr;
  int (*integrate_batch)(ratsdf_engine*, int, const uint8_t* const*, const float* const*,
                         const float* const*, const float* const*, int, int, float,
                         const ratsdf_intrinsics*, const ratsdf_pose*, int) = nullptr;
  int (*host_alloc)(size_t, void**) = nullptr;
  int (*host_free)(void*) = nullptr;
  int (*synchronize)(ratsdf_engine*) = nullptr;
  int (*recover)(ratsdf_engine*) = nullptr;
  int (*query)(ratsdf_engine*, const ratsdf_bounds*, ratsdf_voxel_tsdf**, size_t*) = nullptr;
  int (*gather_valid)(ratsdf_engine*, ratsdf_voxel_tsdf**, size_t*) = nullptr;
  int (*gather_valid_semantic)(ratsdf_engine*, ratsdf_voxel_segm**, size_t*) = nullptr;
  int (*download_all)(ratsdf_engine*, const char*) = nullptr;
  int (*raycast)(ratsdf_engine*, const ratsdf_intrinsics*, int, int, const ratsdf_pose*, float,
                 uint8_t*, uint8_t*) = nullptr;
  int (*gather_valid_mesh)(ratsdf_engine*, float**, size_t*, int32_t**, size_t*, float**) = nullptr;
  int (*download_all_mesh)(ratsdf_engine*, const char*, const char*, const char*) = nullptr;
  int (*free_buffer)(void*) = nullptr;
  int (*num_active_blocks)(ratsdf_engine*, int32_t*) = nullptr;
  const char* (*status_string)(int) = nullptr;
  const char* (*backend)() = nullptr;
  void* handle = nullptr;

  // path == nullptr: $RATSDF_LIB or libratsdf.so next to this layer.  The symbol prefix is "ratsdf_"
  // (the HIP engine); only test builds of this layer compile with another RATSDF_ABI_PREFIX to bind
  // the CPU oracle's copy of the ABI -- the product binaries have no switch for it.
  static const Api& Load(const char* path = nullptr, const char* prefix = RATSDF_ABI_PREFIX);
};

class TSDFGrid {
 public:
  TSDFGrid(float voxel_size, float truncation, int device = 0, const Api* api = nullptr);
  ~TSDFGrid();
  TSDFGrid(const TSDFGrid&) = delete;
  TSDFGrid& operator=(const TSDFGrid&) = delete;

  // voxel_tsdf.cuh:65-67.  Like the reference (errors.cuh:13-20) failures do not throw; the last
  // status is kept and printed to stderr.
  void Integrate(const Image& img_rgb, const Image& img_depth, const Image& img_ht,
                 const Image& img_lt, float max_depth, const CameraIntrinsics<float>& intrinsics,
                 const SE3<float>& cam_T_world);
  // n frames in one call (ratsdf_integrate_batch): what the TSDFSystem worker does with its queue.
  // ht / lt may be null (all-ones images); `pinned`: every buffer comes from Api::host_alloc
  void IntegrateBatch(int n, const uint8_t* const* rgb, const float* const* depth,
                      const float* const* ht, const float* const* lt, int rows, int cols,
                      float max_depth, const CameraIntrinsics<float>& intrinsics,
                      const SE3<float>* cam_T_world, bool pinned);
  // voxel_tsdf.cuh:78-79; the two uchar4 images go to host buffers (H*W*4 bytes each, may be null)
  // instead of GLImage8UC4 textures
  void RayCast(float max_depth, const CameraParams& virtual_cam, const SE3<float>& cam_T_world,
               uint8_t* tsdf_rgba = nullptr, uint8_t* tsdf_normal = nullptr);
  std::vector<VoxelSpatialTSDF> GatherValid();                                   // :86
  std::vector<VoxelSpatialTSDFSEGM> GatherValidSemantic();                       // :93
  std::vector<VoxelSpatialTSDF> GatherVoxels(const BoundingCube<float>& volumn);  // :102
  // voxel_tsdf.cuh:104-106 (Vector3f / Vector3i buffers become flat float / int arrays, 3 per item)
  void GatherValidMesh(std::vector<float>* vertex_buffer, std::vector<int32_t>* index_buffer,
                       std::vector<float>* vertex_prob_buffer);
  void DownloadAll(const std::string& file_path);  // the write of tsdf_module.cc:57-64
  void DownloadAllMesh(const std::string& vertices_path, const std::string& indices_path,
                       const std::string& prob_path);  // tsdf_module.cc:66-86
  int NumActiveBlock();                            // voxel_hash.cu:225
  // cudaStreamSynchronize(stream_), voxel_tsdf.cu:450: the Integrate* calls do not wait for their frames; this does,
  // and it is where a device error of those frames surfaces (last_status())
  void Synchronize();
  // ratsdf_recover: after a sticky engine error (last_status() != 0 for good) rebuild what is derived from the block
  // directory and clear the error; true when the engine is usable again.  No reference counterpart (it asserts).
  bool Recover();
  int last_status() const { return status_; }
  ratsdf_engine* handle() { return engine_; }
  const Api& api() const { return *api_; }

 private:
  void note(int st, const char* what);
  const Api* api_;
  ratsdf_engine* engine_ = nullptr;
  int status_ = 0;
};

}  // namespace ratsdf
