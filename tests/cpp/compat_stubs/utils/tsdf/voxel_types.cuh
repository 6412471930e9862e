// stand-in declarations of the reference's utils/tsdf/voxel_types.cuh:46-70 (type check only)
#pragma once
#include <Eigen/Dense>
class VoxelSpatialTSDF {
 public:
  Eigen::Vector3f position;
  float tsdf;
};
class VoxelSpatialTSDFSEGM {
 public:
  Eigen::Vector3f position;
  float tsdf;
  float probability;
};
