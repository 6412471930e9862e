// Type check (and, with the CPU oracle bound, a functional smoke run) of the reference-side binding
// ra-slam_amd/host/include/ratsdf/compat/{voxel_tsdf,tsdf_module}.h: the calls below are the ones
// the reference's callers make (main/offline_eval.cc:54-99, modules/renderer_module.cc:56,
// examples/tsdf/offline.cc:90,169-208, examples/scannet_evaluation/eval_one.cc:33,75-82), written
// against the reference's own types.  cv::Mat / Eigen / GLImage8UC4 / SE3 / CameraIntrinsics come
// from tests/cpp/compat_stubs (stand-in declarations: a syntax / type check, not those libraries).
// usage: test_compat_shim <abi library> <prefix>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ratsdf/compat/tsdf_module.h"

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  setenv("RATSDF_LIB", argv[1], 1);
  (void)argv[2];  // the prefix is compiled in (RATSDF_ABI_PREFIX); kept for the log
  const int H = 60, W = 80;
  std::vector<unsigned char> rgb((size_t)H * W * 3, 128);
  std::vector<float> depth((size_t)H * W, 1.5f), ht((size_t)H * W, 0.7f), lt((size_t)H * W, 0.3f);
  const cv::Mat m_rgb(H, W, CV_8UC3, rgb.data()), m_depth(H, W, CV_32FC1, depth.data());
  const cv::Mat m_ht(H, W, CV_32FC1, ht.data()), m_lt(H, W, CV_32FC1, lt.data());
  const CameraIntrinsics<float> K(70.f, 70.f, 39.5f, 29.5f);
  const SE3<float> pose = SE3<float>::Identity();

  // --- TSDFSystem as main/offline_eval.cc uses it ---
  {
    TSDFSystem sys(0.02f, 0.12f, 4.f, K, SE3<float>::Identity());
    for (int i = 0; i < 12; ++i) sys.Integrate(pose, m_rgb, m_depth, m_ht, m_lt);
    sys.Integrate(pose, m_rgb, m_depth);  // no semantics
    sys.engine().Flush();
    const BoundingCube<float> box{-2, 2, -2, 2, 0, 3};
    const std::vector<VoxelSpatialTSDF> q = sys.Query(box);
    if (q.empty()) { fprintf(stderr, "Query returned nothing\n"); return 1; }
    // modules/renderer_module.cc:56
    GLImage8UC4 tsdf_rgba, tsdf_normal;
    tsdf_rgba.BindImage(H, W);
    tsdf_normal.BindImage(H, W);
    const CameraParams virtual_cam(K, H, W);
    sys.Render(virtual_cam, pose, &tsdf_rgba, &tsdf_normal, 6.f);
    sys.Render(virtual_cam, pose, &tsdf_rgba, &tsdf_normal);
    size_t lit = 0;
    for (size_t i = 3; i < tsdf_rgba.pixels_.size(); i += 4) lit += tsdf_rgba.pixels_[i] != 0;
    if (!lit) { fprintf(stderr, "Render produced an empty image\n"); return 1; }
    sys.DownloadAll("/tmp/ratsdf_compat_all.bin");
    sys.DownloadAllMesh("/tmp/ratsdf_compat_v.bin", "/tmp/ratsdf_compat_i.bin", "/tmp/ratsdf_compat_p.bin");
    sys.SetPause(false);
    sys.terminate();
    if (!sys.is_terminated()) return 1;
    printf("TSDFSystem: %zu query records, %zu lit pixels\n", q.size(), lit);
  }
  // --- TSDFGrid as examples/tsdf/offline.cc and eval_one.cc use it ---
  {
    TSDFGrid tsdf(0.02f, 0.12f);
    for (int i = 0; i < 12; ++i) tsdf.Integrate(m_rgb, m_depth, m_ht, m_lt, 4.f, K, pose);
    const auto valid = tsdf.GatherValid();
    const auto sem = tsdf.GatherValidSemantic();
    const auto box = tsdf.GatherVoxels(BoundingCube<float>{-2, 2, -2, 2, 0, 3});
    std::vector<Eigen::Vector3f> v;
    std::vector<Eigen::Vector3i> idx;
    std::vector<float> prob;
    tsdf.GatherValidMesh(&v, &idx, &prob);
    GLImage8UC4 rgba;
    rgba.BindImage(H, W);
    tsdf.RayCast(6.f, CameraParams(K, H, W), pose, &rgba);
    if (valid.empty() || valid.size() != sem.size() || box.empty() || v.size() != prob.size() || idx.empty()) {
      fprintf(stderr, "TSDFGrid outputs inconsistent: %zu %zu %zu %zu %zu %zu\n", valid.size(), sem.size(),
              box.size(), v.size(), prob.size(), idx.size());
      return 1;
    }
    printf("TSDFGrid: %zu voxels, %zu vertices, %zu triangles\n", valid.size(), v.size(), idx.size());
  }
  return 0;
}
