// ratsdf_system_bench -- throughput of the drop-in calling convention itself:
//   TSDFSystem::Integrate(pose, rgb, depth, ht, lt) from pageable host images, as main/offline_eval.cc:64-87
//   calls it (modules/tsdf_module.cc:22-37: deep copy into the queue; :88-115: the worker thread), then Flush().
// Frames come from a file written by bench.py (synthetic stream): int32 H, W, n; per frame 7 floats pose
// (qx qy qz qw tx ty tz), 4 floats intrinsics, depth f32[H*W], ht f32[H*W], lt f32[H*W], rgb u8[H*W*3].
//
//   ratsdf_system_bench <frames.bin> [--lib libratsdf.so] [--frames N] [--voxel 0.005] [--max-depth 4]
//                       [--no-sem] [--device 0]
// Prints one JSON line.  The images handed to Integrate live in ordinary (pageable) std::vector memory and
// every call passes a DIFFERENT buffer than the one before (a ring of the file's frames), like a reader's.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "ratsdf/tsdf_system.hpp"

using namespace ratsdf;

int main(int argc, char** argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s frames.bin [--lib L] [--frames N] [--voxel V] [--max-depth D] [--no-sem]\n", argv[0]);
    return 2;
  }
  const char* lib = nullptr;
  int total = 2000, device = 0;
  float voxel = 0.005f, max_depth = 4.f;
  bool sem = true;
  for (int i = 2; i < argc; ++i) {
    const std::string a = argv[i];
    auto next = [&]() { return i + 1 < argc ? argv[++i] : (fprintf(stderr, "missing value\n"), exit(2), argv[0]); };
    if (a == "--lib") lib = next();
    else if (a == "--frames") total = atoi(next());
    else if (a == "--voxel") voxel = strtof(next(), nullptr);
    else if (a == "--max-depth") max_depth = strtof(next(), nullptr);
    else if (a == "--device") device = atoi(next());
    else if (a == "--no-sem") sem = false;
  }
  std::ifstream in(argv[1], std::ios::binary);
  int32_t hdr[3];
  if (!in.read(reinterpret_cast<char*>(hdr), 12) || hdr[0] <= 0 || hdr[1] <= 0 || hdr[2] <= 0) {
    fprintf(stderr, "bad frame file\n");
    return 1;
  }
  const int H = hdr[0], W = hdr[1], n = hdr[2];
  const size_t npix = (size_t)H * W;
  struct Fr {
    float pose[7], intr[4];
    std::vector<float> depth, ht, lt;
    std::vector<uint8_t> rgb;
  };
  std::vector<Fr> fr((size_t)n);
  for (auto& f : fr) {
    f.depth.resize(npix);
    f.ht.resize(npix);
    f.lt.resize(npix);
    f.rgb.resize(npix * 3);
    in.read(reinterpret_cast<char*>(f.pose), 28);
    in.read(reinterpret_cast<char*>(f.intr), 16);
    in.read(reinterpret_cast<char*>(f.depth.data()), (std::streamsize)npix * 4);
    in.read(reinterpret_cast<char*>(f.ht.data()), (std::streamsize)npix * 4);
    in.read(reinterpret_cast<char*>(f.lt.data()), (std::streamsize)npix * 4);
    in.read(reinterpret_cast<char*>(f.rgb.data()), (std::streamsize)npix * 3);
    if (!in) {
      fprintf(stderr, "truncated frame file\n");
      return 1;
    }
  }
  const CameraIntrinsics<float> K(fr[0].intr[0], fr[0].intr[1], fr[0].intr[2], fr[0].intr[3]);
  TSDFSystem sys(voxel, voxel * 6, max_depth, K, SE3<float>::Identity(), device, &Api::Load(lib));
  auto push = [&](const Fr& f) {
    const SE3<float> p(Quaternion<float>{f.pose[0], f.pose[1], f.pose[2], f.pose[3]},
                       Vector3<float>{f.pose[4], f.pose[5], f.pose[6]});
    const Image rgb{f.rgb.data(), H, W, kU8C3}, depth{f.depth.data(), H, W, kF32C1};
    if (sem) sys.Integrate(p, rgb, depth, Image{f.ht.data(), H, W, kF32C1}, Image{f.lt.data(), H, W, kF32C1});
    else sys.Integrate(p, rgb, depth);
  };
  for (int i = 0; i < 2 * n && i < 64; ++i) push(fr[(size_t)(i % n)]);  // warm-up: map built, pools filled
  sys.Flush();
  size_t max_queue = 0;
  const size_t allocs0 = sys.pool_system_allocs(), frees0 = sys.pool_system_frees(), pageable0 = sys.pool_pageable_blocks();
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < total; ++i) {
    const int k = i % (2 * n);
    push(fr[(size_t)(k < n ? k : 2 * n - 1 - k)]);  // ping-pong like bench.py's stream
    if ((i & 63) == 0) max_queue = std::max(max_queue, sys.QueueSize());
  }
  const double t_push = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  sys.Flush();
  const double t_all = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  // page-locked allocations / frees of the queue's block pool INSIDE the timed region (0 / 0 once the pool is warm)
  const size_t pool_allocs = sys.pool_system_allocs() - allocs0, pool_frees = sys.pool_system_frees() - frees0;
  const double bytes = (double)npix * (sem ? 15.0 : 7.0);
  printf("{\"frames\": %d, \"seconds\": %.4f, \"frames_per_s\": %.1f, \"producer_seconds\": %.4f, "
         "\"producer_frames_per_s\": %.1f, \"h2d_gbps\": %.2f, \"max_queue\": %zu, \"semantics\": %s, "
         "\"width\": %d, \"height\": %d, \"active_blocks\": %d, \"pool_allocs_steady_state\": %zu, "
         "\"pool_frees_steady_state\": %zu, \"frames_queued_in_pageable_memory\": %zu}\n",
         total, t_all, total / t_all, t_push, total / t_push, total * bytes / t_all / 1e9, max_queue,
         sem ? "true" : "false", W, H, sys.NumActiveBlock(), pool_allocs, pool_frees,
         sys.pool_pageable_blocks() - pageable0);
  sys.terminate();
  return 0;
}
