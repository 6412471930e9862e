// Exercises the TSDFSystem / TSDFGrid host layer against one ABI library.
//   usage: test_tsdf_system <library.so> <symbol prefix>
// Run by tests/test_host_layer.py with the CPU oracle (no GPU) and with the HIP engine (-m gpu).
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "ratsdf/tsdf_system.hpp"

using namespace ratsdf;

#define CHECK(cond)                                                       \
  do {                                                                    \
    if (!(cond)) {                                                        \
      fprintf(stderr, "CHECK failed at line %d: %s\n", __LINE__, #cond);  \
      exit(1);                                                            \
    }                                                                     \
  } while (0)

struct Frame {
  std::vector<uint8_t> rgb;
  std::vector<float> depth, ht, lt;
  SE3<float> pose;
};

static const int W = 80, H = 60;

static Frame make_frame(int i) {
  Frame f;
  f.rgb.resize((size_t)W * H * 3);
  f.depth.resize((size_t)W * H);
  f.ht.resize((size_t)W * H);
  f.lt.resize((size_t)W * H);
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      const int k = y * W + x;
      f.depth[k] = 1.5f + 0.002f * (float)x + 0.001f * (float)i;
      f.rgb[3 * k] = (uint8_t)(x * 3 + i);
      f.rgb[3 * k + 1] = (uint8_t)(y * 4);
      f.rgb[3 * k + 2] = (uint8_t)(x + y);
      f.ht[k] = 0.2f + 0.6f * (float)x / W;
      f.lt[k] = 1.f - f.ht[k];
    }
  f.pose = SE3<float>(Quaternion<float>{0, 0, 0, 1}, Vector3<float>{0.01f * i, 0, 0});
  return f;
}

static Image img(const std::vector<uint8_t>& v) { return Image{v.data(), H, W, kU8C3}; }
static Image img(const std::vector<float>& v) { return Image{v.data(), H, W, kF32C1}; }

static bool same(const std::vector<VoxelSpatialTSDF>& a, const std::vector<VoxelSpatialTSDF>& b) {
  return a.size() == b.size() &&
         (a.empty() || memcmp(a.data(), b.data(), a.size() * sizeof(VoxelSpatialTSDF)) == 0);
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const Api& api = Api::Load(argv[1], argv[2]);
  printf("backend: %s\n", api.backend());
  const float vs = 0.02f, tr = 0.12f, md = 4.f;
  const CameraIntrinsics<float> K(71.45f, 71.45f, 39.5f, 29.5f);
  std::vector<Frame> frames;
  for (int i = 0; i < 5; ++i) frames.push_back(make_frame(i));

  // 1+2: queue order is preserved and the threaded path equals direct TSDFGrid calls
  TSDFGrid grid(vs, tr, 0, &api);
  CHECK(grid.last_status() == 0);
  for (auto& f : frames) grid.Integrate(img(f.rgb), img(f.depth), img(f.ht), img(f.lt), md, K, f.pose);
  {  // Recover() on a healthy engine changes nothing observable (what it rebuilds is derived from the directory)
    const int before = grid.NumActiveBlock();
    CHECK(grid.Recover());
    CHECK(grid.last_status() == 0 && grid.NumActiveBlock() == before);
  }
  {
    TSDFSystem sys(vs, tr, md, K, SE3<float>::Identity(), 0, &api);
    for (auto& f : frames) {
      Frame copy = f;  // caller buffers may be reused / destroyed right after Integrate returns
      sys.Integrate(copy.pose, img(copy.rgb), img(copy.depth), img(copy.ht), img(copy.lt));
      memset(copy.depth.data(), 0, copy.depth.size() * 4);
    }
    sys.Flush();
    CHECK(sys.frames_integrated() == frames.size());
    CHECK(sys.QueueSize() == 0);
    CHECK(sys.NumActiveBlock() == grid.NumActiveBlock());
    CHECK(sys.NumActiveBlock() > 0);
    const BoundingCube<float> all{-10, 10, -10, 10, -10, 10};
    CHECK(same(sys.Query(all), grid.GatherVoxels(all)));
    const BoundingCube<float> part{-0.3f, 0.3f, -0.3f, 0.3f, 1.0f, 2.0f};
    const auto q = sys.Query(part);
    CHECK(same(q, grid.GatherVoxels(part)));
    CHECK(q.size() < grid.GatherValid().size());
    CHECK(q.size() % 512 == 0);

    // Render == RayCast(2 * max_depth) on the same map (tsdf_module.cc:45-49)
    {
      const CameraParams cam(K, H, W);
      std::vector<uint8_t> a((size_t)W * H * 4), an(a.size()), b(a.size()), bn(a.size());
      for (int rep = 0; rep < 8; ++rep)  // weights must reach 10 before anything is rendered
        for (auto& f : frames) {
          sys.Integrate(f.pose, img(f.rgb), img(f.depth), img(f.ht), img(f.lt));
          grid.Integrate(img(f.rgb), img(f.depth), img(f.ht), img(f.lt), md, K, f.pose);
        }
      sys.Flush();
      sys.Render(cam, frames[0].pose, a.data(), an.data());
      grid.RayCast(2 * md, cam, frames[0].pose, b.data(), bn.data());
      CHECK(a == b && an == bn);
      size_t hit = 0;
      for (size_t i = 3; i < a.size(); i += 4) hit += a[i] == 255;
      CHECK(hit > (size_t)W * H / 4);
      // mesh export through the system == through the grid (DownloadAllMesh, tsdf_module.cc:66-86)
      std::vector<float> mv, mp;
      std::vector<int32_t> mi;
      grid.GatherValidMesh(&mv, &mi, &mp);
      CHECK(mv.size() > 300 && mi.size() > 300 && mp.size() * 3 == mv.size());
      sys.DownloadAllMesh("/tmp/ratsdf_host_v.bin", "/tmp/ratsdf_host_i.bin", "/tmp/ratsdf_host_p.bin");
      FILE* fi = fopen("/tmp/ratsdf_host_i.bin", "rb");
      CHECK(fi != nullptr);
      std::vector<int32_t> fi_data(mi.size());
      CHECK(fread(fi_data.data(), 4, mi.size(), fi) == mi.size());
      CHECK(fgetc(fi) == EOF);
      fclose(fi);
      CHECK(fi_data == mi);
      remove("/tmp/ratsdf_host_v.bin");
      remove("/tmp/ratsdf_host_i.bin");
      remove("/tmp/ratsdf_host_p.bin");
      sys.Render(cam, frames[0].pose, a.data(), nullptr, 0.5f);  // too short to reach the surface
      hit = 0;
      for (size_t i = 3; i < a.size(); i += 4) hit += a[i] == 255;
      CHECK(hit == 0);
    }

    // 3: pause blocks the producer
    sys.SetPause(true);
    std::atomic<bool> returned{false};
    std::thread producer([&] {
      sys.Integrate(frames[0].pose, img(frames[0].rgb), img(frames[0].depth));
      returned = true;
    });
    std::this_thread::sleep_for(std::chrono::milliseconds(100));
    CHECK(!returned.load());
    const size_t done_before = sys.frames_integrated();
    CHECK(done_before >= frames.size());
    sys.SetPause(false);
    producer.join();
    sys.Flush();
    CHECK(returned.load());
    CHECK(sys.frames_integrated() == done_before + 1);

    // 6: terminate twice, frames queued after termination are never integrated
    CHECK(!sys.is_terminated());
    sys.terminate();
    sys.terminate();
    CHECK(sys.is_terminated());
    sys.Integrate(frames[1].pose, img(frames[1].rgb), img(frames[1].depth));
    std::this_thread::sleep_for(std::chrono::milliseconds(20));
    CHECK(sys.frames_integrated() == done_before + 1);
  }  // destructor after terminate(): must not throw / hang

  // 4: missing ht/lt -> all-ones images -> probability stays exactly 0.5
  {
    TSDFSystem sys(vs, tr, md, K, SE3<float>::Identity(), 0, &api);
    for (auto& f : frames) sys.Integrate(f.pose, img(f.rgb), img(f.depth), Image{}, img(f.lt));
    sys.Flush();
    const char* path = "/tmp/ratsdf_host_test_download.bin";
    sys.DownloadAll(path);
    FILE* fp = fopen(path, "rb");
    CHECK(fp != nullptr);
    fseek(fp, 0, SEEK_END);
    const long bytes = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    CHECK(bytes > 0 && bytes % (long)sizeof(VoxelSpatialTSDFSEGM) == 0);
    std::vector<VoxelSpatialTSDFSEGM> rec((size_t)bytes / sizeof(VoxelSpatialTSDFSEGM));
    CHECK(fread(rec.data(), sizeof(VoxelSpatialTSDFSEGM), rec.size(), fp) == rec.size());
    fclose(fp);
    remove(path);
    CHECK((long)rec.size() == (long)sys.NumActiveBlock() * 512);
    for (auto& r : rec) CHECK(r.prob == 0.5f);
  }

  // 7: extrinsics are composed as cam_T_posecam * posecam_T_world
  {
    const float c = std::cos(0.1f), s = std::sin(0.1f);
    const float m[16] = {c, 0, s, 0.05f, 0, 1, 0, -0.02f, -s, 0, c, 0.01f, 0, 0, 0, 1};
    const SE3<float> E(m);
    TSDFSystem sys(vs, tr, md, K, E, 0, &api);
    TSDFGrid ref(vs, tr, 0, &api);
    for (auto& f : frames) {
      sys.Integrate(f.pose, img(f.rgb), img(f.depth), img(f.ht), img(f.lt));
      ref.Integrate(img(f.rgb), img(f.depth), img(f.ht), img(f.lt), md, K, E * f.pose);
    }
    sys.Flush();
    const BoundingCube<float> all{-10, 10, -10, 10, -10, 10};
    CHECK(same(sys.Query(all), ref.GatherVoxels(all)));
    // SE3 round trip
    const SE3<float> I = E * E.Inverse();
    CHECK(std::fabs(I.GetR().w) > 0.999999f && std::fabs(I.GetT().x) < 1e-6f);
  }

  // 8: the queue's page-locked block pool, bounded queue (the default).  The first frame reserves what a running
  // system keeps in flight (two batches of 32 + an arena = 72 blocks = 9 arenas); 100 frames pushed as fast as
  // possible never allocate or free anything further -- the producer waits for the worker instead (the round-4 pool
  // parked 52 blocks and then freed on EVERY release) -- and the map is the frame-by-frame one.
  {
    TSDFSystem sys(vs, tr, md, K, SE3<float>::Identity(), 0, &api);
    TSDFGrid ref(vs, tr, 0, &api);
    auto load = [&]() {
      for (int i = 0; i < 100; ++i) {
        const Frame& f = frames[(size_t)i % frames.size()];
        sys.Integrate(f.pose, img(f.rgb), img(f.depth), img(f.ht), img(f.lt));
        ref.Integrate(img(f.rgb), img(f.depth), img(f.ht), img(f.lt), md, K, f.pose);
      }
      sys.Flush();
    };
    sys.Integrate(frames[0].pose, img(frames[0].rgb), img(frames[0].depth), img(frames[0].ht), img(frames[0].lt));
    ref.Integrate(img(frames[0].rgb), img(frames[0].depth), img(frames[0].ht), img(frames[0].lt), md, K, frames[0].pose);
    const size_t reserve = sys.pool_system_allocs();
    CHECK(reserve == (2 * 32 + HostBlockPool::kArenaBlocks) / HostBlockPool::kArenaBlocks);  // 72 blocks = 9 arenas
    load();
    load();
    CHECK(sys.pool_system_allocs() == reserve && sys.pool_system_frees() == 0 && sys.pool_pageable_blocks() == 0);
    CHECK(sys.frames_integrated() == 201);
    CHECK(sys.QueueSize() == 0);
    const BoundingCube<float> all{-10, 10, -10, 10, -10, 10};
    CHECK(same(sys.Query(all), ref.GatherVoxels(all)));
    // a producer that waits for a free block is released by terminate() (nobody would ever free one): no deadlock
    std::atomic<int> pushed{0};
    std::thread producer([&] {
      for (int i = 0; i < 2000; ++i) {
        const Frame& f = frames[(size_t)i % frames.size()];
        sys.Integrate(f.pose, img(f.rgb), img(f.depth), img(f.ht), img(f.lt));
        ++pushed;
      }
    });
    std::this_thread::sleep_for(std::chrono::milliseconds(30));
    sys.terminate();
    producer.join();
    CHECK(pushed.load() == 2000);
  }

  // 9: SetQueueBounded(false) = the reference's unbounded queue (tsdf_module.cc:99-100): a queue that outgrows its
  // page-locked budget continues in ordinary memory.  The budget here is 12 blocks and 300 frames are pushed as fast
  // as possible; frames of both kinds of memory reach the engine in order (separate calls): same map as frame by frame
  {
    TSDFSystem sys(vs, tr, md, K, SE3<float>::Identity(), 0, &api);
    sys.SetPinnedBudget((size_t)12 * W * H * 16);
    sys.SetQueueBounded(false);
    TSDFGrid ref(vs, tr, 0, &api);
    for (int i = 0; i < 300; ++i) {
      const Frame& f = frames[(size_t)i % frames.size()];
      sys.Integrate(f.pose, img(f.rgb), img(f.depth), img(f.ht), img(f.lt));
    }
    for (int i = 0; i < 300; ++i) {
      const Frame& f = frames[(size_t)i % frames.size()];
      ref.Integrate(img(f.rgb), img(f.depth), img(f.ht), img(f.lt), md, K, f.pose);
    }
    sys.Flush();
    CHECK(sys.frames_integrated() == 300);
    CHECK(sys.pool_system_allocs() <= 12);
    if (std::string(api.backend()) == "cpu-oracle") CHECK(sys.pool_pageable_blocks() > 0);  // (the oracle is slow enough)
    const BoundingCube<float> all{-10, 10, -10, 10, -10, 10};
    CHECK(same(sys.Query(all), ref.GatherVoxels(all)));
  }

  // bad arguments are reported, not fatal (the reference only asserts)
  {
    TSDFGrid g(vs, tr, 0, &api);
    std::vector<float> small((size_t)10 * 10, 1.f);
    g.Integrate(img(frames[0].rgb), Image{small.data(), 10, 10, kF32C1}, Image{}, Image{}, md, K,
                frames[0].pose);
    CHECK(g.last_status() == RATSDF_ERR_BAD_ARGUMENT);
    CHECK(g.NumActiveBlock() == 0);
  }
  printf("OK\n");
  return 0;
}
