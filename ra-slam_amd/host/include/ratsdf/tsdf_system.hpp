// tsdf_system.hpp -- TSDFSystem: the threaded front of the TSDF map, same public surface and
// locking as the reference (modules/tsdf_module.h:37-165, modules/tsdf_module.cc:7-131).
//
//   * the constructor starts the integration thread                         (tsdf_module.cc:7-12)
//   * Integrate() may be called from any thread, blocks while paused, deep-copies its images into
//     the queue (callers may reuse their buffers at once), substitutes all-ones ht/lt images when
//     either is empty and composes cam_T_posecam * posecam_T_world          (tsdf_module.cc:22-37)
//   * the worker integrates under mtx_read_; Query / DownloadAll take the same mutex, so they are
//     serialised against integration and each other                         (tsdf_module.cc:39-64,88-115)
//   * terminate() stops the worker at the top of its loop, DISCARDING queued frames, exactly like
//     the reference (tsdf_module.cc:91-94,119-125).  Knowingly fixed: terminate() is idempotent
//     (the reference's destructor joins a second time and would throw), and Flush() is added so a
//     harness can wait for the queue to drain first.  Knowingly different: the queue is BOUNDED by default (its
//     elements live in a reserve of page-locked blocks; Integrate() waits for the worker when all are in flight --
//     the reference's queue grows without bound, tsdf_module.cc:99-100); SetQueueBounded(false) restores that.
//   * Render writes into host buffers instead of OpenGL textures.
#pragma once
#include <condition_variable>
#include <memory>
#include <mutex>
#include <queue>
#include <string>
#include <thread>
#include <vector>

#include "tsdf_grid.hpp"

namespace ratsdf {

struct HostBlock {
  void* ptr = nullptr;
  size_t bytes = 0;
  bool pinned = true;  // false: page-locked memory ran out, the block is ordinary memory (the engine stages it)
};

// Page-locked blocks (Api::host_alloc) recycled between frames: the queue's deep copies live in them
// so that the worker uploads without a second copy.
//   * Blocks come from ARENAS of kArenaBlocks consecutive blocks (one host_alloc each) and the free block of
//     lowest address goes out first, so the frames of a batch usually lie side by side in memory and the
//     engine uploads runs of them with one copy (include/ratsdf.h, ratsdf_integrate_batch).
//   * The first block of a size reserves `reserve_blocks` of them at once (what a running system has in flight:
//     two batches + a short queue), so that a steady stream never allocates: page-locking memory takes
//     milliseconds per arena and holds up every other HIP call of the process while it runs -- measured in round
//     5: 28 arenas allocated while the queue grew cost the worker half its rate.
//   * Page-locked memory is bounded (kPinnedBytes): a queue that outgrows it -- the reference's queue is unbounded,
//     modules/tsdf_module.cc:99-100 only warns -- continues in ordinary memory (pinned = false; the worker hands
//     such frames over as pageable), freed on release.
//   * Nothing page-locked is given back while frames flow: release() parks.  trim() -- called by the worker when
//     its queue has drained -- returns whole idle arenas beyond max(kParkedBytes, the reserve).  (Until round 5
//     release() itself enforced a 256 MiB cap = 52 blocks at 640x480, below the 2 x 32 frames + queue a running
//     system has in flight: every further frame was a hipHostMalloc + hipHostFree pair.)
//   * A failed page-locked allocation is not fatal: ordinary memory, as above.
class HostBlockPool {
 public:
  static constexpr size_t kArenaBlocks = 8;
  static constexpr size_t kParkedBytes = (size_t)256 << 20;
  static constexpr size_t kPinnedBytes = (size_t)768 << 20;
  explicit HostBlockPool(const Api* api, size_t reserve_blocks = 0) : api_(api), reserve_(reserve_blocks) {}
  void set_pinned_budget(size_t bytes) { budget_ = bytes; }  // before the first acquire()
  ~HostBlockPool();
  HostBlock acquire(size_t bytes, bool grow_ok = true, bool pageable_ok = true);
  void release(const HostBlock& b);
  bool wait_for_release(int ms);  // true: a block came back since the call began
  void trim();
  // host_alloc / host_free calls so far (a steady stream of equal-sized frames must not move them)
  size_t system_allocs() const { return allocs_; }
  size_t system_frees() const { return frees_; }
  size_t pageable_blocks() const { return pageable_; }  // blocks handed out in ordinary memory so far

 private:
  struct Arena {
    uint8_t* base = nullptr;
    size_t block_bytes = 0, blocks = 0, in_use = 0;
  };
  bool grow(size_t bytes);  // one more arena of `bytes`-sized blocks; false: cap reached or no page-locked memory
  Arena* arena_of(const void* p);
  const Api* api_;
  size_t reserve_;
  std::mutex mtx_;
  std::vector<Arena> arenas_;
  std::vector<HostBlock> free_;  // page-locked, sorted by address, descending (the lowest is at the back)
  size_t pinned_bytes_ = 0, reserved_for_ = 0, budget_ = kPinnedBytes;
  size_t allocs_ = 0, frees_ = 0, pageable_ = 0, releases_ = 0;
  std::condition_variable cv_release_;
};

// The deep copy of a frame into its queue element (cv::Mat::clone x 4, tsdf_module.cc:28-35: 4.6 MB at
// 640x480) is what bounds the calling convention once the engine integrates a frame in 30 us: one core
// copies 10-12 GB/s = 2.5 k frames/s.  Integrate() therefore splits the copy over a few helper threads
// (the caller takes a share itself and returns when all shares are done, so the contract -- the
// caller's buffers are free on return -- is unchanged).
class ParallelCopier {
 public:
  explicit ParallelCopier(int helpers);
  ~ParallelCopier();
  // copies every (dst, src, bytes) piece; returns when all are done.  One caller at a time.
  void copy(const void* const* src, void* const* dst, const size_t* bytes, int n);

 private:
  struct Piece {
    void* dst;
    const void* src;
    size_t bytes;
  };
  void worker();
  std::mutex mtx_;
  std::condition_variable cv_work_, cv_done_;
  std::vector<Piece> pieces_;
  size_t next_ = 0, done_ = 0;
  bool stop_ = false;
  std::vector<std::thread> threads_;
};

struct TSDFSystemInput {  // tsdf_module.h:19-35, images owned by the queue element
  SE3<float> cam_T_world;
  int rows = 0, cols = 0;
  HostBlock block;       // [depth f32 | ht f32 | lt f32 | rgb u8x3], rows * cols * 16 bytes
  bool has_sem = false;  // ht / lt given (else: all ones, tsdf_module.cc:29-31)
};

class TSDFSystem {
 public:
  TSDFSystem(float voxel_size, float truncation, float max_depth,
             const CameraIntrinsics<float>& intrinsics,
             const SE3<float>& extrinsics = SE3<float>::Identity(), int device = 0,
             const Api* api = nullptr);
  ~TSDFSystem();

  void Integrate(const SE3<float>& posecam_T_world, const Image& img_rgb, const Image& img_depth,
                 const Image& img_ht = {}, const Image& img_lt = {});
  std::vector<VoxelSpatialTSDF> Query(const BoundingCube<float>& volumn);
  // tsdf_module.h:87-99: rgba / normal-shaded views of the map; host buffers replace GLImage8UC4
  void Render(const CameraParams& virtual_cam, const SE3<float> cam_T_world, uint8_t* img_rgba,
              uint8_t* img_normal);
  void Render(const CameraParams& virtual_cam, const SE3<float> cam_T_world, uint8_t* img_rgba,
              uint8_t* img_normal, float max_depth);
  void DownloadAll(const std::string& file_path);
  void DownloadAllMesh(const std::string& vertices_path, const std::string& indices_path,
                       const std::string& prob_path);  // tsdf_module.h:119-120
  bool is_terminated();
  void terminate();
  void SetPause(bool pause);

  // additions (not in the reference)
  void Flush();                 // block until every queued frame has been integrated
  size_t QueueSize();
  int NumActiveBlock();
  size_t frames_integrated();
  size_t pool_system_allocs() { return pool_.system_allocs(); }   // host_alloc calls of the queue's block pool
  size_t pool_system_frees() { return pool_.system_frees(); }
  size_t pool_pageable_blocks() { return pool_.pageable_blocks(); }  // frames the queue kept in ordinary memory
  // page-locked memory the queue may hold (default HostBlockPool::kPinnedBytes); call before the first Integrate
  void SetPinnedBudget(size_t bytes) { pool_.set_pinned_budget(bytes); }
  // true (default): Integrate() waits for the worker once two batches + an arena of frames (72) are queued or in
  // flight; false: the reference's unbounded queue (tsdf_module.cc:99-100).  Call before the first Integrate.
  void SetQueueBounded(bool bounded) { bounded_ = bounded; }

 private:
  void Run();
  static constexpr size_t kMaxBatch = 32;  // frames per engine call (uploads run ahead inside the call)
  TSDFGrid tsdf_;
  HostBlockPool pool_;
  ParallelCopier copier_;
  std::mutex mtx_copy_;  // Integrate() may be called from several threads; the copier serves one at a time
  float max_depth_;
  const CameraIntrinsics<float> intrinsics_;
  const SE3<float> cam_T_posecam_;
  std::mutex mtx_queue_;
  std::queue<std::unique_ptr<TSDFSystemInput>> inputs_;
  std::condition_variable cv_queue_;  // the reference busy-polls (tsdf_module.cc:101)
  bool busy_ = false;
  bool bounded_ = true;
  std::mutex mtx_read_;
  std::mutex mtx_terminate_;
  bool terminate_ = false;
  std::mutex mtx_pause_;
  std::condition_variable cv_pause_;
  bool pause_ = false;
  size_t frames_done_ = 0;
  std::thread t_;  // declared last: everything above exists before Run() starts (tsdf_module.h:144-164)
};

}  // namespace ratsdf
