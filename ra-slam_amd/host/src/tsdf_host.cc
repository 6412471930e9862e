// tsdf_host.cc -- TSDFGrid / TSDFSystem host layer over the C ABI (see the headers for the mapping
// to the reference's modules/tsdf_module.* and utils/tsdf/voxel_tsdf.cuh).
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>

#include "ratsdf/tsdf_system.hpp"

namespace ratsdf {

namespace {
std::string default_library() {
  if (const char* env = getenv("RATSDF_LIB")) return env;
  Dl_info info;
  if (dladdr((void*)&default_library, &info) && info.dli_fname) {
    std::string here(info.dli_fname);
    const size_t slash = here.rfind('/');
    const std::string dir = slash == std::string::npos ? "." : here.substr(0, slash);
    return dir + "/../../csrc/build/libratsdf.so";  // ra-slam_amd/host/build -> ra-slam_amd/csrc/build
  }
  return "libratsdf.so";
}
}  // namespace

const Api& Api::Load(const char* path, const char* prefix) {
  static std::mutex mtx;
  static std::map<std::string, Api> loaded;
  std::lock_guard<std::mutex> lock(mtx);
  const std::string lib = path ? path : default_library();
  const std::string key = lib + "|" + prefix;
  auto it = loaded.find(key);
  if (it != loaded.end()) return it->second;
  Api api;
  api.handle = dlopen(lib.c_str(), RTLD_NOW | RTLD_LOCAL);
  if (!api.handle) {
    fprintf(stderr, "[ratsdf] cannot load %s: %s\n", lib.c_str(), dlerror());
    abort();  // no fallback: the HIP engine is the only implementation of the product path
  }
  auto sym = [&](const char* name) {
    const std::string full = std::string(prefix) + name;
    void* p = dlsym(api.handle, full.c_str());
    if (!p) {
      fprintf(stderr, "[ratsdf] %s does not export %s\n", lib.c_str(), full.c_str());
      abort();
    }
    return p;
  };
  api.create = reinterpret_cast<decltype(api.create)>(sym("create"));
  api.destroy = reinterpret_cast<decltype(api.destroy)>(sym("destroy"));
  api.integrate = reinterpret_cast<decltype(api.integrate)>(sym("integrate"));
  api.integrate_batch = reinterpret_cast<decltype(api.integrate_batch)>(sym("integrate_batch"));
  api.host_alloc = reinterpret_cast<decltype(api.host_alloc)>(sym("host_alloc"));
  api.host_free = reinterpret_cast<decltype(api.host_free)>(sym("host_free"));
  api.synchronize = reinterpret_cast<decltype(api.synchronize)>(sym("synchronize"));
  api.recover = reinterpret_cast<decltype(api.recover)>(sym("recover"));
  api.query = reinterpret_cast<decltype(api.query)>(sym("query"));
  api.gather_valid = reinterpret_cast<decltype(api.gather_valid)>(sym("gather_valid"));
  api.gather_valid_semantic =
      reinterpret_cast<decltype(api.gather_valid_semantic)>(sym("gather_valid_semantic"));
  api.download_all = reinterpret_cast<decltype(api.download_all)>(sym("download_all"));
  api.raycast = reinterpret_cast<decltype(api.raycast)>(sym("raycast"));
  api.gather_valid_mesh =
      reinterpret_cast<decltype(api.gather_valid_mesh)>(sym("gather_valid_mesh"));
  api.download_all_mesh =
      reinterpret_cast<decltype(api.download_all_mesh)>(sym("download_all_mesh"));
  api.free_buffer = reinterpret_cast<decltype(api.free_buffer)>(sym("free_buffer"));
  api.num_active_blocks = reinterpret_cast<decltype(api.num_active_blocks)>(sym("num_active_blocks"));
  api.status_string = reinterpret_cast<decltype(api.status_string)>(sym("status_string"));
  api.backend = reinterpret_cast<decltype(api.backend)>(sym("backend"));
  return loaded.emplace(key, api).first->second;
}

// ---- TSDFGrid -------------------------------------------------------------------------------
TSDFGrid::TSDFGrid(float voxel_size, float truncation, int device, const Api* api)
    : api_(api ? api : &Api::Load()) {
  note(api_->create(voxel_size, truncation, device, &engine_), "create");
}

TSDFGrid::~TSDFGrid() {
  if (engine_) api_->destroy(engine_);
}

void TSDFGrid::note(int st, const char* what) {
  status_ = st;
  if (st != RATSDF_OK)
    fprintf(stderr, "[ratsdf] %s failed: %s\n", what, api_->status_string(st));
}

void TSDFGrid::Integrate(const Image& rgb, const Image& depth, const Image& ht, const Image& lt,
                         float max_depth, const CameraIntrinsics<float>& K,
                         const SE3<float>& cam_T_world) {
  if (!engine_) return;
  // the reference's asserts, voxel_tsdf.cu:419-428
  if (rgb.type != kU8C3 || depth.type != kF32C1 || rgb.rows != depth.rows ||
      rgb.cols != depth.cols ||
      (!ht.empty() && (ht.type != kF32C1 || ht.rows != depth.rows || ht.cols != depth.cols)) ||
      (!lt.empty() && (lt.type != kF32C1 || lt.rows != depth.rows || lt.cols != depth.cols))) {
    note(RATSDF_ERR_BAD_ARGUMENT, "Integrate");
    return;
  }
  const ratsdf_intrinsics k{K.fx, K.fy, K.cx, K.cy};
  const ratsdf_pose p = cam_T_world.abi();
  note(api_->integrate(engine_, static_cast<const uint8_t*>(rgb.data),
                       static_cast<const float*>(depth.data),
                       ht.empty() ? nullptr : static_cast<const float*>(ht.data),
                       lt.empty() ? nullptr : static_cast<const float*>(lt.data), depth.rows,
                       depth.cols, max_depth, &k, &p),
       "Integrate");
}

void TSDFGrid::IntegrateBatch(int n, const uint8_t* const* rgb, const float* const* depth,
                              const float* const* ht, const float* const* lt, int rows, int cols,
                              float max_depth, const CameraIntrinsics<float>& K,
                              const SE3<float>* cam_T_world, bool pinned) {
  if (!engine_ || n <= 0) return;
  std::vector<ratsdf_intrinsics> ks((size_t)n, ratsdf_intrinsics{K.fx, K.fy, K.cx, K.cy});
  std::vector<ratsdf_pose> ps((size_t)n);
  for (int i = 0; i < n; ++i) ps[(size_t)i] = cam_T_world[i].abi();
  note(api_->integrate_batch(engine_, n, rgb, depth, ht, lt, rows, cols, max_depth, ks.data(), ps.data(),
                             pinned ? 1 : 0),
       "IntegrateBatch");
}

void TSDFGrid::RayCast(float max_depth, const CameraParams& cam, const SE3<float>& cam_T_world,
                       uint8_t* tsdf_rgba, uint8_t* tsdf_normal) {
  if (!engine_) return;
  const ratsdf_intrinsics k{cam.intrinsics.fx, cam.intrinsics.fy, cam.intrinsics.cx,
                            cam.intrinsics.cy};
  const ratsdf_pose p = cam_T_world.abi();
  note(api_->raycast(engine_, &k, cam.img_h, cam.img_w, &p, max_depth, tsdf_rgba, tsdf_normal),
       "RayCast");
}

std::vector<VoxelSpatialTSDF> TSDFGrid::GatherValid() {
  std::vector<VoxelSpatialTSDF> out;
  if (!engine_) return out;
  ratsdf_voxel_tsdf* buf = nullptr;
  size_t n = 0;
  note(api_->gather_valid(engine_, &buf, &n), "GatherValid");
  if (status_ == RATSDF_OK) out.assign(buf, buf + n);
  if (buf) api_->free_buffer(buf);
  return out;
}

std::vector<VoxelSpatialTSDFSEGM> TSDFGrid::GatherValidSemantic() {
  std::vector<VoxelSpatialTSDFSEGM> out;
  if (!engine_) return out;
  ratsdf_voxel_segm* buf = nullptr;
  size_t n = 0;
  note(api_->gather_valid_semantic(engine_, &buf, &n), "GatherValidSemantic");
  if (status_ == RATSDF_OK) out.assign(buf, buf + n);
  if (buf) api_->free_buffer(buf);
  return out;
}

std::vector<VoxelSpatialTSDF> TSDFGrid::GatherVoxels(const BoundingCube<float>& v) {
  std::vector<VoxelSpatialTSDF> out;
  if (!engine_) return out;
  const ratsdf_bounds b{v.xmin, v.xmax, v.ymin, v.ymax, v.zmin, v.zmax};
  ratsdf_voxel_tsdf* buf = nullptr;
  size_t n = 0;
  note(api_->query(engine_, &b, &buf, &n), "GatherVoxels");
  if (status_ == RATSDF_OK) out.assign(buf, buf + n);
  if (buf) api_->free_buffer(buf);
  return out;
}

void TSDFGrid::GatherValidMesh(std::vector<float>* vb, std::vector<int32_t>* ib,
                               std::vector<float>* pb) {
  if (!engine_ || !vb || !ib || !pb) return;
  float *v = nullptr, *p = nullptr;
  int32_t* idx = nullptr;
  size_t nv = 0, nt = 0;
  note(api_->gather_valid_mesh(engine_, &v, &nv, &idx, &nt, &p), "GatherValidMesh");
  if (status_ == RATSDF_OK) {
    vb->assign(v, v + nv * 3);
    pb->assign(p, p + nv);
    ib->assign(idx, idx + nt * 3);
  }
  if (v) api_->free_buffer(v);
  if (p) api_->free_buffer(p);
  if (idx) api_->free_buffer(idx);
}

void TSDFGrid::DownloadAllMesh(const std::string& vp, const std::string& ip, const std::string& pp) {
  if (engine_) note(api_->download_all_mesh(engine_, vp.c_str(), ip.c_str(), pp.c_str()), "DownloadAllMesh");
}

void TSDFGrid::DownloadAll(const std::string& path) {
  if (engine_) note(api_->download_all(engine_, path.c_str()), "DownloadAll");
}

void TSDFGrid::Synchronize() {
  if (engine_) note(api_->synchronize(engine_), "Synchronize");
}

bool TSDFGrid::Recover() {
  if (!engine_) return false;
  status_ = api_->recover(engine_);
  return status_ == RATSDF_OK;
}

int TSDFGrid::NumActiveBlock() {
  int32_t n = 0;
  if (engine_) note(api_->num_active_blocks(engine_, &n), "NumActiveBlock");
  return n;
}

// ---- page-locked block pool -----------------------------------------------------------------
HostBlockPool::Arena* HostBlockPool::arena_of(const void* p) {
  const uint8_t* q = static_cast<const uint8_t*>(p);
  for (Arena& a : arenas_)
    if (q >= a.base && q < a.base + a.block_bytes * a.blocks) return &a;
  return nullptr;
}

bool HostBlockPool::grow(size_t bytes) {
  for (size_t blocks : {kArenaBlocks, (size_t)1}) {  // under memory pressure: single blocks
    if (pinned_bytes_ + bytes * blocks > budget_) continue;
    void* base = nullptr;
    ++allocs_;
    if (api_->host_alloc(bytes * blocks, &base) != RATSDF_OK || !base) continue;
    Arena a;
    a.base = static_cast<uint8_t*>(base);
    a.block_bytes = bytes;
    a.blocks = blocks;
    arenas_.push_back(a);
    pinned_bytes_ += bytes * blocks;
    for (size_t k = blocks; k-- > 0;) {
      const HostBlock b{a.base + k * bytes, bytes, true};
      free_.insert(std::lower_bound(free_.begin(), free_.end(), b,
                                    [](const HostBlock& x, const HostBlock& y) { return x.ptr > y.ptr; }), b);
    }
    return true;
  }
  return false;
}

bool HostBlockPool::wait_for_release(int ms) {
  std::unique_lock<std::mutex> lock(mtx_);
  const size_t seen = releases_;
  return cv_release_.wait_for(lock, std::chrono::milliseconds(ms), [&] { return releases_ != seen; });
}

// grow: may page-lock more memory when no block is free (beyond the reserve, which the first block of a size takes
// in any case); pageable: may hand out ordinary memory when page-locked memory is exhausted.  Neither: nullptr.
HostBlock HostBlockPool::acquire(size_t bytes, bool grow_ok, bool pageable_ok) {
  std::lock_guard<std::mutex> lock(mtx_);
  if (reserved_for_ != bytes) {  // the first block of this size: everything a running system keeps in flight
    reserved_for_ = bytes;
    size_t have = 0;
    for (const Arena& a : arenas_)
      if (a.block_bytes == bytes) have += a.blocks;
    while (have < reserve_ && grow(bytes)) have += arenas_.back().blocks;
  }
  for (int attempt = 0; attempt < 2; ++attempt) {
    for (size_t i = free_.size(); i-- > 0;)  // lowest address first
      if (free_[i].bytes == bytes) {
        const HostBlock b = free_[i];
        free_.erase(free_.begin() + (long)i);
        if (Arena* a = arena_of(b.ptr)) ++a->in_use;
        return b;
      }
    // (a pool without any block of this size -- the reserve could not be page-locked -- grows whatever the caller says)
    bool have = false;
    for (const Arena& a : arenas_) have = have || a.block_bytes == bytes;
    if (attempt == 0 && ((!grow_ok && have) || !grow(bytes))) break;
  }
  if (!pageable_ok) {
    bool have = false;
    for (const Arena& a : arenas_) have = have || a.block_bytes == bytes;
    if (have) return HostBlock{nullptr, bytes, true};  // the caller waits for a release
  }
  // the page-locked budget is spent (or there is no such memory): ordinary memory, staged by the engine
  ++pageable_;
  return HostBlock{malloc(bytes), bytes, false};  // (nullptr: the caller drops the frame and says so)
}

void HostBlockPool::release(const HostBlock& b) {
  if (!b.ptr) return;
  if (!b.pinned) {
    free(b.ptr);
    return;
  }
  std::lock_guard<std::mutex> lock(mtx_);
  if (Arena* a = arena_of(b.ptr)) --a->in_use;
  free_.insert(std::lower_bound(free_.begin(), free_.end(), b,
                                [](const HostBlock& x, const HostBlock& y) { return x.ptr > y.ptr; }), b);
  ++releases_;
  cv_release_.notify_all();
}

// Whole idle arenas beyond the larger of kParkedBytes and the reserve go back to the system (page-locked memory
// is not reclaimable by anybody else).  Called when no frame is queued or in flight.
void HostBlockPool::trim() {
  std::lock_guard<std::mutex> lock(mtx_);
  const size_t keep = std::max(kParkedBytes, reserve_ * reserved_for_);
  for (size_t i = arenas_.size(); i-- > 0 && pinned_bytes_ > keep;) {
    const Arena a = arenas_[i];
    if (a.in_use) continue;
    free_.erase(std::remove_if(free_.begin(), free_.end(),
                               [&](const HostBlock& f) {
                                 return static_cast<uint8_t*>(f.ptr) >= a.base &&
                                        static_cast<uint8_t*>(f.ptr) < a.base + a.block_bytes * a.blocks;
                               }),
                free_.end());
    ++frees_;
    api_->host_free(a.base);
    pinned_bytes_ -= a.block_bytes * a.blocks;
    arenas_.erase(arenas_.begin() + (long)i);
  }
}

HostBlockPool::~HostBlockPool() {
  for (const Arena& a : arenas_) api_->host_free(a.base);
}

// ---- parallel clone ---------------------------------------------------------------------------
ParallelCopier::ParallelCopier(int helpers) {
  for (int i = 0; i < helpers; ++i) threads_.emplace_back(&ParallelCopier::worker, this);
}

ParallelCopier::~ParallelCopier() {
  {
    std::lock_guard<std::mutex> lock(mtx_);
    stop_ = true;
  }
  cv_work_.notify_all();
  for (auto& t : threads_) t.join();
}

void ParallelCopier::worker() {
  std::unique_lock<std::mutex> lock(mtx_);
  while (true) {
    while (!stop_ && next_ >= pieces_.size()) cv_work_.wait(lock);
    if (stop_) return;
    const Piece p = pieces_[next_++];
    lock.unlock();
    memcpy(p.dst, p.src, p.bytes);
    lock.lock();
    if (++done_ == pieces_.size()) cv_done_.notify_all();
  }
}

void ParallelCopier::copy(const void* const* src, void* const* dst, const size_t* bytes, int n) {
  // pieces of at most 1 MiB: enough of them for every helper, large enough to amortise the hand-off
  constexpr size_t kPiece = 1u << 20;
  std::unique_lock<std::mutex> lock(mtx_);
  pieces_.clear();
  next_ = done_ = 0;
  for (int i = 0; i < n; ++i)
    for (size_t off = 0; off < bytes[i]; off += kPiece)
      pieces_.push_back(Piece{static_cast<uint8_t*>(dst[i]) + off, static_cast<const uint8_t*>(src[i]) + off,
                              std::min(kPiece, bytes[i] - off)});
  if (pieces_.empty()) return;
  if (!threads_.empty()) cv_work_.notify_all();
  while (next_ < pieces_.size()) {  // the caller copies too
    const Piece p = pieces_[next_++];
    lock.unlock();
    memcpy(p.dst, p.src, p.bytes);
    lock.lock();
    ++done_;
  }
  while (done_ < pieces_.size()) cv_done_.wait(lock);
}

// ---- TSDFSystem -----------------------------------------------------------------------------
TSDFSystem::TSDFSystem(float voxel_size, float truncation, float max_depth,
                       const CameraIntrinsics<float>& intrinsics, const SE3<float>& extrinsics,
                       int device, const Api* api)
    : tsdf_(voxel_size, truncation, device, api),
      pool_(&tsdf_.api(), 2 * kMaxBatch + HostBlockPool::kArenaBlocks),
      copier_(getenv("RATSDF_COPY_THREADS") ? atoi(getenv("RATSDF_COPY_THREADS")) : 3),
      max_depth_(max_depth),
      intrinsics_(intrinsics),
      cam_T_posecam_(extrinsics),
      t_(&TSDFSystem::Run, this) {}

TSDFSystem::~TSDFSystem() {
  this->terminate();
  while (!inputs_.empty()) {  // frames dropped by terminate(): give their blocks back
    pool_.release(inputs_.front()->block);
    inputs_.pop();
  }
}

void TSDFSystem::Integrate(const SE3<float>& posecam_T_world, const Image& rgb, const Image& depth,
                           const Image& ht, const Image& lt) {
  std::unique_lock<std::mutex> pause_lock(mtx_pause_);
  while (pause_) cv_pause_.wait(pause_lock);
  auto in = std::make_unique<TSDFSystemInput>();
  in->cam_T_world = cam_T_posecam_ * posecam_T_world;  // tsdf_module.cc:28,33
  in->rows = depth.rows;
  in->cols = depth.cols;
  const size_t npix = (size_t)depth.rows * depth.cols;
  // clone(), tsdf_module.cc:28-35 -- into ONE page-locked block [depth | ht | lt | rgb] so that the
  // worker can upload it without another copy.  Missing ht / lt stay missing: the engine treats
  // them as the all-ones images the reference would build here (tsdf_module.cc:29-31).
  // The queue's copy lives in a page-locked block.  By default the queue is BOUNDED by the pool's reserve (two
  // batches + an arena of frames): when every block is in flight the producer waits for the worker -- a knowing
  // deviation from the reference, whose queue grows without bound and only warns (tsdf_module.cc:99-100); there a
  // producer that outruns the integration runs out of memory, here out of page-locked memory first.
  // SetQueueBounded(false) restores the reference's behaviour (page-locked up to the budget, then ordinary memory).
  in->block = pool_.acquire(npix * 16, /*grow=*/!bounded_, /*pageable=*/!bounded_);
  while (!in->block.ptr && bounded_) {
    {
      std::lock_guard<std::mutex> tl(mtx_terminate_);
      if (terminate_) return;  // nobody will ever free a block: the frame is dropped, like every frame queued now
    }
    pool_.wait_for_release(2);
    in->block = pool_.acquire(npix * 16, false, false);
  }
  if (!in->block.ptr) {  // no memory at all for the queue's copy: the frame is dropped, like a reference run
    fprintf(stderr, "[TSDF System] out of memory: frame dropped\n");  // whose cv::Mat::clone threw
    return;
  }
  in->has_sem = !(ht.empty() || lt.empty());
  uint8_t* b = static_cast<uint8_t*>(in->block.ptr);
  {
    const void* src[4] = {depth.data, rgb.data, ht.data, lt.data};
    void* dst[4] = {b, b + npix * 12, b + npix * 4, b + npix * 8};
    const size_t bytes[4] = {npix * 4, npix * 3, npix * 4, npix * 4};
    std::lock_guard<std::mutex> lock(mtx_copy_);
    copier_.copy(src, dst, bytes, in->has_sem ? 4 : 2);
  }
  {
    std::lock_guard<std::mutex> lock(mtx_queue_);
    inputs_.push(std::move(in));
  }
  cv_queue_.notify_all();
}

std::vector<VoxelSpatialTSDF> TSDFSystem::Query(const BoundingCube<float>& volumn) {
  std::lock_guard<std::mutex> lock(mtx_read_);
  return tsdf_.GatherVoxels(volumn);
}

void TSDFSystem::Render(const CameraParams& virtual_cam, const SE3<float> cam_T_world,
                        uint8_t* img_rgba, uint8_t* img_normal) {
  std::lock_guard<std::mutex> lock(mtx_read_);
  tsdf_.RayCast(max_depth_ * 2, virtual_cam, cam_T_world, img_rgba, img_normal);  // tsdf_module.cc:45-49
}

void TSDFSystem::Render(const CameraParams& virtual_cam, const SE3<float> cam_T_world,
                        uint8_t* img_rgba, uint8_t* img_normal, float max_depth) {
  std::lock_guard<std::mutex> lock(mtx_read_);
  tsdf_.RayCast(max_depth, virtual_cam, cam_T_world, img_rgba, img_normal);       // tsdf_module.cc:51-55
}

void TSDFSystem::DownloadAll(const std::string& file_path) {
  std::lock_guard<std::mutex> lock(mtx_read_);
  tsdf_.DownloadAll(file_path);
}

void TSDFSystem::DownloadAllMesh(const std::string& vertices_path, const std::string& indices_path,
                                 const std::string& prob_path) {
  std::lock_guard<std::mutex> lock(mtx_read_);
  tsdf_.DownloadAllMesh(vertices_path, indices_path, prob_path);
}

void TSDFSystem::Run() {
  while (true) {
    std::vector<std::unique_ptr<TSDFSystemInput>> batch;
    {
      std::unique_lock<std::mutex> lock(mtx_queue_);
      // wake up for new input or for termination; the reference spins here (tsdf_module.cc:88-102)
      while (inputs_.empty()) {
        {
          std::lock_guard<std::mutex> tl(mtx_terminate_);
          if (terminate_) return;
        }
        cv_queue_.wait_for(lock, std::chrono::milliseconds(2));
      }
      {
        std::lock_guard<std::mutex> tl(mtx_terminate_);
        if (terminate_) return;  // queued frames are dropped, as in the reference
      }
      if (inputs_.size() > 10)
        fprintf(stderr, "[TSDF System] Processing cannot catch up (input size: %zu)\n",
                inputs_.size());
      // everything that is queued right now (same image size), at most kMaxBatch frames: one engine
      // call, uploads overlapped with integration; the reference takes one frame per iteration
      // (... and the same kind of memory: frames the queue had to keep in ordinary memory go as pageable calls)
      while (!inputs_.empty() && batch.size() < kMaxBatch &&
             (batch.empty() || (inputs_.front()->rows == batch[0]->rows &&
                                inputs_.front()->cols == batch[0]->cols &&
                                inputs_.front()->block.pinned == batch[0]->block.pinned))) {
        batch.push_back(std::move(inputs_.front()));
        inputs_.pop();
      }
      busy_ = true;
    }
    {
      std::lock_guard<std::mutex> lock(mtx_read_);
      const size_t n = batch.size(), npix = (size_t)batch[0]->rows * batch[0]->cols;
      std::vector<const uint8_t*> rgb(n);
      std::vector<const float*> depth(n), ht(n), lt(n);
      std::vector<SE3<float>> poses(n);
      bool pinned = true;
      for (size_t i = 0; i < n; ++i) {
        pinned = pinned && batch[i]->block.pinned;
        const uint8_t* b = static_cast<const uint8_t*>(batch[i]->block.ptr);
        depth[i] = reinterpret_cast<const float*>(b);
        ht[i] = batch[i]->has_sem ? reinterpret_cast<const float*>(b + npix * 4) : nullptr;
        lt[i] = batch[i]->has_sem ? reinterpret_cast<const float*>(b + npix * 8) : nullptr;
        rgb[i] = b + npix * 12;
        poses[i] = batch[i]->cam_T_world;
      }
      tsdf_.IntegrateBatch((int)n, rgb.data(), depth.data(), ht.data(), lt.data(), batch[0]->rows,
                           batch[0]->cols, max_depth_, intrinsics_, poses.data(), pinned);
    }
    for (auto& in : batch) pool_.release(in->block);
    bool idle;
    {
      std::lock_guard<std::mutex> lock(mtx_queue_);
      busy_ = false;
      frames_done_ += batch.size();
      idle = inputs_.empty();
    }
    cv_queue_.notify_all();
    if (idle) pool_.trim();  // (only when nothing is queued: the steady state never returns memory)
  }
}

bool TSDFSystem::is_terminated() {
  std::lock_guard<std::mutex> lock(mtx_terminate_);
  return terminate_;
}

void TSDFSystem::terminate() {
  {
    std::lock_guard<std::mutex> lock(mtx_terminate_);
    terminate_ = true;
  }
  cv_queue_.notify_all();
  if (t_.joinable()) t_.join();  // idempotent, unlike tsdf_module.cc:119-125
}

void TSDFSystem::SetPause(bool pause) {
  std::unique_lock<std::mutex> pause_lock(mtx_pause_);
  pause_ = pause;
  cv_pause_.notify_all();
}

void TSDFSystem::Flush() {
  {
    std::unique_lock<std::mutex> lock(mtx_queue_);
    while (!inputs_.empty() || busy_) {
      {
        std::lock_guard<std::mutex> tl(mtx_terminate_);
        if (terminate_) return;
      }
      cv_queue_.wait_for(lock, std::chrono::milliseconds(2));
    }
  }
  // the worker's engine calls return when the images are on their way, not when the frames are integrated
  std::lock_guard<std::mutex> lock(mtx_read_);
  tsdf_.Synchronize();
}

size_t TSDFSystem::QueueSize() {
  std::lock_guard<std::mutex> lock(mtx_queue_);
  return inputs_.size();
}

int TSDFSystem::NumActiveBlock() {
  std::lock_guard<std::mutex> lock(mtx_read_);
  return tsdf_.NumActiveBlock();
}

size_t TSDFSystem::frames_integrated() {
  std::lock_guard<std::mutex> lock(mtx_queue_);
  return frames_done_;
}

}  // namespace ratsdf
