// ratsdf_engine.hip -- host side of the MI355X-native TSDF engine: device memory, the per-frame
// launch sequence on one HIP stream, and the C ABI of include/ratsdf.h.
//
// Replaces TSDFGrid (utils/tsdf/voxel_tsdf.cu:376-559,847-883), VoxelHashTable / VoxelMemPool host
// parts (voxel_hash.cu:25-44,225; voxel_mem.cu:13-35,63-67).  gfx950 only; there is no CPU path.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "kernels_frame.h"
#include "kernels_mesh.h"

using namespace ratsdf;

#define HIPCHK(expr)                                                                      \
  do {                                                                                    \
    hipError_t err__ = (expr);                                                            \
    if (err__ != hipSuccess) {                                                            \
      fprintf(stderr, "[ratsdf] HIP error %s at %s:%d: %s\n", hipGetErrorName(err__),     \
              __FILE__, __LINE__, #expr);                                                 \
      return RATSDF_ERR_DEVICE;                                                           \
    }                                                                                     \
  } while (0)

// Scoped "current device" of the calling thread: HIP's current device is per thread and defaults to
// 0, so every entry point that allocates or launches selects the engine's device first and restores
// the caller's on the way out.
struct DeviceGuard {
  int prev = -1;
  bool changed = false;
  bool selected = true;  // false: the engine's device could not be made current -- every entry point
                         // then returns RATSDF_ERR_DEVICE instead of working on the caller's device
                         // with another device's pointers
  explicit DeviceGuard(int device) {
    if (device < 0) return;  // (no engine: the entry point reports the bad argument itself)
    if (hipGetDevice(&prev) != hipSuccess) {
      selected = false;
      return;
    }
    if (prev != device) {
      changed = hipSetDevice(device) == hipSuccess;
      selected = changed;
    }
  }
  bool ok() const { return selected; }
  ~DeviceGuard() {
    if (changed) (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};

namespace {

__global__ void k_init_table(Entry* entries, uint32_t* claim, unsigned long long* occ,
                             uint32_t num_entry, uint32_t num_bucket) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < num_entry) entries[i] = Entry{0, 0, 0, 0, -1};  // init_hash_table_kernel, voxel_hash.cu:14-17
  if (i < num_bucket) claim[i] = kInf;
  if (i < (num_entry + 63) / 64) occ[i] = 0ull;
}
// block_threads() (device_types.h) reads the workgroup size from a fixed place in the implicit kernel arguments:
// checked once per engine against blockDim.x, with an odd size, so that a toolchain that lays them out
// differently fails ratsdf_create instead of computing garbage.
__global__ void k_check_block_threads(uint32_t* out) {
  if (threadIdx.x == 0) out[0] = block_threads() == blockDim.x ? blockDim.x : 0u;
}

// ratsdf_recover: what the directory says, into the structures derived from it.  One lane per hash entry: a live
// entry sets its occupancy bit, fills its slot of Table::active and marks its pool block as in use; an entry the
// chained-bucket resolver placed but whose commit never ran (pool index pending) is emptied -- its chain links stay,
// a dead node in a chain is walked over.
__global__ void k_recover_scan(Table tab, uint32_t* unused) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= tab.num_entry) return;
  uint32_t* pe = reinterpret_cast<uint32_t*>(tab.entries + e);
  const int32_t idx = (int32_t)pe[2];
  if (idx == kPlaceholderIdx || idx >= tab.num_block) {
    pe[2] = (uint32_t)-1;
    return;
  }
  if (idx < 0) return;
  atomicOr(&tab.occ[e >> 6], 1ull << (e & 63));
  reinterpret_cast<uint4*>(tab.active)[idx] = make_uint4(pe[0], pe[1] & 0xFFFFu, (uint32_t)idx, e);
  unused[idx] = 0u;
}
// ... and the free list: the pool blocks no entry names, in ascending order (the lowest positions of the heap hold
// the lowest indices, as after creation: AquireBlock pops from the top)
__global__ void k_recover_heap(const uint32_t* unused, const uint32_t* pos, int32_t* heap, int32_t n) {
  const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && unused[i]) heap[pos[i]] = i;
}
__global__ void k_fill_u32(uint32_t* p, uint32_t v, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

__global__ void k_init_heap(int32_t* heap, int32_t n) {   // heap_init_kernel, voxel_mem.cu:6-11
  const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) heap[i] = i;
}

constexpr uint32_t kSlowCap = kSlowSortCap;
constexpr int kDefaultVPL = 2;  // voxels per lane in k_integrate (RATSDF_VPL=2|4|8 overrides: tuning)
constexpr uint32_t kSlowDelCap = 1u << 16;
constexpr int kStageSlots = 16;  // frames of the host-image entry points in flight (uploads run ahead)
constexpr int kUploadRun = 4;    // frames that go up with one copy when they lie side by side in page-locked memory

}  // namespace

// Staging copies of the host-image entry points (caller's pageable images -> the engine's page-locked slot):
// a 640x480 frame is 4.6 MB, which one core copies at ~10 GB/s -- 2 200 frames/s before anything else
// happens.  The images of a frame are copied side by side by a few helper threads (started on first use,
// parked on a condition variable in between); the caller takes the last piece itself.
class HostCopyPool {
 public:
  struct Piece {
    void* dst;
    const void* src;
    size_t bytes;
  };
  // (thread creation can fail -- RLIMIT_NPROC, a cgroup's pids limit -- and nothing may be thrown through the C
  // ABI: the pool carries on with the helpers that did start; with none, copy() is a plain memcpy loop)
  explicit HostCopyPool(unsigned helpers) {
    for (unsigned i = 0; i < helpers; ++i) {
      try {
        threads_.emplace_back([this] { run(); });
      } catch (...) {
        break;
      }
    }
  }
  ~HostCopyPool() {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& t : threads_) t.join();
  }
  // copies every piece; returns when all are done (one caller at a time: the engine's entry points are
  // serialised per handle, SURVEY 8b "Threading")
  void copy(const Piece* pieces, int n) {
    if (n <= 0) return;
    {
      std::lock_guard<std::mutex> lk(m_);
      pieces_ = pieces;
      next_ = 0;
      count_ = n - 1;  // the caller keeps the last one
      pending_.store(n - 1, std::memory_order_relaxed);
    }
    if (n > 1) cv_.notify_all();
    memcpy(pieces[n - 1].dst, pieces[n - 1].src, pieces[n - 1].bytes);
    take_pieces();  // whatever the helpers have not picked up yet
    while (pending_.load(std::memory_order_acquire) != 0) std::this_thread::yield();
  }

 private:
  void take_pieces() {
    for (;;) {
      const Piece* p = nullptr;
      {
        std::lock_guard<std::mutex> lk(m_);
        if (next_ < count_) p = &pieces_[next_++];
      }
      if (!p) return;
      memcpy(p->dst, p->src, p->bytes);
      pending_.fetch_sub(1, std::memory_order_release);
    }
  }
  void run() {
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [this] { return stop_ || next_ < count_; });
        if (stop_) return;
      }
      take_pieces();
    }
  }
  std::vector<std::thread> threads_;
  std::mutex m_;
  std::condition_variable cv_;
  const Piece* pieces_ = nullptr;
  int next_ = 0, count_ = 0;
  std::atomic<int> pending_{0};
  bool stop_ = false;
};

struct ratsdf_engine {
  int device = 0;
  hipStream_t stream = nullptr;
  float vs = 0, trunc = 0;
  int block_bits = 0, bucket_bits = 0;
  int shard_rank = 0, shard_count = 1, shard_slab_bits = 2;
  int S = 3;
  int vpl = kDefaultVPL;
  int debug = 0;
  unsigned integrate_grid = 4096;
  bool grid_from_env = false;

  Table tab{};
  Pool pool{};
  Ctl* ctl = nullptr;
  ratsdf_frame_stats* d_stats = nullptr;
  uint8_t *d_render = nullptr, *h_render = nullptr;  // ray casting through the host entry points: output + page-locked copy
  size_t render_cap = 0;
  uint32_t* d_occ = nullptr;  // ray casting: hashed occupancy of the blocks (kernels_raycast.h), built per rendering
  uint32_t* h_err = nullptr;  // page-locked landing place of the sticky error word (sticky())
  EngineDev* d_eng = nullptr;  // device copy of the engine record (device_types.h)

  // image-sized scratch
  size_t pix_cap = 0, rank_cap = 0, cur_nranks = 0;
  float4* texA[2] = {nullptr, nullptr};  // packed per-pixel texels, double-buffered: the candidate
  uint32_t* texB[2] = {nullptr, nullptr};  // pass of frame f+1 writes while k_integrate(f) reads
  CandSet cand[2] = {};                  // candidate sets, used alternately (kernels_cand.h)
  uint32_t* cand_count = nullptr;        // both sets' list counters
  unsigned parity = 0;                   // which texel buffer / candidate set the NEXT frame uses
  bool cand_ready = false;               // that frame's candidate pass has already been enqueued
  bool cand_split_env = false;
// (compile-time defaults that tools/ build variants of for same-box sweeps)
#ifndef RATSDF_GRID_VGA
#define RATSDF_GRID_VGA 4096
#endif
#ifndef RATSDF_GRID_HD
#define RATSDF_GRID_HD 8192
#endif
#ifndef RATSDF_CAND_SPLIT_HD
#define RATSDF_CAND_SPLIT_HD 100
#endif
#ifndef RATSDF_CAND_SPLIT_DEFAULT
#define RATSDF_CAND_SPLIT_DEFAULT 10  // (round 5, with the cheaper visible-list role: 5 - 15 % measure the same, 0 and 20 % are 1.5 % slower)
#endif
  unsigned cand_split = RATSDF_CAND_SPLIT_DEFAULT;  // percent of the look-ahead pass placed in k_front,
  unsigned cand_split_b = 0;             // in k_alloc_rank; the rest rides in k_integrate
  bool fused_serial = true;              // the frame's serial role rides in k_integrate (no k_alloc_rank)
  // ... or, in ordinary frames, at the tail of k_front (front_tail_role, RATSDF_FRONT_TAIL=1).  Off by default:
  // measured (profiles/r04_front_tail.txt) it shortens k_integrate to the pure voxel update (16.8 -> 12.1 us
  // at 640x480 with the whole look-ahead pass in k_front) but lengthens k_front by more (8.6 -> 15.8 us): the
  // role is ~7 us of dependent round trips wherever it runs, and inside k_integrate it hides behind the update.
  bool front_tail = false;
  uint32_t front_prio = 0u;              // 2: k_front's directory workgroups run at raised wave priority (+1 %)
  bool sort_lists = false;               // diagnostic build, RATSDF_SORT_LISTS=1: work lists sorted by image tile on the host (experiment)
  bool inline_off = false;               // diagnostic build, RATSDF_INLINE_CAND=0: k_cand + k_front for frames without look-ahead
  int commit_rot_env = -1;               // RATSDF_COMMIT_ROT: first committing workgroup (measurements)
  // With the serial role in the launch: which update workgroups take the frame's commits.  A grid of
  // at most two rounds of resident workgroups (256 CUs x 8): the first ones, which wait for the role
  // after their first block (anything later is the tail).  More rounds: the second round.
  // (`launch_wgs`: update workgroups of the whole launch -- S slices of `grid` for a group, whose
  // later slices start on a full machine whatever their size.)
  uint32_t commit_rotation(unsigned grid, unsigned launch_wgs) const {
    if (commit_rot_env >= 0) return (uint32_t)commit_rot_env < grid ? (uint32_t)commit_rot_env : 0u;
    if (launch_wgs < 8192u) return 0u;
    return grid > 4096u ? 3072u : grid / 4u * 3u;  // profiles/r02_commit_rot_sweep.txt
  }
  uint32_t* serial_scratch = nullptr;    // its scratch for the general paths (kSerialLdsBytes)
  unsigned cand_parts_env = 0;           // RATSDF_CAND_PARTS: consumer workgroups per candidate list
  unsigned cand_wgs = 248;               // look-ahead workgroups per host kernel (about one per CU)
  // dynamic LDS of k_alloc_rank: the serial role needs kSerialLdsBytes; asking for more than half a
  // CU's LDS keeps the look-ahead workgroups of the launch off the serial workgroup's CU (sharing it
  // stretched the frame's critical path by a quarter)
  unsigned serial_lds = 100 * 1024;
  Request* req = nullptr;
  uint32_t req_cap = 0;
  uint32_t* abitmap = nullptr;   // rank bitmap, many-request path only (whole 32-word groups)
  uint32_t* asummary = nullptr;  // one bit per group
  uint32_t* req_k = nullptr;     // rank among the winners, per request
  uint32_t* win_ranks = nullptr; // raster ranks of the winners (few-winners path)
  uint32_t* aprefix = nullptr;       // per-word prefix of the rank bitmap (set groups only)
  uint32_t awords_cap = 0, asum_words = 0;

  SlowRequest* slow = nullptr;
  XLock* xlocks = nullptr;
  unsigned long long* sort_scratch = nullptr;  // resolver's sort keys beyond the LDS capacity

  // directory-sized scratch
  unsigned long long* masks = nullptr;  // selection / visibility mask, one bit per directory entry
  uint32_t* wg_count = nullptr;         // selected entries per kVisWG-word workgroup
  uint32_t nwg = 0;
  VisItem* vis = nullptr;
  // per frame parity (frame f appends while the end of frame f-1's carve pass still reads its own):
  DelItem* del_list[2] = {nullptr, nullptr};  // slot-0 deletes of a pass (pool release pending)
  uint32_t* upd_wg[2] = {nullptr, nullptr};   // voxels updated, per k_integrate workgroup (mod 1024)
  bool pending = false;            // the last pass still owes its carve_finalize (kernels_carve.h)
  uint32_t* dbitmap = nullptr;   // delete bitmap indexed by hash entry (self-cleaning)
  uint32_t* dsummary = nullptr;
  uint32_t* dprefix = nullptr;
  void* d_mc = nullptr;   // marching-cubes tables (device), built on first use
  uint32_t vis_cap = 0;   // total items of `vis`
  uint32_t seg_cap = 0;   // items per work list (vis holds kNumLists + 1 segments)
  uint32_t dwords = 0;
  SlowDelete* slowdel[2] = {nullptr, nullptr};

  // query-side download buffers (grow-only)
  void* dl_dev = nullptr;
  void* dl_host = nullptr;
  size_t dl_cap = 0;

  // staging for the host-image entry points: kStageSlots frames of 16 bytes/pixel each
  size_t stage_pix = 0;
  uint8_t* h_stage = nullptr;  // pinned
  uint8_t* d_stage = nullptr;
  hipEvent_t stage_ev[kStageSlots] = {};  // upload of the slot's last user has been executed
  hipEvent_t use_ev[kStageSlots + 1] = {};  // the frame that read the slot has been executed (+1: call fence)
  hipStream_t copy_stream = nullptr;   // uploads of ratsdf_integrate_batch: even frames
  hipStream_t copy_stream2 = nullptr;  // ... odd frames (two copy engines: one sustains ~31 GB/s)
  // Both host-image entry points use the slots as ONE ring (slot = stage_no % kStageSlots, counted over every
  // frame either of them has taken) and neither waits for its frames: what a slot's next user has to wait for is
  // in the slot's two events, whichever call recorded them -- no fence between calls.
  uint64_t stage_no = 0;
  // a frame with ht / lt has been integrated, or blocks were imported with their probabilities (FrameParams::segm_live)
  mutable bool ever_sem = false;
  bool sync_integrate = false;         // RATSDF_SYNC_INTEGRATE=1: wait for every frame (the round-3 behaviour)
  HostCopyPool* copy_pool = nullptr;

  // HIP graphs of the batch entry point (ratsdf_integrate_device_batch): the launches of an n-frame batch are
  // captured once per (image size, n) -- the kernels of the group path with one member, which take every
  // per-frame operand (pose, intrinsics, image pointers, counter-set parity) from a FrameJob table in device
  // memory -- and replayed with a fresh table.  Host cost per frame: a table row instead of two launches.
  struct BatchGraph {
    int H = 0, W = 0, n = 0;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    FrameJob* d_jobs = nullptr;
    FrameJob* h_jobs[2] = {nullptr, nullptr};  // page-locked, used alternately
    hipEvent_t ev[2] = {nullptr, nullptr};     // the copy out of h_jobs[i] has been executed
    unsigned turn = 0;
    uint64_t last_use = 0;
  };
  std::vector<BatchGraph> graphs;
  bool use_graphs = true;                // RATSDF_GRAPH=0: every frame launched by itself
  uint64_t graph_clock = 0;
  void free_graph(BatchGraph& g);
  int batch_graph(int n, int H, int W, BatchGraph** out);
  struct GraphShape {
    int H, W, n;
  };
  std::vector<GraphShape> graph_failed;  // shapes whose capture failed once: launched frame by frame from then on

  // profiling of the dominant kernel
  bool profiling = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
  size_t prof_used = 0;
  uint64_t prof_frame = 0;
  uint64_t prof_batch = 0;
  double prof_ms = 0;
  int64_t prof_n = 0;
  int prof_mode = 1;                     // 1: every 4th frame, sums only; 2: every frame, per-frame records
  std::vector<float> prof_k_us, prof_period_us;  // mode 2: kernel time of a frame / start-to-start period

  int free_all();
  int ensure_image(size_t npix, size_t nranks);
  int ensure_stage(size_t npix);
  EngineDev record() const;
  int upload_record();
  RankBufs rank_bufs(uint32_t nranks) const;
  int alloc_rank(uint32_t nranks, unsigned par, const CandJob* next = nullptr, bool frame = false);
  int settle();
  void abandon_pipeline();
  CarveBufs carve_bufs(unsigned par) const;
  int select(int mode, const GridBounds& gb, uint32_t* count_slot);
  struct FrameIn {
    const void *rgb, *depth, *ht, *lt;
    const ratsdf_intrinsics* K;
    const ratsdf_pose* T;
  };
  FrameParams frame_params(const FrameIn& in, int H, int W, float md) const;
  CandJob cand_job(const FrameIn& in, const FrameParams& P, unsigned par) const;
  int frame(const FrameIn& cur, const FrameIn* next, int H, int W, float md);
  int sticky();
  int read_small(void* dst, const void* dev_src, size_t bytes);
  int drain_profile(bool final = true);
  FrameParams base_params() const;
  struct Geom {
    unsigned n_vis_wg, parts, n_front_wg, n_cand_wg, grid;
    AheadGeom a, b, c;  // look-ahead shares of k_front, k_alloc_rank, k_integrate
  };
  Geom geometry(int H, int W, bool has_next, int split_a, int split_b) const;
};

FrameParams ratsdf_engine::base_params() const {
  FrameParams P;
  memset(&P, 0, sizeof(P));
  P.T = Se3{Quat{0, 0, 0, 1}, V3{0, 0, 0}};
  P.Ti = P.T;
  P.K = Intr{1, 1, 0, 0};
  P.Ki = P.K;
  P.vs = vs;
  P.trunc = trunc;
  P.md = 0;
  P.W = P.H = 0;
  P.S = 1;
  P.has_sem = 0;
  P.segm_live = 1;
  P.shard_rank = shard_rank;
  P.shard_count = shard_count;
  P.shard_slab_bits = shard_slab_bits;
  P.shard_bias = (32768 + shard_count - 1) / shard_count * shard_count;
  P.shard_magic = shard_count > 1 ? 0xFFFFFFFFu / (uint32_t)shard_count + 1u : 0u;
  P.debug = debug;
  return P;
}

int ratsdf_engine::free_all() {
  if (stream) (void)hipStreamSynchronize(stream);
  void* ptrs[] = {tab.active, tab.del_log, tab.del_count, tab.entries, tab.claim, tab.occ, pool.rgbw, pool.tsdf, pool.segm, pool.heap, ctl,
                  d_stats, d_eng, texA[0], texA[1], texB[0], texB[1], cand[0].list, cand[1].list, cand_count,
                  req, req_k, win_ranks, abitmap, asummary, aprefix,
                  slow, xlocks,
                  sort_scratch, masks, wg_count, vis, del_list[0], del_list[1], upd_wg[0],
                  upd_wg[1], tab.dclaim, dbitmap, dsummary, dprefix, slowdel[0], slowdel[1], d_stage, d_mc, serial_scratch, d_occ};
  if (copy_stream) (void)hipStreamSynchronize(copy_stream);
  if (copy_stream2) (void)hipStreamSynchronize(copy_stream2);
  if (stream) (void)hipStreamSynchronize(stream);
  delete copy_pool;
  copy_pool = nullptr;
  for (auto& g : graphs) free_graph(g);
  graphs.clear();
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (h_stage) (void)hipHostFree(h_stage);
  if (h_err) (void)hipHostFree(h_err);
  h_err = nullptr;
  if (d_render) (void)hipFree(d_render);
  if (h_render) (void)hipHostFree(h_render);
  d_render = h_render = nullptr;
  render_cap = 0;
  if (dl_dev) (void)hipFree(dl_dev);
  if (dl_host) (void)hipHostFree(dl_host);
  for (auto& ev : stage_ev)
    if (ev) (void)hipEventDestroy(ev);
  for (auto& ev : use_ev)
    if (ev) (void)hipEventDestroy(ev);
  if (copy_stream) (void)hipStreamDestroy(copy_stream);
  if (copy_stream2) (void)hipStreamDestroy(copy_stream2);
  for (auto& ev : prof_events) {
    (void)hipEventDestroy(ev.first);
    (void)hipEventDestroy(ev.second);
  }
  if (stream) (void)hipStreamDestroy(stream);
  return RATSDF_OK;
}

RankBufs ratsdf_engine::rank_bufs(uint32_t nranks) const {
  RankBufs rb;
  rb.req = req;
  rb.req_cap = req_cap;
  rb.req_k = req_k;
  rb.win_ranks = win_ranks;
  rb.slow = slow;
  rb.slow_cap = kSlowCap;
  rb.xlocks = xlocks;
  rb.bitmap = abitmap;
  rb.summary = asummary;
  rb.prefix = aprefix;
  rb.nwords = (nranks + 31) / 32;
  rb.sort_scratch = sort_scratch;
  return rb;
}

EngineDev ratsdf_engine::record() const {
  EngineDev r;
  memset(&r, 0, sizeof(r));
  r.tab = tab;
  r.pool = pool;
  r.cb[0] = carve_bufs(0);
  r.cb[1] = carve_bufs(1);
  r.rb = rank_bufs((uint32_t)cur_nranks);
  r.ctl = ctl;
  r.stats = d_stats;
  r.slow = slow;
  r.serial_scratch = serial_scratch;
  r.slow_cap = kSlowCap;
  r.seg_cap = seg_cap;
  r.vis = vis + kFreshCap;  // (the frame kernels' view: device_types.h, kFreshCap)
  for (int i = 0; i < 2; ++i) {
    r.texA[i] = texA[i];
    r.texB[i] = texB[i];
    r.cand[i] = cand[i];
  }
  return r;
}

// the device copy follows every (re)allocation; in stream order, so launches already enqueued keep
// reading the record they were enqueued with
int ratsdf_engine::upload_record() {
  const EngineDev r = record();
  HIPCHK(hipMemcpyAsync(d_eng, &r, sizeof(r), hipMemcpyHostToDevice, stream));
  HIPCHK(hipStreamSynchronize(stream));  // `r` is a stack object
  return RATSDF_OK;
}

int ratsdf_engine::ensure_image(size_t npix, size_t nranks) {
  if (npix <= pix_cap && nranks == cur_nranks) return RATSDF_OK;
  HIPCHK(hipStreamSynchronize(stream));
  if (npix > pix_cap) {
    for (int i = 0; i < 2; ++i) {
      if (texA[i]) (void)hipFree(texA[i]);
      if (texB[i]) (void)hipFree(texB[i]);
      texA[i] = nullptr;
      texB[i] = nullptr;
      HIPCHK(hipMalloc(&texA[i], npix * sizeof(float4)));
      HIPCHK(hipMalloc(&texB[i], npix * sizeof(uint32_t)));
    }
    pix_cap = npix;
  }
  if (nranks > rank_cap) {
    // candidate lists: every sample of the image could in principle ask for a different block
    const uint32_t seg = (uint32_t)((nranks + kCandSegs - 1) / kCandSegs) + 1024;
    for (int i = 0; i < 2; ++i) {
      if (cand[i].list) (void)hipFree(cand[i].list);
      cand[i].list = nullptr;
      HIPCHK(hipMalloc(&cand[i].list, (size_t)seg * kCandSegs * sizeof(uint4)));
      cand[i].seg_cap = seg;
    }
    if (req) (void)hipFree(req);
    if (req_k) (void)hipFree(req_k);
    if (abitmap) (void)hipFree(abitmap);
    if (asummary) (void)hipFree(asummary);
    if (aprefix) (void)hipFree(aprefix);
    req_cap = (uint32_t)nranks;
    awords_cap = (uint32_t)((nranks + 31) / 32);
    awords_cap = (awords_cap + kGroupWords - 1) / kGroupWords * kGroupWords;
    asum_words = (awords_cap / kGroupWords + 31) / 32;
    HIPCHK(hipMalloc(&req, (size_t)req_cap * sizeof(Request)));
    HIPCHK(hipMalloc(&req_k, (size_t)req_cap * 4));
    HIPCHK(hipMalloc(&abitmap, (size_t)awords_cap * 4));
    HIPCHK(hipMalloc(&asummary, (size_t)asum_words * 4));
    HIPCHK(hipMalloc(&aprefix, (size_t)awords_cap * 4));
    rank_cap = nranks;
  }
  // the rank bitmap cleans itself after every use; start from a clean one when (re)allocated
  if (nranks != cur_nranks) {
    HIPCHK(hipMemsetAsync(abitmap, 0, (size_t)awords_cap * 4, stream));
    HIPCHK(hipMemsetAsync(asummary, 0, (size_t)asum_words * 4, stream));
    cur_nranks = nranks;
  }
  return upload_record();
}

int ratsdf_engine::ensure_stage(size_t npix) {
  if (npix <= stage_pix) return RATSDF_OK;
  if (copy_stream) HIPCHK(hipStreamSynchronize(copy_stream));
  if (copy_stream2) HIPCHK(hipStreamSynchronize(copy_stream2));
  HIPCHK(hipStreamSynchronize(stream));
  if (h_stage) (void)hipHostFree(h_stage);
  if (d_stage) (void)hipFree(d_stage);
  h_stage = nullptr;
  d_stage = nullptr;
  const size_t bytes = npix * 16 * kStageSlots;  // per slot: depth 4 + ht 4 + lt 4 + rgb 3 (padded to 4)
  HIPCHK(hipHostMalloc(&h_stage, bytes, hipHostMallocDefault));
  HIPCHK(hipMalloc(&d_stage, bytes));
  for (int i = 0; i < kStageSlots; ++i)
    if (!stage_ev[i]) HIPCHK(hipEventCreateWithFlags(&stage_ev[i], hipEventDisableTiming));
  for (int i = 0; i <= kStageSlots; ++i)
    if (!use_ev[i]) HIPCHK(hipEventCreateWithFlags(&use_ev[i], hipEventDisableTiming));
  if (!copy_stream) HIPCHK(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
  if (!copy_stream2) HIPCHK(hipStreamCreateWithFlags(&copy_stream2, hipStreamNonBlocking));
  stage_pix = npix;
  return RATSDF_OK;
}

// rank kernel (resolve + mark + scan) on a rank space of `nranks`; the commit itself happens inside
// k_integrate for frames and in k_commit_only for the stand-alone test hook
CarveBufs ratsdf_engine::carve_bufs(unsigned par) const {
  CarveBufs cb;
  cb.del = del_list[par & 1u];
  cb.del_cap = (uint32_t)tab.num_block;
  cb.slow = slowdel[par & 1u];
  cb.slow_cap = kSlowDelCap;
  cb.upd_wg = upd_wg[par & 1u];
  cb.bitmap = dbitmap;
  cb.summary = dsummary;
  cb.prefix = dprefix;
  return cb;
}

int ratsdf_engine::alloc_rank(uint32_t nranks, unsigned par, const CandJob* next, bool frame) {
  CandJob none;
  memset(&none, 0, sizeof(none));
  const CandJob& job = next ? *next : none;
  const unsigned extra = job.n_tiles ? (job.n_tiles + job.tiles_per_wg - 1) / job.tiles_per_wg : 0;
  const RankBufs rb = rank_bufs(nranks);
  hipLaunchKernelGGL(k_alloc_rank, dim3(1 + extra), dim3(1024),
                     serial_lds, stream, tab, pool, rb, carve_bufs(par ^ 1u), ctl,
                     (uint32_t)par, d_stats, frame ? cand[par].count : (uint32_t*)nullptr, job);
  HIPCHK(hipGetLastError());
  return RATSDF_OK;
}

// After a failed launch or copy in the middle of a batch: nothing may stay queued that reads the
// caller's buffers, and the look-ahead state must not leak into the next call (a frame that skipped
// k_cand because "its candidate pass already ran" would consume the stale lists of a frame that
// never came).
void ratsdf_engine::abandon_pipeline() {
  if (stream) (void)hipStreamSynchronize(stream);
  if (cand_ready && cand[parity].count)  // lists filled for a frame that will not be integrated
    (void)hipMemsetAsync(cand[parity].count, 0, (size_t)kCandSegs * kCandCountStride * 4, stream);
  cand_ready = false;
  (void)settle();
  if (stream) (void)hipStreamSynchronize(stream);
}

// Everything but a following frame needs the last frame's carve pass completed first.
int ratsdf_engine::settle() {
  if (!pending) return RATSDF_OK;
  hipLaunchKernelGGL(k_settle, dim3(1), dim3(1024), 0, stream, tab, pool, carve_bufs(parity ^ 1u), ctl,
                     (uint32_t)(parity ^ 1u), d_stats);
  HIPCHK(hipGetLastError());
  pending = false;
  return RATSDF_OK;
}

// ordered compaction of the directory into `vis`; the count lands in *count_slot (device)
int ratsdf_engine::select(int mode, const GridBounds& gb, uint32_t* count_slot) {
  FrameParams P = base_params();
  if (mode == kSelValid)
    hipLaunchKernelGGL(k_select_flags<kSelValid>, dim3(nwg), dim3(kVisWG), 0, stream, tab, P, gb,
                       masks, wg_count);
  else if (mode == kSelOwned)
    hipLaunchKernelGGL(k_select_flags<kSelOwned>, dim3(nwg), dim3(kVisWG), 0, stream, tab, P, gb,
                       masks, wg_count);
  else
    hipLaunchKernelGGL(k_select_flags<kSelBounds>, dim3(nwg), dim3(kVisWG), 0, stream, tab, P, gb,
                       masks, wg_count);
  hipLaunchKernelGGL(k_select_scatter, dim3(nwg), dim3(kVisWG), 0, stream, tab, masks, wg_count, vis,
                     vis_cap, count_slot);
  HIPCHK(hipGetLastError());
  return RATSDF_OK;
}

FrameParams ratsdf_engine::frame_params(const FrameIn& in, int H, int W, float md) const {
  FrameParams P = base_params();
  const ratsdf_pose* T = in.T;
  const ratsdf_intrinsics* K = in.K;
  P.T = Se3{Quat{T->qx, T->qy, T->qz, T->qw}, V3{T->tx, T->ty, T->tz}};
  P.Ti = se3_inverse(P.T);                       // voxel_tsdf.cu:459
  P.K = Intr{K->fx, K->fy, K->cx, K->cy};
  P.Ki = intr_inverse(P.K);                      // camera.cuh:67
  P.md = md;
  P.W = W;
  P.H = H;
  P.S = S;
  P.has_sem = (in.ht && in.lt) ? 1 : 0;
  if (P.has_sem) ever_sem = true;  // (frames are prepared in the order they are integrated)
  P.segm_live = ever_sem ? 1 : 0;
  return P;
}

CandJob ratsdf_engine::cand_job(const FrameIn& in, const FrameParams& P, unsigned par) const {
  CandJob j;
  memset(&j, 0, sizeof(j));
  j.P = P;
  j.depth = (const float*)in.depth;
  j.rgb = (const uint8_t*)in.rgb;
  j.ht = (const float*)in.ht;
  j.lt = (const float*)in.lt;
  j.texA = texA[par];
  j.texB = texB[par];
  j.set = cand[par];
  j.tiles_x = (uint32_t)((P.W + 15) / 16);
  j.tiles_x_magic = cand_tiles_magic(j.tiles_x);
  j.first_tile = 0;
  j.n_tiles = j.tiles_x * (uint32_t)((P.H + 15) / 16) * 4u;
  j.tiles_per_wg = 4;
  return j;
}

// Launch geometry of a frame: workgroup counts and the split of the NEXT frame's candidate pass over
// this frame's three launches (percent in k_front and in k_alloc_rank; the rest rides in k_integrate).
ratsdf_engine::Geom ratsdf_engine::geometry(int H, int W, bool has_next, int split_a,
                                            int split_b) const {
  Geom g;
  memset(&g, 0, sizeof(g));
  const size_t npix = (size_t)H * W;
  const uint32_t tiles_x = (uint32_t)((W + 15) / 16);
  const uint32_t tiles = tiles_x * (uint32_t)((H + 15) / 16) * 4u;
  g.n_cand_wg = (tiles + 3) / 4;
  g.a.tiles_x = g.b.tiles_x = g.c.tiles_x = tiles_x;
  g.a.tiles_x_magic = g.b.tiles_x_magic = g.c.tiles_x_magic = cand_tiles_magic(tiles_x);
  g.a.tiles_per_wg = g.b.tiles_per_wg = g.c.tiles_per_wg = 4;
  if (has_next) {
    // k_front and k_integrate take whole 16x16 super-tiles (4 tiles per 256-thread workgroup)
    // (shares begin at multiples of 16 super-tiles: cand_pixel_work pairs neighbouring super-tiles per XCD)
    const uint32_t tiles_a = (uint32_t)((uint64_t)(tiles / 4) * (unsigned)split_a / 100) / 16 * 64;
    uint32_t tiles_b = (uint32_t)((uint64_t)(tiles / 4) * (unsigned)split_b / 100) / 16 * 64;
    if (split_a + split_b >= 100 || tiles_a + tiles_b > tiles) tiles_b = tiles - tiles_a;
    g.a.n_tiles = tiles_a;
    g.b.first_tile = tiles_a;
    g.b.n_tiles = tiles_b;
    // k_alloc_rank runs 1024-thread workgroups; a look-ahead workgroup uses as many of its 16 waves
    // as it needs to cover its tiles
    const uint32_t tpw = (tiles_b + cand_wgs - 1) / cand_wgs;
    g.b.tiles_per_wg = std::min<uint32_t>(std::max<uint32_t>(tpw, 1u), 16u);
    g.c.first_tile = tiles_a + tiles_b;
    g.c.n_tiles = tiles - tiles_a - tiles_b;
  }
  // visible-list workgroups: one lane per pool slot, from the top of the pool down (kernels_visible.h); 128 of them
  // cover a map of 65 536 blocks in one round of two loads per lane, larger maps take more rounds
  g.n_vis_wg = std::min<unsigned>(128u, std::max<unsigned>(1u, ((unsigned)tab.num_block + 2 * 256 - 1) / (2 * 256)));
  // consumer workgroups per candidate list: every 16x16 super-tile reserves kCandReserve entries
  // (one pass of 256 lanes per consumer when nothing overflows); finer voxels need more
  const unsigned supers = ((unsigned)W + 15) / 16 * (((unsigned)H + 15) / 16);
  unsigned parts = (supers * kCandReserve / kCandSegs + 255) / 256;
  if (vs < 0.004f) parts *= 2;
  parts = std::min(std::max(parts, 1u), 32u);
  if (cand_parts_env) parts = cand_parts_env;
  g.parts = parts;
  g.n_front_wg = g.n_vis_wg + kCandSegs * parts + kReleaseWGs + (g.a.n_tiles + 3) / 4;
  // more workgroups for images with several times more visible blocks than 640x480 (1280x720 / 2 mm: 13 k - 22 k
  // visible blocks).  Round 2 measured 16 384 as best (profiles/r02_grid_sweep.txt); with round 4's kernels 8 192
  // is: 60.3 vs 62.4 us per frame on the 20-frame ping-pong (6 144: 61.9, 12 288: 61.5), 11 316 vs 10 838
  // frames/s on the 416 MB map -- a workgroup takes 1.5 - 2.6 blocks, fewer workgroups to dispatch
  g.grid = grid_from_env ? integrate_grid : (npix >= 600000 ? (unsigned)RATSDF_GRID_HD : (unsigned)RATSDF_GRID_VGA);
  return g;
}

// One frame.  `next` (same image size) is the frame the caller will integrate right after this one,
// if it already knows it: its candidate pass then rides in this frame's single-workgroup kernels.
int ratsdf_engine::frame(const FrameIn& cur, const FrameIn* next, int H, int W, float md) {
  const size_t npix = (size_t)H * W;
  if (npix * (size_t)S >= 0xFFFFFFFFull) return RATSDF_ERR_BAD_ARGUMENT;
  int st = ensure_image(npix, npix * (size_t)S);
  if (st != RATSDF_OK) return st;
  const FrameParams P = frame_params(cur, H, W, md);
  const unsigned par = parity;

  // nobody looked ahead (a single frame, the first of a batch): this frame's candidate pass rides in k_front, every
  // pixel workgroup its own consumer (k_front_inline)
  const bool inline_cand = !cand_ready;
#ifdef RATSDF_STAMPS
  const bool inline_kernel = inline_cand && !tab.tail_on && !inline_off;  // (RATSDF_INLINE_CAND=0: k_cand + k_front, for A/B)
  if (inline_cand && !inline_kernel) {
    const CandJob job = cand_job(cur, P, par);
    hipLaunchKernelGGL(k_cand, dim3((job.n_tiles + 3) / 4), dim3(256), 0, stream, job, ctl);
  }
#else
  const bool inline_kernel = inline_cand;
#endif
  // Where the NEXT frame's candidate pass rides (interleaved A/B, profiles/r02_split_ab.txt): at
  // 640x480 10 % in k_front (20 % until round 5) and the rest at the head of k_integrate's grid (10-30 % measured the same,
  // 0 and 40 % are ~2.5 % slower: k_front is a chain of dependent round trips that a few riders do not
  // lengthen, the voxel update hides the rest); at 1280x720 all of it in k_front (best by 1-3 %, and
  // k_integrate stays the pure voxel update its roofline figure is about)
  // (k_integrate<1> runs 512-thread workgroups and hosts no look-ahead: everything in k_front then)
  const int split = vpl == 1 ? 100 : (int)(cand_split_env ? cand_split : (npix >= 600000 ? (unsigned)RATSDF_CAND_SPLIT_HD : cand_split));
  const int split_b = (fused_serial && vpl != 1) ? 0 : (int)(cand_split_env ? cand_split_b : 0u);
  const Geom g = geometry(H, W, next != nullptr, split, split_b);
  // shares of the next frame's candidate pass: k_front, k_alloc_rank, k_integrate
  CandJob ahead_a, ahead_b, ahead_c;
  memset(&ahead_a, 0, sizeof(ahead_a));
  memset(&ahead_b, 0, sizeof(ahead_b));
  memset(&ahead_c, 0, sizeof(ahead_c));
  if (next) {
    const FrameParams Pn = frame_params(*next, H, W, md);
    ahead_a = cand_job(*next, Pn, par ^ 1u);
    ahead_b = ahead_a;
    ahead_c = ahead_a;
    auto set = [](CandJob& j, const AheadGeom& ag) {
      j.first_tile = ag.first_tile;
      j.n_tiles = ag.n_tiles;
      j.tiles_per_wg = ag.tiles_per_wg;
    };
    set(ahead_a, g.a);
    set(ahead_b, g.b);
    set(ahead_c, g.c);
  }
  parity = par ^ 1u;
  cand_ready = next != nullptr;

  // fr[par] was zeroed when the frame before last was finalised (or at creation)
  const bool fused = fused_serial && vpl != 1;
  if (inline_kernel) {
    const CandJob now = cand_job(cur, P, par);
    const unsigned n_now_wg = (now.n_tiles + 3) / 4;
    hipLaunchKernelGGL(k_front_inline, dim3(g.n_vis_wg + n_now_wg + kReleaseWGs + (g.a.n_tiles + 3) / 4), dim3(256), 0,
                       stream, tab, (uint32_t)g.n_vis_wg, (uint32_t)n_now_wg, req, req_cap, slow, kSlowCap,
                       vis + kFreshCap, seg_cap, pool, carve_bufs(par ^ 1u), ctl, (uint32_t)par, now, ahead_a);
  } else
#ifdef RATSDF_STAMPS
  if (tab.tail_on)
    hipLaunchKernelGGL(k_front<true>, dim3(g.n_front_wg), dim3(256), 0, stream, tab, P, g.n_vis_wg, cand[par],
                       (uint32_t)g.parts, req, req_cap, slow, kSlowCap, vis + kFreshCap, seg_cap, pool,
                       carve_bufs(par ^ 1u), ctl, (uint32_t)par, d_stats, 1u | front_prio, ahead_a);
  else
#endif
    hipLaunchKernelGGL(k_front<false>, dim3(g.n_front_wg), dim3(256), 0, stream, tab, P, g.n_vis_wg, cand[par],
                       (uint32_t)g.parts, req, req_cap, slow, kSlowCap, vis + kFreshCap, seg_cap, pool,
                       carve_bufs(par ^ 1u), ctl, (uint32_t)par, d_stats, 0u, ahead_a);
#ifdef RATSDF_STAMPS
  // (experiment, RATSDF_SORT_LISTS=1: every work list put into image-tile order on the HOST between the two launches --
  // an upper bound for what tile-ordered lists would buy the update; the engine waits for k_front here, so only the
  // update's own time means anything in such a run)
  if (sort_lists) {
    HIPCHK(hipStreamSynchronize(stream));
    std::vector<uint32_t> fc(sizeof(FrameCtl) / 4);
    HIPCHK(hipMemcpy(fc.data(), &ctl->fr[par], sizeof(FrameCtl), hipMemcpyDeviceToHost));
    const FrameCtl* hf = reinterpret_cast<const FrameCtl*>(fc.data());
    const double qx = P.T.q.x, qy = P.T.q.y, qz = P.T.q.z, qw = P.T.q.w;
    for (int l = 0; l < kNumLists; ++l) {
      uint32_t n = hf->n_list[l * kListStride];
      if (n > seg_cap - kFreshCap) n = seg_cap - kFreshCap;
      if (n < 2) continue;
      std::vector<VisItem> items(n);
      VisItem* dl = vis + kFreshCap + (size_t)l * seg_cap;
      HIPCHK(hipMemcpy(items.data(), dl, (size_t)n * sizeof(VisItem), hipMemcpyDeviceToHost));
      auto key = [&](const VisItem& it) {
        const double x = (it.x * 8 + 4) * (double)P.vs, y = (it.y * 8 + 4) * (double)P.vs, z = (it.z * 8 + 4) * (double)P.vs;
        // rotate by the quaternion, translate, project
        const double ux = 2 * (qy * z - qz * y), uy = 2 * (qz * x - qx * z), uz = 2 * (qx * y - qy * x);
        const double cx = x + qw * ux + (qy * uz - qz * uy) + P.T.t.x, cy = y + qw * uy + (qz * ux - qx * uz) + P.T.t.y,
                     cz = z + qw * uz + (qx * uy - qy * ux) + P.T.t.z;
        double u = (P.K.fx * cx + P.K.cx * cz) / cz, v = (P.K.fy * cy + P.K.cy * cz) / cz;
        u = std::min(std::max(u, 0.0), (double)(P.W - 1));
        v = std::min(std::max(v, 0.0), (double)(P.H - 1));
        // finer than the 8x8 tiles: 32x32 cells in raster order of cells
        return (int)(v * 32.0 / P.H) * 32 + (int)(u * 32.0 / P.W);
      };
      std::stable_sort(items.begin(), items.end(), [&](const VisItem& a, const VisItem& b) { return key(a) < key(b); });
      HIPCHK(hipMemcpy(dl, items.data(), (size_t)n * sizeof(VisItem), hipMemcpyHostToDevice));
    }
  }
  if (!fused) {  // (the serial role as a launch of its own: the round-1 layout, kept for A/B in the diagnostic build)
    st = alloc_rank((uint32_t)(npix * (size_t)S), par, next ? &ahead_b : nullptr, true);
    if (st != RATSDF_OK) return st;
  }
#endif

  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // (sampled: events perturb the stream; mode 2 times every frame, for latency distributions)
  const bool timed = profiling && (prof_mode == 2 || prof_frame++ % 4 == 0);
  if (timed) {
    if (prof_used == prof_events.size()) {
      hipEvent_t a, b;
      HIPCHK(hipEventCreate(&a));
      HIPCHK(hipEventCreate(&b));
      prof_events.emplace_back(a, b);
    }
    ev0 = prof_events[prof_used].first;
    ev1 = prof_events[prof_used].second;
    ++prof_used;
  }
  const unsigned integrate_grid = g.grid;
  // hipExtLaunchKernelGGL attaches the two events to the dispatch itself: their difference is the
  // kernel's own start-to-end time (what rocprofv3 reports), without the barrier packets that
  // hipEventRecord before / after a launch would add (~3 us here).  Null events = a plain launch.
  const unsigned extra_c = ((ahead_c.n_tiles + 3) / 4 + 7u) & ~7u;  // whole groups of 8 (XCD mapping)
  const uint32_t n_serial_wg = fused ? 8u : 0u;  // the first of them works
  const uint32_t commit_rot = fused ? commit_rotation(integrate_grid, integrate_grid) : 0u;
  IntegArgs ia;
  ia.rgbw = pool.rgbw;
  ia.tsdf = pool.tsdf;
  ia.segm = pool.segm;
  ia.texA = texA[par];
  ia.texB = texB[par];
  ia.vis = vis + kFreshCap;
  ia.seg_cap = seg_cap;
  ia.F = &ctl->fr[par];
  ia.upd_wg = upd_wg[par];
  ia.par = par;
#define RATSDF_LAUNCH_INTEGRATE(V, T, NT)                                                              \
  hipExtLaunchKernelGGL((k_integrate<V, T>), dim3(integrate_grid + n_serial_wg + extra_c), dim3(NT), 0, \
                        stream, ev0, ev1, 0, ia, P, (EnginePtr)d_eng, (uint32_t)integrate_grid,         \
                        n_serial_wg, (uint32_t)extra_c, commit_rot, ahead_c)
  // (the voxels-per-lane variants 1 / 4 / 8 are tuning options: without the front-tail path)
#ifdef RATSDF_STAMPS
  switch (vpl) {
    case 1: RATSDF_LAUNCH_INTEGRATE(1, false, 512); break;
    case 8: RATSDF_LAUNCH_INTEGRATE(8, false, RATSDF_INTEG_NT); break;
    case 4: RATSDF_LAUNCH_INTEGRATE(4, false, RATSDF_INTEG_NT); break;
    default:
      if (tab.tail_on) RATSDF_LAUNCH_INTEGRATE(2, true, RATSDF_INTEG_NT);
      else RATSDF_LAUNCH_INTEGRATE(2, false, RATSDF_INTEG_NT);
  }
#else
  RATSDF_LAUNCH_INTEGRATE(2, false, RATSDF_INTEG_NT);  // the one form the product ships
#endif
#undef RATSDF_LAUNCH_INTEGRATE

  HIPCHK(hipGetLastError());
  pending = true;
  if (profiling && prof_used >= (prof_mode == 2 ? 60000u : 4096u)) return drain_profile(false);
  return RATSDF_OK;
}

// `final`: nothing follows the timed frames (a read-out).  An intermediate drain in mode 2 (the event pool is
// full) keeps the LAST pair for the next drain: its period ends at the start of a frame that has not been
// launched yet, and prof_k_us / prof_period_us must stay index-aligned (ratsdf_profile_read_frames).
int ratsdf_engine::drain_profile(bool final) {
  if (!prof_used) return RATSDF_OK;
  HIPCHK(hipStreamSynchronize(stream));
  const bool keep_last = prof_mode == 2 && !final && prof_used > 1;
  const size_t n = keep_last ? prof_used - 1 : prof_used;
  for (size_t i = 0; i < n; ++i) {
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, prof_events[i].first, prof_events[i].second));
    prof_ms += ms;
    ++prof_n;
    if (prof_mode == 2) {
      prof_k_us.push_back(ms * 1e3f);
      float gap = 0;  // consecutive frames: start of this frame's k_integrate to the next one's (last frame: 0)
      if (i + 1 < prof_used) HIPCHK(hipEventElapsedTime(&gap, prof_events[i].first, prof_events[i + 1].first));
      prof_period_us.push_back(gap * 1e3f);
    }
  }
  if (keep_last) std::swap(prof_events[0], prof_events[prof_used - 1]);
  prof_used = keep_last ? 1 : 0;
  return RATSDF_OK;
}

// The sticky error word, after everything enqueued so far.  No device-to-host copy in the ordinary call: set_error
// (kernels_alloc.h) also raises a flag in page-locked host memory (h_err[0], through Ctl::err_flag), which is read
// once the stream has drained; only when it is set is the device word -- the FIRST error -- fetched (into h_err[1]:
// page-locked as well; a copy into pageable memory goes through the runtime's staging path).  A synchronising call on
// an idle engine went from 20 - 25 us to a stream synchronisation; the per-frame convention of TSDFGrid::Integrate
// (ratsdf_integrate_device + ratsdf_synchronize) from 59 - 61 us per frame to the figure in profiles/r05_sync_path.txt.
int ratsdf_engine::sticky() {
  HIPCHK(hipStreamSynchronize(stream));
  if (*(volatile uint32_t*)h_err == 0u) return RATSDF_OK;
  HIPCHK(hipMemcpyAsync(h_err + 1, &ctl->error, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
  HIPCHK(hipStreamSynchronize(stream));
  return (int)((volatile uint32_t*)h_err)[1];
}

// A few words of device memory for the host, after everything enqueued so far: through the page-locked landing buffer
// (h_err + 16 words on), not straight into the caller's pageable memory (the runtime's staging path: ~20 us per call,
// and these are the calls a per-frame logger makes -- NumActiveBlock, the frame statistics).
int ratsdf_engine::read_small(void* dst, const void* dev_src, size_t bytes) {
  if (bytes > 384) return RATSDF_ERR_BAD_ARGUMENT;
  HIPCHK(hipMemcpyAsync(h_err + 16, dev_src, bytes, hipMemcpyDeviceToHost, stream));
  HIPCHK(hipStreamSynchronize(stream));
  memcpy(dst, h_err + 16, bytes);
  return RATSDF_OK;
}

// Non-finite camera parameters are refused at the boundary (RATSDF_ERR_BAD_ARGUMENT).  The reference
// would integrate garbage (a NaN pose projects every voxel to pixel (0, 0): float -> int of NaN is 0 in
// CUDA); the engine's short pixel pick (kernels_integrate.h) reproduces the reference for
// every finite pose -- including voxels in the camera plane, z == 0 -- and relies on this check for the
// rest.
static bool finite_frame(const ratsdf_intrinsics& K, const ratsdf_pose& T, float max_depth) {
  const float v[] = {K.fx, K.fy, K.cx, K.cy, T.qx, T.qy, T.qz, T.qw, T.tx, T.ty, T.tz, max_depth};
  for (float x : v)
    if (!std::isfinite(x)) return false;
  return true;
}

void ratsdf_engine::free_graph(BatchGraph& g) {
  if (g.exec) (void)hipGraphExecDestroy(g.exec);
  if (g.graph) (void)hipGraphDestroy(g.graph);
  if (g.d_jobs) (void)hipFree(g.d_jobs);
  for (int i = 0; i < 2; ++i) {
    if (g.h_jobs[i]) (void)hipHostFree(g.h_jobs[i]);
    if (g.ev[i]) (void)hipEventDestroy(g.ev[i]);
  }
  g = BatchGraph();
}

// The graph of an n-frame batch at H x W (built on first use, a few kept): k_cand_g for the first frame, then
// k_front_g / k_integrate_g per frame with the look-ahead shares frame() would choose, one member (blockIdx.y = 0),
// operands from d_eng and the graph's own job table.
int ratsdf_engine::batch_graph(int n, int H, int W, BatchGraph** out) {
  *out = nullptr;
  for (auto& g : graphs)
    if (g.H == H && g.W == W && g.n == n) {
      g.last_use = ++graph_clock;
      *out = &g;
      return RATSDF_OK;
    }
  for (const auto& f : graph_failed)
    if (f.H == H && f.W == W && f.n == n) return RATSDF_ERR_DEVICE;  // (reported when it happened)
  // A graph per batch LENGTH: a caller that drains a queue (ratsdf::TSDFSystem hands over 1 .. 32 frames) meets
  // every length sooner or later, so the cache holds all of them for a couple of image sizes before anything is
  // evicted (a graph is ~0.2 KiB of job table per frame plus the executable graph).
  if (graphs.size() >= 72) {  // least recently used out
    size_t victim = 0;
    for (size_t i = 1; i < graphs.size(); ++i)
      if (graphs[i].last_use < graphs[victim].last_use) victim = i;
    HIPCHK(hipStreamSynchronize(stream));
    free_graph(graphs[victim]);
    graphs.erase(graphs.begin() + (long)victim);
  }
  BatchGraph g;
  g.H = H;
  g.W = W;
  g.n = n;
  auto fail = [&](const char* what) {
    fprintf(stderr, "[ratsdf] batch graph %dx%d x %d: %s failed; batches of this shape are launched frame by frame\n",
            W, H, n, what);
    free_graph(g);
    graph_failed.push_back(GraphShape{H, W, n});
    return RATSDF_ERR_DEVICE;
  };
  if (hipMalloc(&g.d_jobs, (size_t)n * sizeof(FrameJob)) != hipSuccess) return fail("hipMalloc");
  for (int i = 0; i < 2; ++i)
    if (hipHostMalloc(&g.h_jobs[i], (size_t)n * sizeof(FrameJob), hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&g.ev[i], hipEventDisableTiming) != hipSuccess)
      return fail("staging allocation");
  const size_t npix = (size_t)H * W;
  const int split = vpl == 1 ? 100 : (int)(cand_split_env ? cand_split : (npix >= 600000 ? (unsigned)RATSDF_CAND_SPLIT_HD : cand_split));
  const Geom g1 = geometry(H, W, true, split, 0);
  const Geom g0 = geometry(H, W, false, 0, 0);
  const uint32_t n_serial_wg = 8u;
  const uint32_t commit_rot = commit_rotation(g0.grid, g0.grid);
  [[maybe_unused]] const uint32_t tail = (tab.tail_on ? 1u : 0u) | front_prio;
  EnginePtr engs = (EnginePtr)d_eng;
  if (hipStreamBeginCapture(stream, hipStreamCaptureModeRelaxed) != hipSuccess) return fail("hipStreamBeginCapture");
  {
    AheadGeom all = g0.a;
    all.first_tile = 0;
    all.n_tiles = g0.n_cand_wg * 4;
    hipLaunchKernelGGL(k_cand_g, dim3(g0.n_cand_wg, 1), dim3(256), 0, stream, engs, (JobPtr)g.d_jobs, all);
  }
  for (int f = 0; f < n; ++f) {
    const bool has_next = f + 1 < n;
    const Geom& gg = has_next ? g1 : g0;
    JobPtr cur = (JobPtr)(g.d_jobs + f);
    JobPtr nxt = (JobPtr)(g.d_jobs + (has_next ? f + 1 : f));
#ifdef RATSDF_STAMPS
    if (tab.tail_on)
      hipLaunchKernelGGL(k_front_g<true>, dim3(gg.n_front_wg, 1), dim3(256), 0, stream, engs, cur, nxt,
                         (uint32_t)gg.n_vis_wg, (uint32_t)gg.parts, tail, gg.a);
    else
#endif
      hipLaunchKernelGGL(k_front_g<false>, dim3(gg.n_front_wg, 1), dim3(256), 0, stream, engs, cur, nxt,
                         (uint32_t)gg.n_vis_wg, (uint32_t)gg.parts, 0u, gg.a);
    const unsigned extra_c = ((gg.c.n_tiles + 3) / 4 + 7u) & ~7u;
#define RATSDF_GRAPH_INTEGRATE(V, T)                                                                                \
  hipLaunchKernelGGL((k_integrate_g<V, T>), dim3(gg.grid + n_serial_wg + extra_c, 1), dim3(RATSDF_INTEG_NT), 0, stream, \
                     engs, cur, nxt, (uint32_t)gg.grid, n_serial_wg, (uint32_t)extra_c, commit_rot, gg.c)
#ifdef RATSDF_STAMPS
    switch (vpl) {
      case 8: RATSDF_GRAPH_INTEGRATE(8, false); break;
      case 4: RATSDF_GRAPH_INTEGRATE(4, false); break;
      default:
        if (tab.tail_on) RATSDF_GRAPH_INTEGRATE(2, true);
        else RATSDF_GRAPH_INTEGRATE(2, false);
    }
#else
    RATSDF_GRAPH_INTEGRATE(2, false);
#endif
#undef RATSDF_GRAPH_INTEGRATE
  }
  const hipError_t launch_err = hipGetLastError();
  if (hipStreamEndCapture(stream, &g.graph) != hipSuccess || launch_err != hipSuccess || !g.graph)
    return fail("capture");
  if (hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0) != hipSuccess) return fail("hipGraphInstantiate");
  g.last_use = ++graph_clock;
  graphs.push_back(g);
  *out = &graphs.back();
  return RATSDF_OK;
}

// ================================== C ABI =====================================================
extern "C" {

int ratsdf_create_ex(const ratsdf_config* cfg, ratsdf_engine** out) {
  if (!cfg || !out) return RATSDF_ERR_BAD_ARGUMENT;
  if (!(cfg->voxel_size > 0) || !(cfg->truncation > 0)) return RATSDF_ERR_BAD_ARGUMENT;
  const int bb = cfg->block_bits ? cfg->block_bits : RATSDF_DEFAULT_BLOCK_BITS;
  const int kb = cfg->bucket_bits ? cfg->bucket_bits : RATSDF_DEFAULT_BUCKET_BITS;
  if (bb < 1 || bb > 24 || kb < 9 || kb > 26) return RATSDF_ERR_BAD_ARGUMENT;
  if (cfg->shard_count > 1 && (cfg->shard_rank < 0 || cfg->shard_rank >= cfg->shard_count))
    return RATSDF_ERR_BAD_ARGUMENT;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return RATSDF_ERR_NO_DEVICE;
  if (cfg->device < 0 || cfg->device >= ndev) return RATSDF_ERR_BAD_ARGUMENT;
  DeviceGuard guard(cfg->device);  // the caller's current device is restored on return
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  ratsdf_engine* e = new (std::nothrow) ratsdf_engine();
  if (!e) return RATSDF_ERR_DEVICE;
  e->device = cfg->device;
  e->vs = cfg->voxel_size;
  e->trunc = cfg->truncation;
  e->block_bits = bb;
  e->bucket_bits = kb;
  e->shard_rank = cfg->shard_rank;
  e->shard_count = cfg->shard_count > 1 ? cfg->shard_count : 1;
  e->shard_slab_bits = cfg->shard_slab_bits > 0 ? cfg->shard_slab_bits : 2;
  e->S = (int)ceilf(2.f * e->trunc / e->vs / RATSDF_BLOCK_LEN) + 2;
  // What the environment may change in the shipped library: the two documented behaviours below (and
  // RATSDF_COPY_STREAMS in ratsdf_integrate_batch).  Every tuning and ablation switch of the measurements in
  // DESIGN.md / profiles/ -- voxels per lane, look-ahead split, grid, the serial role as a launch of its own or at
  // the tail of k_front, fault injection -- exists in the diagnostic build only (make stamps, -DRATSDF_STAMPS),
  // together with the kernel variants it selects.
  if (const char* v = getenv("RATSDF_SYNC_INTEGRATE")) e->sync_integrate = atoi(v) != 0;
  if (const char* v = getenv("RATSDF_GRAPH")) e->use_graphs = atoi(v) != 0;
#ifdef RATSDF_STAMPS
  if (const char* v = getenv("RATSDF_VPL")) {
    const int x = atoi(v);
    if (x == 1 || x == 2 || x == 4 || x == 8) e->vpl = x;
  }
  if (const char* v = getenv("RATSDF_DEBUG")) e->debug = atoi(v);
  if (const char* v = getenv("RATSDF_CAND_SPLIT")) {  // "a" or "a,b": percent in k_front[, k_alloc_rank]
    const int x = atoi(v);
    if (x >= 0 && x <= 100) {
      e->cand_split = (unsigned)x;
      e->cand_split_b = 100u - (unsigned)x;
      e->cand_split_env = true;
      if (const char* c = strchr(v, ',')) {
        const int y = atoi(c + 1);
        if (y >= 0 && x + y <= 100) e->cand_split_b = (unsigned)y;
      }
    }
  }
  if (const char* v = getenv("RATSDF_CAND_PARTS")) {
    const int x = atoi(v);
    if (x >= 1 && x <= 64) e->cand_parts_env = (unsigned)x;
  }
  if (const char* v = getenv("RATSDF_COMMIT_ROT")) e->commit_rot_env = atoi(v);
  if (const char* v = getenv("RATSDF_FUSED_SERIAL")) e->fused_serial = atoi(v) != 0;  // 0: k_alloc_rank launch
  if (const char* v = getenv("RATSDF_FRONT_TAIL")) e->front_tail = atoi(v) != 0;  // 0: the role always in k_integrate
  if (const char* v = getenv("RATSDF_FRONT_PRIO")) e->front_prio = atoi(v) ? 2u : 0u;
  if (const char* v = getenv("RATSDF_INLINE_CAND")) e->inline_off = atoi(v) == 0;
  if (const char* v = getenv("RATSDF_SORT_LISTS")) e->sort_lists = atoi(v) != 0;
  if (const char* v = getenv("RATSDF_SERIAL_LDS")) {
    const int x = atoi(v);
    if (x >= kSerialLdsBytes && x <= 160 * 1024) e->serial_lds = (unsigned)x;
  }
  if (const char* v = getenv("RATSDF_CAND_WGS")) {
    const int x = atoi(v);
    if (x >= 1 && x <= 4096) e->cand_wgs = (unsigned)x;
  }
  if (const char* v = getenv("RATSDF_GRID")) {
    const int x = atoi(v);
    if (x >= 64 && x <= 65536) {
      e->integrate_grid = (unsigned)x;
      e->grid_from_env = true;
    }
  }
#endif
  Table& t = e->tab;
  t.tail_on = (e->front_tail && e->fused_serial && e->vpl == 2) ? 1u : 0u;
  t.delta_on = 0;
  t.num_block = 1 << bb;
  t.num_bucket = 1u << kb;
  t.num_entry = t.num_bucket << 1;
  t.bucket_mask = t.num_bucket - 1;
  t.entry_mask = t.num_entry - 1;
  const uint32_t occ_words = (t.num_entry + 63) / 64;
  e->nwg = (occ_words + kVisWG - 1) / kVisWG;
  e->dwords = (t.num_entry + 31) / 32;  // delete bitmap is indexed by hash entry
  const size_t nvox = (size_t)t.num_block << 9;

#define CREATE_CHK(expr)                 \
  do {                                   \
    if ((expr) != hipSuccess) {          \
      fprintf(stderr, "[ratsdf] create failed: %s\n", #expr); \
      e->free_all();                     \
      delete e;                          \
      return RATSDF_ERR_DEVICE;          \
    }                                    \
  } while (0)

  CREATE_CHK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
  CREATE_CHK(hipMalloc(&t.entries, (size_t)t.num_entry * sizeof(Entry)));
  CREATE_CHK(hipMalloc(&t.claim, (size_t)t.num_bucket * 4));
  // occupancy bitmap, and behind it the dirty bitmap of the directory delta (device_types.h: Table)
  CREATE_CHK(hipMalloc(&t.occ, (size_t)occ_words * 8 * 2));
  CREATE_CHK(hipMemsetAsync(t.occ + occ_words, 0, (size_t)occ_words * 8, e->stream));
  // the live blocks by pool index (device_types.h: Table::active): every slot empty (idx = -1)
  CREATE_CHK(hipMalloc(&t.active, (size_t)t.num_block * sizeof(VisItem)));
  CREATE_CHK(hipMemsetAsync(t.active, 0xFF, (size_t)t.num_block * sizeof(VisItem), e->stream));
  t.del_cap = (uint32_t)t.num_block;
  CREATE_CHK(hipMalloc(&t.del_log, (size_t)t.del_cap * sizeof(uint2)));
  CREATE_CHK(hipMalloc(&t.del_count, 128));
  CREATE_CHK(hipMemsetAsync(t.del_count, 0, 128, e->stream));
  CREATE_CHK(hipMalloc(&e->pool.rgbw, nvox * 4));
  CREATE_CHK(hipMalloc(&e->pool.tsdf, nvox * 4));
  CREATE_CHK(hipMalloc(&e->pool.segm, nvox * 4));
  CREATE_CHK(hipMalloc(&e->pool.heap, (size_t)t.num_block * 4));
  CREATE_CHK(hipMalloc(&e->ctl, sizeof(Ctl)));
  CREATE_CHK(hipMalloc(&e->d_stats, sizeof(ratsdf_frame_stats)));
  CREATE_CHK(hipHostMalloc(&e->h_err, 512, hipHostMallocDefault));
  CREATE_CHK(hipMalloc(&e->d_eng, sizeof(EngineDev)));
  CREATE_CHK(hipMalloc(&e->slow, (size_t)kSlowCap * sizeof(SlowRequest)));
  CREATE_CHK(hipMalloc(&e->xlocks, (size_t)kXLockCap * sizeof(XLock)));
  CREATE_CHK(hipMalloc(&e->sort_scratch, (size_t)kSlowSortCap * sizeof(unsigned long long) +
                                               (size_t)kSlowPlanCap * sizeof(SlowPlan)));  // sort keys | plans
  CREATE_CHK(hipMalloc(&e->serial_scratch, (size_t)kSerialLdsBytes));
  CREATE_CHK(hipMalloc(&e->masks, (size_t)e->nwg * kVisWG * 8));
  CREATE_CHK(hipMalloc(&e->wg_count, (size_t)e->nwg * 4));
  // 8 per-XCD work lists; a segment = kFreshCap items for the list's new blocks (front_tail_role) in front of
  // room for every block of the pool (device_types.h: kFreshCap); queries use the buffer as one flat list
  e->seg_cap = (uint32_t)t.num_block + kFreshCap;
  e->vis_cap = kFreshCap + kNumLists * e->seg_cap;
  CREATE_CHK(hipMalloc(&e->vis, (size_t)e->vis_cap * sizeof(VisItem)));
  for (int i = 0; i < 2; ++i) {
    CREATE_CHK(hipMalloc(&e->del_list[i], (size_t)t.num_block * sizeof(DelItem)));
    CREATE_CHK(hipMalloc(&e->upd_wg[i], kUpdCounters * 4));
    CREATE_CHK(hipMemsetAsync(e->upd_wg[i], 0, kUpdCounters * 4, e->stream));
    CREATE_CHK(hipMalloc(&e->slowdel[i], (size_t)kSlowDelCap * sizeof(SlowDelete)));
  }
  CREATE_CHK(hipMalloc(&e->win_ranks, (size_t)kFusedRank * 4));
  CREATE_CHK(hipMalloc(&t.dclaim, (size_t)t.num_bucket * 4));
  CREATE_CHK(hipMemsetAsync(t.dclaim, 0xFF, (size_t)t.num_bucket * 4, e->stream));
  e->dwords = (e->dwords + kGroupWords - 1) / kGroupWords * kGroupWords;
  const uint32_t dsum_words = (e->dwords / kGroupWords + 31) / 32;
  CREATE_CHK(hipMalloc(&e->dbitmap, (size_t)e->dwords * 4));
  CREATE_CHK(hipMalloc(&e->dsummary, (size_t)dsum_words * 4));
  CREATE_CHK(hipMalloc(&e->dprefix, (size_t)e->dwords * 4));
  CREATE_CHK(hipMalloc(&e->cand_count, 2 * kCandSegs * kCandCountStride * 4));
  CREATE_CHK(hipMemsetAsync(e->cand_count, 0, 2 * kCandSegs * kCandCountStride * 4, e->stream));
  for (int i = 0; i < 2; ++i) e->cand[i].count = e->cand_count + i * kCandSegs * kCandCountStride;
  // voxel memory starts zeroed (defined value for the reference's uninitialised rgb)
  CREATE_CHK(hipMemsetAsync(e->pool.rgbw, 0, nvox * 4, e->stream));
  CREATE_CHK(hipMemsetAsync(e->pool.tsdf, 0, nvox * 4, e->stream));
  CREATE_CHK(hipMemsetAsync(e->pool.segm, 0, nvox * 4, e->stream));
  CREATE_CHK(hipMemsetAsync(e->ctl, 0, sizeof(Ctl), e->stream));
  CREATE_CHK(hipMemsetAsync(e->d_stats, 0, sizeof(ratsdf_frame_stats), e->stream));
  CREATE_CHK(hipMemsetAsync(e->dbitmap, 0, (size_t)e->dwords * 4, e->stream));
  CREATE_CHK(hipMemsetAsync(e->dsummary, 0, (size_t)dsum_words * 4, e->stream));
  hipLaunchKernelGGL(k_init_table, dim3((t.num_entry + 255) / 256), dim3(256), 0, e->stream,
                     t.entries, t.claim, t.occ, t.num_entry, t.num_bucket);
  hipLaunchKernelGGL(k_init_heap, dim3((t.num_block + 255) / 256), dim3(256), 0, e->stream,
                     e->pool.heap, t.num_block);
  const int32_t nf = t.num_block;
  CREATE_CHK(hipMemcpyAsync(&e->ctl->num_free, &nf, 4, hipMemcpyHostToDevice, e->stream));
  CREATE_CHK(hipMemcpyAsync(&e->ctl->free_low, &nf, 4, hipMemcpyHostToDevice, e->stream));
  {  // the error flag's home in page-locked host memory, as the device addresses it (sticky())
    e->h_err[0] = e->h_err[1] = 0u;
    void* flag_dev = nullptr;
    CREATE_CHK(hipHostGetDevicePointer(&flag_dev, e->h_err, 0));
    CREATE_CHK(hipMemcpyAsync(&e->ctl->err_flag, &flag_dev, sizeof(flag_dev), hipMemcpyHostToDevice, e->stream));
  }
  CREATE_CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_alloc_rank),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 8 * 1024));
  hipLaunchKernelGGL(k_check_block_threads, dim3(3), dim3(192), 0, e->stream, &e->ctl->n_sel);
  uint32_t bt = 0;
  CREATE_CHK(hipMemcpyAsync(&bt, &e->ctl->n_sel, 4, hipMemcpyDeviceToHost, e->stream));
  CREATE_CHK(hipStreamSynchronize(e->stream));
  CREATE_CHK(hipGetLastError());
  if (bt != 192u) {
    fprintf(stderr, "[ratsdf] create failed: the workgroup size is not where block_threads() reads it (code object ABI?)\n");
    e->free_all();
    delete e;
    return RATSDF_ERR_DEVICE;
  }
  if (e->upload_record() != RATSDF_OK) {
    e->free_all();
    delete e;
    return RATSDF_ERR_DEVICE;
  }
#undef CREATE_CHK
  *out = e;
  return RATSDF_OK;
}

int ratsdf_create(float voxel_size, float truncation, int device, ratsdf_engine** out) {
  ratsdf_config c;
  memset(&c, 0, sizeof(c));
  c.voxel_size = voxel_size;
  c.truncation = truncation;
  c.device = device;
  return ratsdf_create_ex(&c, out);
}

int ratsdf_destroy(ratsdf_engine* e) {
  DeviceGuard guard(e ? e->device : -1);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  if (!e) return RATSDF_ERR_BAD_ARGUMENT;
  e->free_all();
  delete e;
  return RATSDF_OK;
}

int ratsdf_integrate_device(ratsdf_engine* e, const void* d_rgb, const void* d_depth,
                            const void* d_ht, const void* d_lt, int height, int width,
                            float max_depth, const ratsdf_intrinsics* K, const ratsdf_pose* T) {
  DeviceGuard guard(e ? e->device : -1);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  if (!e || !d_rgb || !d_depth || !K || !T || height <= 0 || width <= 0)
    return RATSDF_ERR_BAD_ARGUMENT;
  if (!finite_frame(*K, *T, max_depth)) return RATSDF_ERR_BAD_ARGUMENT;
  if (!d_ht || !d_lt) d_ht = d_lt = nullptr;
  const ratsdf_engine::FrameIn in{d_rgb, d_depth, d_ht, d_lt, K, T};
  return e->frame(in, nullptr, height, width, max_depth);
}

int ratsdf_integrate_device_batch(ratsdf_engine* e, int n, const void* const* d_rgb,
                                  const void* const* d_depth, const void* const* d_ht,
                                  const void* const* d_lt, int height, int width, float max_depth,
                                  const ratsdf_intrinsics* K, const ratsdf_pose* T) {
  DeviceGuard guard(e ? e->device : -1);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  if (!e || n < 0 || (n > 0 && (!d_rgb || !d_depth || !K || !T)) || height <= 0 || width <= 0)
    return RATSDF_ERR_BAD_ARGUMENT;
  for (int i = 0; i < n; ++i)
    if (!d_rgb[i] || !d_depth[i] || !finite_frame(K[i], T[i], max_depth)) return RATSDF_ERR_BAD_ARGUMENT;
  auto input = [&](int i) {
    const void* ht = (d_ht && d_lt) ? d_ht[i] : nullptr;
    const void* lt = (d_ht && d_lt) ? d_lt[i] : nullptr;
    if (!ht || !lt) ht = lt = nullptr;
    return ratsdf_engine::FrameIn{d_rgb[i], d_depth[i], ht, lt, &K[i], &T[i]};
  };
  // The captured form: one graph launch for the whole batch (not while individual launches carry profiling
  // events, not for the 512-thread voxel-per-lane variant, not with the serial role as a launch of its own).
  // (profiling, mode 1: every fourth batch is launched frame by frame and carries the events -- a sample of the
  // same stream inside the same timed region; mode 2 times every frame: no graphs)
  const bool sampled = e->profiling && (e->prof_mode == 2 || (e->prof_batch++ & 3u) == 0);
  if (e->use_graphs && n >= 2 && !sampled && !e->cand_ready && e->fused_serial && e->vpl != 1 &&
      (size_t)height * width * (size_t)e->S < 0xFFFFFFFFull) {
    const size_t npix = (size_t)height * width;
    int st = e->ensure_image(npix, npix * (size_t)e->S);
    if (st != RATSDF_OK) return st;
    ratsdf_engine::BatchGraph* g = nullptr;
    if (e->batch_graph(n, height, width, &g) == RATSDF_OK && g) {
      const unsigned turn = g->turn++ & 1u;
      HIPCHK(hipEventSynchronize(g->ev[turn]));  // the copy that last used this staging table is done
      FrameJob* hj = g->h_jobs[turn];
      for (int i = 0; i < n; ++i) {
        const ratsdf_engine::FrameIn in = input(i);
        FrameJob& j = hj[i];
        j.P = e->frame_params(in, height, width, max_depth);
        j.depth = (const float*)in.depth;
        j.rgb = (const uint8_t*)in.rgb;
        j.ht = (const float*)in.ht;
        j.lt = (const float*)in.lt;
        j.par = (e->parity + (unsigned)i) & 1u;
        j.pad = 0;
      }
      if (hipMemcpyAsync(g->d_jobs, hj, (size_t)n * sizeof(FrameJob), hipMemcpyHostToDevice, e->stream) != hipSuccess ||
          hipEventRecord(g->ev[turn], e->stream) != hipSuccess)
        return RATSDF_ERR_DEVICE;
      if (hipGraphLaunch(g->exec, e->stream) != hipSuccess) {
        e->abandon_pipeline();
        return RATSDF_ERR_DEVICE;
      }
      e->parity = (e->parity + (unsigned)n) & 1u;
      e->cand_ready = false;
      e->pending = true;
      return RATSDF_OK;
    }
  }
  // (mode 1 samples frames 2, 6, 10, ... of the batch, never its first: the start stamp of a dispatch that finds
  // the queue idle is taken early -- events showed 100+ us for such a frame where rocprofv3's trace of the same
  // launch shows an ordinary one, tools/event_probe.py)
  if (e->profiling && e->prof_mode != 2 && n > 2) e->prof_frame = 2;
  for (int i = 0; i < n; ++i) {
    const ratsdf_engine::FrameIn cur = input(i);
    ratsdf_engine::FrameIn nxt{};
    if (i + 1 < n) nxt = input(i + 1);
    const int st = e->frame(cur, i + 1 < n ? &nxt : nullptr, height, width, max_depth);
    if (st != RATSDF_OK) {
      e->abandon_pipeline();
      return st;
    }
  }
  return RATSDF_OK;
}

// Builds what the first ratsdf_integrate_device_batch of this shape would build on the spot -- image-sized scratch
// and the HIP graph of an n-frame batch -- so that a caller with a deadline does not pay for allocation, capture
// and instantiation inside its first batch.  Nothing is launched; a shape the engine launches frame by frame
// anyway (n < 2, graphs off) only gets its scratch.
int ratsdf_prepare_device_batch(ratsdf_engine* e, int n, int height, int width) {
  DeviceGuard guard(e ? e->device : -1);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  if (!e || n < 0 || height <= 0 || width <= 0) return RATSDF_ERR_BAD_ARGUMENT;
  const size_t npix = (size_t)height * width;
  if (npix * (size_t)e->S >= 0xFFFFFFFFull) return RATSDF_ERR_BAD_ARGUMENT;
  const int st = e->ensure_image(npix, npix * (size_t)e->S);
  if (st != RATSDF_OK) return st;
  if (e->use_graphs && n >= 2 && !e->cand_ready && e->fused_serial && e->vpl != 1) {
    ratsdf_engine::BatchGraph* g = nullptr;
    (void)e->batch_graph(n, height, width, &g);  // (a failed capture is remembered: such batches go frame by frame)
  }
  return RATSDF_OK;
}

int ratsdf_integrate(ratsdf_engine* e, const uint8_t* rgb, const float* depth, const float* ht,
                     const float* lt, int height, int width, float max_depth,
                     const ratsdf_intrinsics* K, const ratsdf_pose* T) {
  DeviceGuard guard(e ? e->device : -1);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  if (!e || !rgb || !depth || !K || !T || height <= 0 || width <= 0)
    return RATSDF_ERR_BAD_ARGUMENT;
  if (!finite_frame(*K, *T, max_depth)) return RATSDF_ERR_BAD_ARGUMENT;
  if (!ht || !lt) ht = lt = nullptr;  // modules/tsdf_module.cc:27-31
  const size_t npix = (size_t)height * width;
  int st = e->ensure_stage(npix);
  if (st != RATSDF_OK) return st;
  // The slots of the staging ring, one per call in turn (layout of a slot: depth | ht | lt | rgb).  The call
  // returns when the caller's images sit in the slot's page-locked memory: the upload runs on a copy stream
  // (it overlaps the previous frame's kernels), the frame's launches follow it on the engine's stream, and
  // nothing here waits for the GPU unless the ring is full -- the slot's previous upload (8 calls ago) must
  // have left its host memory before it is overwritten.  (Until round 4 every call ended with a stream
  // synchronisation: 2 700 frames/s at 640x480, a third of it the single-threaded staging copy.)
  // (the slot stride is the ring's, not this call's: a smaller image after a larger one must land in the SAME slot
  // memory the slot's events guard -- with a per-call stride its bytes would fall inside other slots that frames of
  // earlier calls, which are not waited for, may still be reading)
  const size_t slot_bytes = e->stage_pix * 16;
  const int slot = (int)(e->stage_no++ % kStageSlots);
  uint8_t* h = e->h_stage + (size_t)slot * slot_bytes;
  uint8_t* d = e->d_stage + (size_t)slot * slot_bytes;
  HIPCHK(hipEventSynchronize(e->stage_ev[slot]));  // (an event never recorded counts as complete)
  if (!e->copy_pool) e->copy_pool = new (std::nothrow) HostCopyPool(3);
  HostCopyPool::Piece pieces[8];
  int np = 0;
  auto add = [&](size_t off, const void* src, size_t bytes, int parts) {  // `parts` pieces of ~equal size
    const size_t step = ((bytes + parts - 1) / parts + 63) & ~(size_t)63;
    for (size_t o = 0; o < bytes; o += step)
      pieces[np++] = HostCopyPool::Piece{h + off + o, (const uint8_t*)src + o, std::min(step, bytes - o)};
  };
  if (ht) {
    add(0, depth, npix * 4, 1);
    add(npix * 4, ht, npix * 4, 1);
    add(npix * 8, lt, npix * 4, 1);
    add(npix * 12, rgb, npix * 3, 1);
  } else {
    add(0, depth, npix * 4, 2);
    add(npix * 12, rgb, npix * 3, 2);
  }
  if (e->copy_pool) {
    e->copy_pool->copy(pieces, np);
  } else {
    for (int i = 0; i < np; ++i) memcpy(pieces[i].dst, pieces[i].src, pieces[i].bytes);
  }
  hipStream_t cs = (slot & 1) ? e->copy_stream2 : e->copy_stream;
  // whatever happens from here on, nothing stays queued that reads a half-prepared slot
  auto fail = [&](int status) {
    (void)hipStreamSynchronize(e->copy_stream);
    (void)hipStreamSynchronize(e->copy_stream2);
    e->abandon_pipeline();
    return status;
  };
  // the slot's device memory was last read by the frame that used it kStageSlots frames ago (of either host-image
  // entry point): its use_ev.  Nothing else on the engine's stream reads the staging slots.
  if (hipStreamWaitEvent(cs, e->use_ev[slot], 0) != hipSuccess) return fail(RATSDF_ERR_DEVICE);
  if (ht) {
    if (hipMemcpyAsync(d, h, npix * 15, hipMemcpyHostToDevice, cs) != hipSuccess) return fail(RATSDF_ERR_DEVICE);
  } else {
    if (hipMemcpyAsync(d, h, npix * 4, hipMemcpyHostToDevice, cs) != hipSuccess ||
        hipMemcpyAsync(d + npix * 12, h + npix * 12, npix * 3, hipMemcpyHostToDevice, cs) != hipSuccess)
      return fail(RATSDF_ERR_DEVICE);
  }
  if (hipEventRecord(e->stage_ev[slot], cs) != hipSuccess ||
      hipStreamWaitEvent(e->stream, e->stage_ev[slot], 0) != hipSuccess)
    return fail(RATSDF_ERR_DEVICE);
  const ratsdf_engine::FrameIn in{d + npix * 12, d, ht ? d + npix * 4 : nullptr,
                                  ht ? d + npix * 8 : nullptr, K, T};
  st = e->frame(in, nullptr, height, width, max_depth);
  if (st != RATSDF_OK) return fail(st);
  if (hipEventRecord(e->use_ev[slot], e->stream) != hipSuccess) return fail(RATSDF_ERR_DEVICE);
  if (!e->sync_integrate) return RATSDF_OK;
  return e->sticky();  // cudaStreamSynchronize(stream_), voxel_tsdf.cu:450
}

int ratsdf_integrate_batch(ratsdf_engine* e, int n, const uint8_t* const* rgb,
                           const float* const* depth, const float* const* ht,
                           const float* const* lt, int height, int width, float max_depth,
                           const ratsdf_intrinsics* K, const ratsdf_pose* T, int pinned) {
  DeviceGuard guard(e ? e->device : -1);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  if (!e || n < 0 || (n > 0 && (!rgb || !depth || !K || !T)) || height <= 0 || width <= 0)
    return RATSDF_ERR_BAD_ARGUMENT;
  for (int i = 0; i < n; ++i)
    if (!rgb[i] || !depth[i] || !finite_frame(K[i], T[i], max_depth)) return RATSDF_ERR_BAD_ARGUMENT;
  if (n == 0) return e->sticky();
  const size_t npix = (size_t)height * width;
  int st = e->ensure_stage(npix);
  if (st != RATSDF_OK) return st;
  const size_t slot_bytes = e->stage_pix * 16;  // the ring's stride (see ratsdf_integrate)
  const uint64_t base = e->stage_no;            // frame i of this call takes slot (base + i) % kStageSlots
  e->stage_no += (uint64_t)n;
  auto slot_of = [&](int i) { return (int)((base + (uint64_t)i) % kStageSlots); };
  auto sem = [&](int i) { return ht && lt && ht[i] && lt[i]; };  // tsdf_module.cc:27-31
  // Uploads run on their own stream, up to kStageSlots frames ahead of the frames that use them, so
  // the PCIe copy of later frames overlaps the integration of earlier ones (one stream would
  // serialise them).  Per device slot: up_ev = its upload has been executed, use_ev = the frame that
  // read it has been executed.  Layout of a slot: depth | ht | lt | rgb.
  static const bool two_streams = !(getenv("RATSDF_COPY_STREAMS") && atoi(getenv("RATSDF_COPY_STREAMS")) == 1);
  // a frame whose four images lie side by side in one page-locked block in the slot's own order (ratsdf::TSDFSystem's
  // queue keeps them so): one copy instead of four -- each costs ~10 us of launch overhead on the copy engine
  auto packed = [&](int i) {
    const uint8_t* h0 = reinterpret_cast<const uint8_t*>(depth[i]);
    return pinned && sem(i) && reinterpret_cast<const uint8_t*>(ht[i]) == h0 + npix * 4 &&
           reinterpret_cast<const uint8_t*>(lt[i]) == h0 + npix * 8 &&
           reinterpret_cast<const uint8_t*>(rgb[i]) == h0 + npix * 12;
  };
  unsigned copy_no = 0;
  // uploads frames [i, i + *took), *took <= max_run: more than one when the frames are packed blocks that lie side
  // by side in host memory at the ring's own stride (the blocks of one arena of ratsdf::HostBlockPool) and their
  // slots do not wrap -- then ONE copy fills the slots (16 bytes per pixel: the 15 the frame has + the slot's pad)
  auto upload = [&](int i, int max_run, int* took) -> int {
    const int slot = slot_of(i);
    int run = 1;
    if (packed(i) && npix == e->stage_pix)
      while (run < max_run && i + run < n && slot + run < kStageSlots && packed(i + run) &&
             reinterpret_cast<const uint8_t*>(depth[i + run]) == reinterpret_cast<const uint8_t*>(depth[i]) + (size_t)run * slot_bytes)
        ++run;
    *took = run;
    hipStream_t cs = ((copy_no++ & 1) && two_streams) ? e->copy_stream2 : e->copy_stream;
    uint8_t* d = e->d_stage + (size_t)slot * slot_bytes;
    // (the slot's last reader: a frame of this call or of an earlier one; an event never recorded counts as complete)
    for (int j = 0; j < run; ++j) HIPCHK(hipStreamWaitEvent(cs, e->use_ev[slot + j], 0));
    const uint8_t* h0 = reinterpret_cast<const uint8_t*>(depth[i]);
    if (packed(i)) {
      HIPCHK(hipMemcpyAsync(d, h0, (size_t)(run - 1) * slot_bytes + npix * 15, hipMemcpyHostToDevice, cs));
    } else if (pinned) {  // straight from the caller's page-locked buffers
      HIPCHK(hipMemcpyAsync(d, depth[i], npix * 4, hipMemcpyHostToDevice, cs));
      if (sem(i)) {
        HIPCHK(hipMemcpyAsync(d + npix * 4, ht[i], npix * 4, hipMemcpyHostToDevice, cs));
        HIPCHK(hipMemcpyAsync(d + npix * 8, lt[i], npix * 4, hipMemcpyHostToDevice, cs));
      }
      HIPCHK(hipMemcpyAsync(d + npix * 12, rgb[i], npix * 3, hipMemcpyHostToDevice, cs));
    } else {
      uint8_t* h = e->h_stage + (size_t)slot * slot_bytes;
      HIPCHK(hipEventSynchronize(e->stage_ev[slot]));  // the slot's last upload has left its page-locked memory
      // (the frame's images side by side, by the engine's copy helpers: HostCopyPool)
      HostCopyPool::Piece pieces[4];
      int np = 0;
      pieces[np++] = HostCopyPool::Piece{h, depth[i], npix * 4};
      if (sem(i)) {
        pieces[np++] = HostCopyPool::Piece{h + npix * 4, ht[i], npix * 4};
        pieces[np++] = HostCopyPool::Piece{h + npix * 8, lt[i], npix * 4};
      }
      pieces[np++] = HostCopyPool::Piece{h + npix * 12, rgb[i], npix * 3};
      if (!e->copy_pool) e->copy_pool = new (std::nothrow) HostCopyPool(3);
      if (e->copy_pool) {
        e->copy_pool->copy(pieces, np);
      } else {
        for (int q = 0; q < np; ++q) memcpy(pieces[q].dst, pieces[q].src, pieces[q].bytes);
      }
      if (sem(i)) {
        HIPCHK(hipMemcpyAsync(d, h, npix * 15, hipMemcpyHostToDevice, cs));
      } else {  // depth | (no ht, lt) | rgb
        HIPCHK(hipMemcpyAsync(d, h, npix * 4, hipMemcpyHostToDevice, cs));
        HIPCHK(hipMemcpyAsync(d + npix * 12, h + npix * 12, npix * 3, hipMemcpyHostToDevice, cs));
      }
    }
    for (int j = 0; j < run; ++j) HIPCHK(hipEventRecord(e->stage_ev[slot + j], cs));
    return RATSDF_OK;
  };
  auto input = [&](int i) {
    uint8_t* d = e->d_stage + (size_t)slot_of(i) * slot_bytes;
    return ratsdf_engine::FrameIn{d + npix * 12, d, sem(i) ? d + npix * 4 : nullptr,
                                  sem(i) ? d + npix * 8 : nullptr, &K[i], &T[i]};
  };
  // whatever happens, the caller's buffers (and the staging slots) are no longer in use on return
  auto fail = [&](int status) {
    (void)hipStreamSynchronize(e->copy_stream);
    (void)hipStreamSynchronize(e->copy_stream2);
    e->abandon_pipeline();
    return status;
  };
  // (No fence against earlier calls: frames of earlier host-image calls that are still in flight are what the
  // slots' events stand for, and nothing else on the engine's stream touches the staging slots.  Until round 5
  // every call began by making both copy streams wait for ALL earlier work of the engine's stream and ended with
  // a synchronisation: the link idled ~0.4 ms per 32-frame call, 13 % of it -- tools/copy_gaps.py.)
  // Uploads are enqueued up to `ahead` frames in front of the frame being launched: slot (u % kStageSlots) was last
  // read by frame u - kStageSlots, whose use_ev must have been RECORDED (i.e. that frame launched) before a copy
  // stream is told to wait for it.
  const int ahead = kStageSlots - 1;
  int uploaded = 0;
  for (int i = 0; i < n; ++i) {
    // frame i's launches host the look-ahead of frame i+1: both uploads precede them
    while (uploaded < n && uploaded <= i + 1) {
      int took = 0;
      st = upload(uploaded, std::min(kUploadRun, i + ahead - uploaded + 1), &took);
      if (st != RATSDF_OK) return fail(st);
      uploaded += took;
    }
    if (hipStreamWaitEvent(e->stream, e->stage_ev[slot_of(i)], 0) != hipSuccess ||
        (i + 1 < n && hipStreamWaitEvent(e->stream, e->stage_ev[slot_of(i + 1)], 0) != hipSuccess))
      return fail(RATSDF_ERR_DEVICE);
    const ratsdf_engine::FrameIn cur = input(i);
    ratsdf_engine::FrameIn nxt{};
    if (i + 1 < n) nxt = input(i + 1);
    st = e->frame(cur, i + 1 < n ? &nxt : nullptr, height, width, max_depth);
    if (st != RATSDF_OK) return fail(st);
    // frame i's images were last read by its candidate pass, which ran in frame i-1's launches or
    // before; recording after frame i is the simple, safe point
    if (hipEventRecord(e->use_ev[slot_of(i)], e->stream) != hipSuccess) return fail(RATSDF_ERR_DEVICE);
    // run further ahead with the uploads while the queue is busy -- in whole runs (or the batch's tail), so that
    // side-by-side frames keep going up together instead of one by one as slots fall free
    while (uploaded < n) {
      const int want = std::min(kUploadRun, n - uploaded);
      if (uploaded + want - 1 > i + ahead) break;
      int took = 0;
      st = upload(uploaded, want, &took);
      if (st != RATSDF_OK) return fail(st);
      uploaded += took;
    }
  }
  // The call returns when the caller's buffers are no longer in use: at once for pageable images (they were copied
  // into the staging ring), after the last upload for page-locked ones.  It does not wait for the frames'
  // kernels -- the next call's uploads overlap them -- so a device error of these frames is reported by the next
  // entry point that synchronises, as for ratsdf_integrate.
  if (pinned) {
    if (hipStreamSynchronize(e->copy_stream) != hipSuccess || hipStreamSynchronize(e->copy_stream2) != hipSuccess)
      return fail(RATSDF_ERR_DEVICE);
  }
  if (!e->sync_integrate) return RATSDF_OK;
  st = e->sticky();
  if (st == RATSDF_ERR_DEVICE) return fail(st);
  return st;
}

int ratsdf_host_alloc(size_t bytes, void** out) {
  if (!out) return RATSDF_ERR_BAD_ARGUMENT;
  *out = nullptr;
  if (bytes == 0) return RATSDF_OK;
  HIPCHK(hipHostMalloc(out, bytes, hipHostMallocDefault));
  return RATSDF_OK;
}

int ratsdf_host_free(void* p) {
  if (p) HIPCHK(hipHostFree(p));
  return RATSDF_OK;
}

int ratsdf_synchronize(ratsdf_engine* e) {
  DeviceGuard guard(e ? e->device : -1);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  if (!e) return RATSDF_ERR_BAD_ARGUMENT;
  // (no k_settle here: what the last frame's carve pass still owes -- pool releases, its statistics -- is done by the
  // next frame's launches, or by the entry point that reads the map, the free list or the statistics: each of them
  // settles first.  A caller that synchronises after every frame, TSDFGrid::Integrate's convention, paid a fourth
  // launch per frame for it.)
  return e->sticky();
}

static int mask_positions(ratsdf_engine* e, const uint32_t* mask, size_t n, uint32_t* pos, uint32_t* scratch_tiles,
                          uint32_t* d_total, uint32_t* h_total);

// After a sticky error (RATSDF_ERR_TIMEOUT above all: a workgroup gave up waiting and skipped its share of a frame) the
// map is what the frames before left plus a part of the failed one, and the structures DERIVED from the directory no
// longer agree with it: pool indices reserved by a serial role whose commits were skipped, claims never reset, counters
// of a frame that nobody finalised.  The directory itself is whole -- the workgroups that EDIT it (the resolvers, the
// serial role) are the ones that were waited for, they never give up and have finished by the time the stream is idle.
// So: everything is rebuilt from the directory -- occupancy bits, Table::active, the free list (the pool blocks no
// entry names, ascending), the free count and its low-water mark -- the claim tables, both frames' counters, the
// candidate counters and the delete bitmaps go back to their initial state, a consumer of directory deltas is told to
// take a whole directory next, and the error is cleared.  The voxels keep what reached them.  No reference counterpart.
int ratsdf_recover(ratsdf_engine* e) {
  DeviceGuard guard(e ? e->device : -1);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  if (!e) return RATSDF_ERR_BAD_ARGUMENT;
  HIPCHK(hipStreamSynchronize(e->stream));
  if (e->copy_stream) HIPCHK(hipStreamSynchronize(e->copy_stream));
  if (e->copy_stream2) HIPCHK(hipStreamSynchronize(e->copy_stream2));
  Table& t = e->tab;
  const uint32_t occ_words = (t.num_entry + 63) / 64;
  const size_t nb = (size_t)t.num_block;
  const uint32_t ntiles = (uint32_t)((nb + kScanTile - 1) / kScanTile);
  uint32_t* tmp = nullptr;  // unused flags | positions | tile sums | total
  HIPCHK(hipMalloc(&tmp, (2 * nb + ntiles + 2) * 4));
  uint32_t *unused = tmp, *pos = tmp + nb, *tiles = pos + nb, *d_total = tiles + ntiles + 1;
  auto fail = [&](int st) {
    (void)hipStreamSynchronize(e->stream);
    (void)hipFree(tmp);
    return st;
  };
#define REC_CHK(expr) do { if ((expr) != hipSuccess) return fail(RATSDF_ERR_DEVICE); } while (0)
  REC_CHK(hipMemsetAsync(t.occ, 0, (size_t)occ_words * 8, e->stream));
  REC_CHK(hipMemsetAsync(t.active, 0xFF, nb * sizeof(VisItem), e->stream));
  REC_CHK(hipMemsetAsync(t.claim, 0xFF, (size_t)t.num_bucket * 4, e->stream));
  REC_CHK(hipMemsetAsync(t.dclaim, 0xFF, (size_t)t.num_bucket * 4, e->stream));
  REC_CHK(hipMemsetAsync(e->dbitmap, 0, (size_t)e->dwords * 4, e->stream));
  REC_CHK(hipMemsetAsync(e->dsummary, 0, (size_t)((e->dwords / kGroupWords + 31) / 32) * 4, e->stream));
  if (e->abitmap) {
    REC_CHK(hipMemsetAsync(e->abitmap, 0, (size_t)e->awords_cap * 4, e->stream));
    REC_CHK(hipMemsetAsync(e->asummary, 0, (size_t)e->asum_words * 4, e->stream));
  }
  REC_CHK(hipMemsetAsync(e->cand_count, 0, 2 * kCandSegs * kCandCountStride * 4, e->stream));
  for (int i = 0; i < 2; ++i) REC_CHK(hipMemsetAsync(e->upd_wg[i], 0, kUpdCounters * 4, e->stream));
  REC_CHK(hipMemsetAsync(&e->ctl->fr[0], 0, 2 * sizeof(FrameCtl), e->stream));
  hipLaunchKernelGGL(k_fill_u32, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, e->stream, unused, 1u, nb);
  hipLaunchKernelGGL(k_recover_scan, dim3((t.num_entry + 255) / 256), dim3(256), 0, e->stream, t, unused);
  uint32_t n_free = 0;
  if (mask_positions(e, unused, nb, pos, tiles, d_total, &n_free) != RATSDF_OK) return fail(RATSDF_ERR_DEVICE);
  hipLaunchKernelGGL(k_recover_heap, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, e->stream, unused, pos,
                     e->pool.heap, (int32_t)nb);
  const int32_t nf = (int32_t)n_free;
  REC_CHK(hipMemcpyAsync(&e->ctl->num_free, &nf, 4, hipMemcpyHostToDevice, e->stream));
  // (the low-water mark only ever goes down: slots at or above it may have been in use; the rebuilt heap keeps the
  // never-used indices -- the lowest ones -- at its bottom, so the mark stays true.  A free count below it moves it.)
  int32_t low = 0;
  REC_CHK(hipMemcpyAsync(&low, &e->ctl->free_low, 4, hipMemcpyDeviceToHost, e->stream));
  REC_CHK(hipStreamSynchronize(e->stream));
  if (nf < low) REC_CHK(hipMemcpyAsync(&e->ctl->free_low, &nf, 4, hipMemcpyHostToDevice, e->stream));
  if (t.delta_on) {  // the delta log no longer describes what changed: the next export reports an overflow
    const uint32_t over = 0x80000000u;
    REC_CHK(hipMemcpyAsync(t.del_count, &over, 4, hipMemcpyHostToDevice, e->stream));
  }
  const uint32_t zero = 0;
  REC_CHK(hipMemcpyAsync(&e->ctl->error, &zero, 4, hipMemcpyHostToDevice, e->stream));
  REC_CHK(hipGetLastError());
  REC_CHK(hipStreamSynchronize(e->stream));
#undef REC_CHK
  e->h_err[0] = e->h_err[1] = 0u;
  e->pending = false;
  e->cand_ready = false;
  (void)hipFree(tmp);
  return RATSDF_OK;
}

int ratsdf_stream(ratsdf_engine* e, void** out) {
  DeviceGuard guard(e ? e->device : -1);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  if (!e || !out) return RATSDF_ERR_BAD_ARGUMENT;
  *out = (void*)e->stream;
  return RATSDF_OK;
}

int ratsdf_profile_enable(ratsdf_engine* e, int enable) {
  DeviceGuard guard(e ? e->device : -1);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  if (!e) return RATSDF_ERR_BAD_ARGUMENT;
  const int st = e->drain_profile();
  e->profiling = enable != 0;
  e->prof_mode = enable == 2 ? 2 : 1;
  e->prof_k_us.clear();
  e->prof_period_us.clear();
  return st;
}

int ratsdf_profile_read_frames(ratsdf_engine* e, float* k_us, float* period_us, int capacity, int* n) {
  DeviceGuard guard(e ? e->device : -1);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  if (!e || !n || capacity < 0) return RATSDF_ERR_BAD_ARGUMENT;
  const int st = e->drain_profile();
  const int have = (int)e->prof_k_us.size();
  *n = have;
  for (int i = 0; i < have && i < capacity; ++i) {
    if (k_us) k_us[i] = e->prof_k_us[(size_t)i];
    if (period_us) period_us[i] = (size_t)i < e->prof_period_us.size() ? e->prof_period_us[(size_t)i] : 0.f;
  }
  e->prof_k_us.clear();
  e->prof_period_us.clear();
  e->prof_ms = 0;
  e->prof_n = 0;
  return st;
}

int ratsdf_profile_read(ratsdf_engine* e, double* ms, int64_t* launches) {
  DeviceGuard guard(e ? e->device : -1);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  if (!e) return RATSDF_ERR_BAD_ARGUMENT;
  const int st = e->drain_profile();
  if (ms) *ms = e->prof_ms;
  if (launches) *launches = e->prof_n;
  e->prof_ms = 0;
  e->prof_n = 0;
  return st;
}

int ratsdf_num_active_blocks(ratsdf_engine* e, int32_t* out) {
  DeviceGuard guard(e ? e->device : -1);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  if (!e || !out) return RATSDF_ERR_BAD_ARGUMENT;
  { const int st0 = e->settle(); if (st0 != RATSDF_OK) return st0; }
  int32_t nf = 0;
  { const int rs = e->read_small(&nf, &e->ctl->num_free, 4); if (rs != RATSDF_OK) return rs; }
  *out = e->tab.num_block - nf;
  return RATSDF_OK;
}

int ratsdf_last_frame_stats(ratsdf_engine* e, ratsdf_frame_stats* out) {
  DeviceGuard guard(e ? e->device : -1);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  if (!e || !out) return RATSDF_ERR_BAD_ARGUMENT;
  { const int st0 = e->settle(); if (st0 != RATSDF_OK) return st0; }
  { const int rs = e->read_small(out, e->d_stats, sizeof(*out)); if (rs != RATSDF_OK) return rs; }
  return RATSDF_OK;
}

// diagnostic: per-wave stamps of the LAST k_integrate launch (stamps build only)
extern "C" int ratsdf_debug_wave_stamps(ratsdf_engine* e, int enable) {
  DeviceGuard guard(e ? e->device : -1);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  static unsigned long long* buf = nullptr;
  const size_t n = 16384 * 8;
  if (enable > 0) {
    if (!buf) HIPCHK(hipMalloc(&buf, n * 8));
    HIPCHK(hipMemsetAsync(buf, 0, n * 8, e->stream));
    HIPCHK(hipMemcpyAsync(&e->ctl->debug_buf, &buf, sizeof(buf), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return RATSDF_OK;
  }
  std::vector<unsigned long long> h(n);
  HIPCHK(hipMemcpyAsync(h.data(), buf, n * 8, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  // k_integrate keeps one half of the buffer per frame parity (the last two frames of a batch stay apart):
  // enable = -1 / -2 reports the half of parity 0 / 1 alone
  if (enable < 0) {
    const size_t keep = (size_t)(-enable - 1);
    for (size_t w = 0; w < 16384; ++w)
      if ((w >> 13) != keep)
        for (int k = 0; k < 8; ++k) h[w * 8 + k] = 0;
  }
  unsigned long long t0 = ~0ull, t1 = 0;
  double ph[4] = {0, 0, 0, 0};
  size_t cnt = 0;
  std::vector<unsigned long long> starts, ends;
  for (size_t w = 0; w < 16384; ++w) {
    const unsigned long long* s = &h[w * 8];
    if (!s[0] || !s[4] || !s[1]) continue;
    t0 = s[5] < t0 ? s[5] : t0;
    t1 = s[6] > t1 ? s[6] : t1;
    ph[0] += (double)(s[1] - s[0]);
    ph[1] += (double)(s[2] - s[1]);
    ph[2] += (double)(s[3] - s[2]);
    ph[3] += (double)(s[4] - s[3]);
    starts.push_back(s[5]);
    ends.push_back(s[6]);
    ++cnt;
  }
  if (!cnt) { fprintf(stderr, "[wave stamps] none\n"); return RATSDF_OK; }
  {  // phase profile of the slowest 5 % of the waves
    std::vector<std::pair<unsigned long long, size_t>> dur;
    for (size_t w = 0; w < 16384; ++w) {
      const unsigned long long* s = &h[w * 8];
      if (!s[0] || !s[4] || !s[1]) continue;
      dur.emplace_back(s[4] - s[0], w);
    }
    std::sort(dur.begin(), dur.end());
    const size_t lo = dur.size() * 95 / 100;
    double q[4] = {0, 0, 0, 0};
    for (size_t i = lo; i < dur.size(); ++i) {
      const unsigned long long* s = &h[dur[i].second * 8];
      q[0] += (double)(s[1] - s[0]); q[1] += (double)(s[2] - s[1]);
      q[2] += (double)(s[3] - s[2]); q[3] += (double)(s[4] - s[3]);
    }
    const double m = (double)(dur.size() - lo);
    fprintf(stderr, "[wave stamps] wave duration cycles: p50 %llu p95 %llu max %llu; slowest 5%% phases: %.0f | %.0f | %.0f | %.0f\n",
            dur[dur.size() / 2].first, dur[lo].first, dur.back().first, q[0] / m, q[1] / m, q[2] / m, q[3] / m);
  }
  std::sort(starts.begin(), starts.end());
  std::sort(ends.begin(), ends.end());
  fprintf(stderr, "[wave stamps] %zu waves (first block of each); span first-start..last-end = %llu ticks of 10 ns\n", cnt, t1 - t0);
  fprintf(stderr, "[wave stamps] mean cycles per phase: %.0f | %.0f | %.0f | %.0f  (k_integrate: issue+project | wait loads | math | store; k_front pixels with RATSDF_DEBUG=8: load+texel | ray math | wait lookups | evaluate)\n",
          ph[0] / cnt, ph[1] / cnt, ph[2] / cnt, ph[3] / cnt);
  fprintf(stderr, "[wave stamps] start spread: p50 %llu p99 %llu max %llu ; end: p1 %llu p50 %llu (relative to first start)\n",
          starts[cnt / 2] - t0, starts[cnt * 99 / 100] - t0, starts[cnt - 1] - t0, ends[cnt / 100] - t0, ends[cnt / 2] - t0);
  {  // the whole launch: when the serial role published, when the waves' LAST passes ended
    unsigned long long st[6], last = 0;
    std::vector<unsigned long long> done;
    for (size_t w = 0; w < 16384; ++w)
      if (h[w * 8 + 7]) done.push_back(h[w * 8 + 7]);
    HIPCHK(hipMemcpyAsync(st, e->ctl->stamps, sizeof(st), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (!done.empty()) {
      std::sort(done.begin(), done.end());
      last = done.back();
      for (int par = 0; par < 2; ++par)
        if ((enable == 0 || par == -enable - 1) && st[par * 3 + 1] > t0 && st[par * 3 + 1] < last)
          fprintf(stderr, "[wave stamps] serial role: started %lld, published at %lld; waves' last passes end: p50 %llu p99 %llu max %llu (10 ns ticks after the first update wave started)\n",
                  (long long)(st[par * 3] - t0), (long long)(st[par * 3 + 1] - t0), done[done.size() / 2] - t0,
                  done[done.size() * 99 / 100] - t0, last - t0);
    }
  }
  return RATSDF_OK;
}

#ifdef RATSDF_STAMPS
// diagnostic (stamps build only): the ablation / fault-injection switch of an engine after its creation (RATSDF_DEBUG
// sets it at creation): tests/test_gpu_errors.py injects a fault, switches it off and recovers
extern "C" int ratsdf_debug_set_switch(ratsdf_engine* e, int value) {
  if (!e) return RATSDF_ERR_BAD_ARGUMENT;
  e->debug = value;
  return RATSDF_OK;
}
#endif

// diagnostic (stamps build only): the raw per-wave record buffer ratsdf_debug_wave_stamps(e, 1) attached (16 384 x 8
// words), copied out and zeroed -- k_raycast's per-wave timeline (tools/raycast_probe.py)
extern "C" int ratsdf_debug_wave_records(ratsdf_engine* e, unsigned long long* out, size_t words) {
  DeviceGuard guard(e ? e->device : -1);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  if (!e || !out || words > 16384 * 8) return RATSDF_ERR_BAD_ARGUMENT;
  unsigned long long* buf = nullptr;
  HIPCHK(hipMemcpyAsync(&buf, &e->ctl->debug_buf, sizeof(buf), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  if (!buf) return RATSDF_ERR_BAD_ARGUMENT;
  HIPCHK(hipMemcpyAsync(out, buf, words * 8, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return RATSDF_OK;
}

// diagnostic (stamps build only): Ctl::dbg -- RATSDF_DEBUG=30 counts update waves that changed no voxel:
// [0] such waves, [1] waves, [2] blocks without an update, [3] blocks; read and reset
extern "C" int ratsdf_debug_counters(ratsdf_engine* e, unsigned long long* out8) {
  DeviceGuard guard(e ? e->device : -1);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  if (!e || !out8) return RATSDF_ERR_BAD_ARGUMENT;
  HIPCHK(hipMemcpyAsync(out8, e->ctl->dbg, 8 * 8, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipMemsetAsync(e->ctl->dbg, 0, 8 * 8, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return RATSDF_OK;
}

// diagnostic: timeline of k_front's tail (stamps build only): sums over frames of wall-clock ticks (10 ns)
// since the launch's first workgroup started
extern "C" int ratsdf_debug_tail_stamps(ratsdf_engine* e) {
  DeviceGuard guard(e ? e->device : -1);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  unsigned long long t[16];
  HIPCHK(hipMemcpyAsync(t, e->ctl->tstamps, sizeof(t), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipMemsetAsync(e->ctl->tstamps, 0, sizeof(t), e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  const double n = t[8] ? (double)t[8] : 1.0;
  fprintf(stderr, "[tail stamps] %llu tail frames; us after the launch's first workgroup started: last directory workgroup "
          "done %.2f | its stores drained %.2f | it knows it is last %.2f | tail: first round of loads in %.2f | claims + "
          "winners listed %.2f | commits issued %.2f | end %.2f ; requests %.1f winners %.1f per frame\n",
          t[8], t[1] / n / 100, t[2] / n / 100, t[3] / n / 100, t[4] / n / 100, t[5] / n / 100, t[6] / n / 100,
          t[7] / n / 100, t[9] / n, t[10] / n);
  if (t[15]) {
    const double m = (double)t[15];
    unsigned long long c[32];
    HIPCHK(hipMemcpyAsync(c, e->ctl->stamps, sizeof(c), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    const double k = c[17] ? (double)c[17] : 1.0;
    fprintf(stderr, "[cand stamps] %llu candidate workgroups sampled (wave 0), shader cycles: inputs arrive %.0f | ray set-up %.0f | "
            "sample loop %.0f (%.2f iterations) ; workgroup: set init + barrier %.0f | pixel work %.0f | barrier wait %.0f | "
            "compaction + stores %.0f\n",
            t[15], t[11] / m, t[12] / m, t[13] / m, t[14] / m, c[14] / k, c[15] / k, c[16] / k, c[18] / k);
  }
  return RATSDF_OK;
}

// diagnostic: prints the accumulated phase stamps of the single-workgroup kernels (stamps build only)
extern "C" int ratsdf_debug_stamps(ratsdf_engine* e) {
  DeviceGuard guard(e ? e->device : -1);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  unsigned long long t[32];
  unsigned long long tot[5];
  HIPCHK(hipMemcpyAsync(t, e->ctl->stamps, sizeof(t), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipMemcpyAsync(tot, e->ctl->totals, sizeof(tot), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  const double n = tot[0] ? (double)tot[0] : 1.0;
  fprintf(stderr, "[stamps] frames=%llu  serial role (shader cycles/frame): loads:%.0f claims+barrier:%.0f lists:%.0f ranks:%.0f tail:%.0f | deletes %.1f winners %.1f requests %.1f per frame\n",
          tot[0], (double)(t[9] - t[8]) / n, (double)(t[10] - t[9]) / n, (double)(t[11] - t[10]) / n,
          (double)(t[12] - t[11]) / n, (double)(t[13] - t[12]) / n, (double)t[16] / n, (double)t[17] / n,
          (double)t[18] / n);
  if (t[29])
    fprintf(stderr, "[stamps] chained-bucket resolver, %llu passes (shader cycles/pass): order + duplicates %.0f | plans %.0f | replay %.0f | apply %.0f | per pass: requests %.1f distinct %.1f stale plans %.2f placed %.1f | step loop %.0f cycles for %.1f steps\n",
            t[29], (double)(t[22] - t[20]) / t[29], (double)(t[23] - t[22]) / t[29], (double)(t[24] - t[23]) / t[29],
            (double)(t[21] - t[24]) / t[29], (double)t[25] / t[29], (double)t[26] / t[29], (double)t[27] / t[29],
            (double)t[28] / t[29], (double)(long long)t[30] / t[29], (double)t[31] / t[29]);
  fprintf(stderr, "[stamps] ranks phase, first pass (cold code) %.0f cycles of the two\n", (double)t[19] / n);
  {
    const double m = t[17] ? (double)t[17] : 1.0;
    fprintf(stderr, "[stamps] candidate pass, thread 0 of sampled workgroups (shader cycles): first barrier %.0f | pixel work %.0f | wait for the workgroup %.0f | compaction + stores %.0f\n",
            (double)t[14] / m, (double)t[15] / m, (double)t[16] / m, (double)t[18] / m);
  }
  return RATSDF_OK;
}

int ratsdf_totals(ratsdf_engine* e, int64_t* out5, int reset) {
  DeviceGuard guard(e ? e->device : -1);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  if (!e) return RATSDF_ERR_BAD_ARGUMENT;
  { const int st0 = e->settle(); if (st0 != RATSDF_OK) return st0; }
  unsigned long long t[5] = {0, 0, 0, 0, 0};
  { const int rs = e->read_small(t, e->ctl->totals, sizeof(t)); if (rs != RATSDF_OK) return rs; }
  if (reset) HIPCHK(hipMemsetAsync(e->ctl->totals, 0, sizeof(t), e->stream));
  if (out5)
    for (int i = 0; i < 5; ++i) out5[i] = (int64_t)t[i];
  return RATSDF_OK;
}

int ratsdf_pipeline_counters(ratsdf_engine* e, int64_t* out4, int reset) {
  DeviceGuard guard(e ? e->device : -1);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  if (!e) return RATSDF_ERR_BAD_ARGUMENT;
  unsigned long long t[4] = {0, 0, 0, 0};
  { const int rs = e->read_small(t, e->ctl->paths, sizeof(t)); if (rs != RATSDF_OK) return rs; }
  if (reset) HIPCHK(hipMemsetAsync(e->ctl->paths, 0, sizeof(t), e->stream));
  if (out4)
    for (int i = 0; i < 4; ++i) out4[i] = (int64_t)t[i];
  return RATSDF_OK;
}

// Device and page-locked staging buffers of the query-side downloads: kept between calls and only
// ever grown (a hipMalloc / hipFree pair and a pageable D2H copy per Query cost more than the kernels).
static int ensure_download_buffers(ratsdf_engine* e, size_t bytes) {
  if (bytes <= e->dl_cap) return RATSDF_OK;
  HIPCHK(hipStreamSynchronize(e->stream));
  if (e->dl_dev) (void)hipFree(e->dl_dev);
  if (e->dl_host) (void)hipHostFree(e->dl_host);
  e->dl_dev = nullptr;
  e->dl_host = nullptr;
  e->dl_cap = 0;
  const size_t cap = bytes + bytes / 4;
  HIPCHK(hipMalloc(&e->dl_dev, cap));
  HIPCHK(hipHostMalloc(&e->dl_host, cap, hipHostMallocDefault));
  e->dl_cap = cap;
  return RATSDF_OK;
}

static int download_selected(ratsdf_engine* e, bool semantic, void** out, size_t* n) {
  uint32_t cnt = 0;
  { const int rs = e->read_small(&cnt, &e->ctl->n_sel, 4); if (rs != RATSDF_OK) return rs; }
  const size_t rec = semantic ? sizeof(ratsdf_voxel_segm) : sizeof(ratsdf_voxel_tsdf);
  const size_t total = (size_t)cnt * RATSDF_BLOCK_VOLUME;
  void* host = malloc(total ? total * rec : 1);
  if (!host) return RATSDF_ERR_DEVICE;
  if (total) {
    const int st = ensure_download_buffers(e, total * rec);
    if (st != RATSDF_OK) {
      free(host);
      return st;
    }
    float* dev = static_cast<float*>(e->dl_dev);
    const unsigned grid = cnt < 4096u ? (cnt + 3) / 4 : 1024u;
    if (semantic)
      hipLaunchKernelGGL(k_download<true>, dim3(grid), dim3(256), 0, e->stream, e->pool, e->vis,
                         &e->ctl->n_sel, e->vs, dev);
    else
      hipLaunchKernelGGL(k_download<false>, dim3(grid), dim3(256), 0, e->stream, e->pool, e->vis,
                         &e->ctl->n_sel, e->vs, dev);
    hipError_t err = hipMemcpyAsync(e->dl_host, dev, total * rec, hipMemcpyDeviceToHost, e->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
    if (err != hipSuccess) {
      free(host);
      return RATSDF_ERR_DEVICE;
    }
    // the caller owns `host` (ratsdf_free_buffer).  A large result is copied out by the engine's helper threads side
    // by side: the destination is fresh memory, and first-touch page faults (10 k of them for the 41 MB of a
    // GatherValid on the bench map) are what the single-threaded copy spent most of its time on
    const size_t bytes = total * rec;
    if (bytes >= ((size_t)4 << 20)) {
      if (!e->copy_pool) e->copy_pool = new (std::nothrow) HostCopyPool(3);
    }
    if (bytes >= ((size_t)4 << 20) && e->copy_pool) {
      HostCopyPool::Piece pieces[16];
      int np = 0;
      const size_t step = ((bytes + 15) / 16 + 4095) & ~(size_t)4095;
      for (size_t o = 0; o < bytes; o += step)
        pieces[np++] = HostCopyPool::Piece{(uint8_t*)host + o, (const uint8_t*)e->dl_host + o, std::min(step, bytes - o)};
      e->copy_pool->copy(pieces, np);
    } else {
      memcpy(host, e->dl_host, bytes);
    }
  }
  *out = host;
  *n = total;
  return RATSDF_OK;
}

static inline int16_t host_f2s(float f) {  // static_cast<short>, BoundingCube::Scale
  if (f != f) return 0;
  if (f >= 2147483648.f) return (int16_t)2147483647;
  if (f <= -2147483648.f) return (int16_t)(-2147483647 - 1);
  return (int16_t)(int)f;
}

int ratsdf_query(ratsdf_engine* e, const ratsdf_bounds* b, ratsdf_voxel_tsdf** out, size_t* n) {
  DeviceGuard guard(e ? e->device : -1);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  if (!e || !b || !out || !n) return RATSDF_ERR_BAD_ARGUMENT;
  { const int st0 = e->settle(); if (st0 != RATSDF_OK) return st0; }
  const float scale = (float)(1. / e->vs);  // volumn.Scale<short>(1. / voxel_size_), voxel_tsdf.cu:534
  GridBounds gb{host_f2s(b->xmin * scale), host_f2s(b->xmax * scale), host_f2s(b->ymin * scale),
                host_f2s(b->ymax * scale), host_f2s(b->zmin * scale), host_f2s(b->zmax * scale)};
  int st = e->select(kSelBounds, gb, &e->ctl->n_sel);
  if (st != RATSDF_OK) return st;
  return download_selected(e, false, (void**)out, n);
}

int ratsdf_gather_valid(ratsdf_engine* e, ratsdf_voxel_tsdf** out, size_t* n) {
  DeviceGuard guard(e ? e->device : -1);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  if (!e || !out || !n) return RATSDF_ERR_BAD_ARGUMENT;
  { const int st0 = e->settle(); if (st0 != RATSDF_OK) return st0; }
  int st = e->select(kSelValid, GridBounds{}, &e->ctl->n_sel);
  if (st != RATSDF_OK) return st;
  return download_selected(e, false, (void**)out, n);
}

int ratsdf_gather_valid_semantic(ratsdf_engine* e, ratsdf_voxel_segm** out, size_t* n) {
  DeviceGuard guard(e ? e->device : -1);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  if (!e || !out || !n) return RATSDF_ERR_BAD_ARGUMENT;
  { const int st0 = e->settle(); if (st0 != RATSDF_OK) return st0; }
  int st = e->select(kSelValid, GridBounds{}, &e->ctl->n_sel);
  if (st != RATSDF_OK) return st;
  return download_selected(e, true, (void**)out, n);
}

int ratsdf_download_all(ratsdf_engine* e, const char* path) {
  DeviceGuard guard(e ? e->device : -1);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  if (!e || !path) return RATSDF_ERR_BAD_ARGUMENT;
  { const int st0 = e->settle(); if (st0 != RATSDF_OK) return st0; }
  ratsdf_voxel_segm* buf = nullptr;
  size_t n = 0;
  const int st = ratsdf_gather_valid_semantic(e, &buf, &n);
  if (st != RATSDF_OK) return st;
  FILE* f = fopen(path, "wb");
  if (!f) {
    free(buf);
    return RATSDF_ERR_BAD_ARGUMENT;
  }
  fwrite(buf, sizeof(ratsdf_voxel_segm), n, f);
  fclose(f);
  free(buf);
  return RATSDF_OK;
}

int ratsdf_free_buffer(void* p) {
  free(p);
  return RATSDF_OK;
}

// rows [row0, row1) of the height x width rendering into device buffers that hold those rows
static int raycast_rows_device(ratsdf_engine* e, const ratsdf_intrinsics* K, int height, int width,
                               const ratsdf_pose* T, float max_depth, int row0, int row1, void* d_rgba,
                               void* d_normal) {
  if (!e || !K || !T || height <= 0 || width <= 0 || !(max_depth > 0) || row0 < 0 || row1 > height || row0 > row1)
    return RATSDF_ERR_BAD_ARGUMENT;
  { const int st0 = e->settle(); if (st0 != RATSDF_OK) return st0; }
  if (row0 == row1) return RATSDF_OK;
  FrameParams P = e->base_params();
  P.T = Se3{Quat{T->qx, T->qy, T->qz, T->qw}, V3{T->tx, T->ty, T->tz}};
  P.Ti = se3_inverse(P.T);                      // voxel_tsdf.cu:892 cam_T_world.Inverse()
  P.K = Intr{K->fx, K->fy, K->cx, K->cy};
  P.Ki = intr_inverse(P.K);
  P.W = width;
  P.H = height;
  const float step_size = e->trunc / 2;         // voxel_tsdf.cu:892
  const float ms = ceilf(max_depth / step_size);
  const int max_step = ms >= 2147483648.f ? 2147483647 : (int)ms;  // voxel_tsdf.cu:298
  // block-level occupancy of the map as it is now (kernels_raycast.h: empty space costs no directory probes)
  if (!e->d_occ) HIPCHK(hipMalloc(&e->d_occ, (kOccWords + kCellWords) * 4));
  HIPCHK(hipMemsetAsync(e->d_occ, 0, (kOccWords + kCellWords) * 4, e->stream));
  hipLaunchKernelGGL(k_occupancy_build, dim3(256), dim3(256), 0, e->stream, e->tab, (const Ctl*)e->ctl, e->d_occ);
  hipLaunchKernelGGL(k_raycast, dim3((width + 15) / 16, (row1 - row0 + 15) / 16), dim3(256), 0, e->stream,
                     e->tab, e->pool, P, step_size, max_step, (uint32_t*)d_rgba, (uint32_t*)d_normal, row0, row1,
                     (const uint32_t*)e->d_occ, e->ctl);
  HIPCHK(hipGetLastError());
  return RATSDF_OK;
}

int ratsdf_raycast_device(ratsdf_engine* e, const ratsdf_intrinsics* K, int height, int width,
                          const ratsdf_pose* T, float max_depth, void* d_rgba, void* d_normal) {
  DeviceGuard guard(e ? e->device : -1);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  return raycast_rows_device(e, K, height, width, T, max_depth, 0, height, d_rgba, d_normal);
}

int ratsdf_raycast_rows(ratsdf_engine* e, const ratsdf_intrinsics* K, int height, int width,
                        const ratsdf_pose* T, float max_depth, int row0, int row1, uint8_t* rgba, uint8_t* normal) {
  DeviceGuard guard(e ? e->device : -1);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  if (!e || height <= 0 || width <= 0 || row0 < 0 || row1 > height || row0 > row1) return RATSDF_ERR_BAD_ARGUMENT;
  const size_t bytes = (size_t)(row1 - row0) * width * 4;
  if (bytes == 0) return raycast_rows_device(e, K, height, width, T, max_depth, row0, row1, nullptr, nullptr);
  // The two images leave through buffers the engine keeps: device memory for the kernel's output and page-locked host
  // memory for the copy out (until round 5: a hipMalloc / hipFree pair per call and two copies into the caller's
  // pageable buffers through the runtime's staging path -- 0.84 ms per 640x480 rendering of which the kernel was half).
  if (e->render_cap < bytes * 2) {
    (void)hipStreamSynchronize(e->stream);
    if (e->d_render) (void)hipFree(e->d_render);
    if (e->h_render) (void)hipHostFree(e->h_render);
    e->d_render = e->h_render = nullptr;
    e->render_cap = 0;
    HIPCHK(hipMalloc(&e->d_render, bytes * 2));
    HIPCHK(hipHostMalloc(&e->h_render, bytes * 2, hipHostMallocDefault));
    e->render_cap = bytes * 2;
  }
  uint8_t* d = e->d_render;
  int st = raycast_rows_device(e, K, height, width, T, max_depth, row0, row1, d, d + bytes);
  if (st != RATSDF_OK) return st;
  HIPCHK(hipMemcpyAsync(e->h_render, d, bytes * 2, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));  // voxel_tsdf.cu:901
  if (rgba) memcpy(rgba, e->h_render, bytes);
  if (normal) memcpy(normal, e->h_render + bytes, bytes);
  return RATSDF_OK;
}

int ratsdf_raycast(ratsdf_engine* e, const ratsdf_intrinsics* K, int height, int width,
                   const ratsdf_pose* T, float max_depth, uint8_t* rgba, uint8_t* normal) {
  return ratsdf_raycast_rows(e, K, height, width, T, max_depth, 0, height, rgba, normal);
}

// exclusive positions of the set items of a 0/1 mask; returns the number of set items
static int mask_positions(ratsdf_engine* e, const uint32_t* mask, size_t n, uint32_t* pos,
                          uint32_t* scratch_tiles, uint32_t* d_total, uint32_t* h_total) {
  const uint32_t ntiles = (uint32_t)((n + kScanTile - 1) / kScanTile);
  hipLaunchKernelGGL(k_mask_tile_sums, dim3(ntiles), dim3(1024), 0, e->stream, mask, n, scratch_tiles);
  hipLaunchKernelGGL(k_scan_tile_sums, dim3(1), dim3(1024), 0, e->stream, scratch_tiles, ntiles,
                     d_total);
  hipLaunchKernelGGL(k_mask_positions, dim3(ntiles), dim3(1024), 0, e->stream, mask, n, scratch_tiles,
                     pos);
  HIPCHK(hipMemcpyAsync(h_total, d_total, 4, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return RATSDF_OK;
}

int ratsdf_gather_valid_mesh(ratsdf_engine* e, float** vertices, size_t* n_vertices,
                             int32_t** indices, size_t* n_triangles, float** vertex_prob) {
  DeviceGuard guard(e ? e->device : -1);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  if (!e || !vertices || !n_vertices || !indices || !n_triangles || !vertex_prob)
    return RATSDF_ERR_BAD_ARGUMENT;
  { const int st0 = e->settle(); if (st0 != RATSDF_OK) return st0; }
  // check_valid_kernel + GatherBlock; a sharded map meshes the blocks it owns (imported neighbours are read only)
  int st = e->select(e->shard_count > 1 ? kSelOwned : kSelValid, GridBounds{}, &e->ctl->n_sel);
  if (st != RATSDF_OK) return st;
  uint32_t nb = 0;
  HIPCHK(hipMemcpyAsync(&nb, &e->ctl->n_sel, 4, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  *vertices = (float*)malloc(4);
  *vertex_prob = (float*)malloc(4);
  *indices = (int32_t*)malloc(4);
  *n_vertices = 0;
  *n_triangles = 0;
  if (nb == 0) return RATSDF_OK;
  const size_t nvs = (size_t)nb * kVertVolume * 3;  // candidate vertices
  const size_t nts = (size_t)nb * 512 * 5;          // candidate triangles
  if (!e->d_mc) {
    const McTables h = make_mc_tables();
    HIPCHK(hipMalloc(&e->d_mc, sizeof(McTables)));
    HIPCHK(hipMemcpy(e->d_mc, &h, sizeof(McTables), hipMemcpyHostToDevice));
  }
  // one scratch allocation: verts | vprob | vmask | vpos | tids | tmask | tpos | tile sums | total
  const size_t ntile_max = (nts > nvs ? nts : nvs) / kScanTile + 2;
  const size_t bytes = nvs * 12 + nvs * 4 * 3 + nts * 12 + nts * 4 * 2 + ntile_max * 4 + 64;
  uint8_t* d = nullptr;
  HIPCHK(hipMalloc(&d, bytes));
  float* verts = (float*)d;
  float* vprob = verts + nvs * 3;
  uint32_t* vmask = (uint32_t*)(vprob + nvs);
  uint32_t* vpos = vmask + nvs;
  int32_t* tids = (int32_t*)(vpos + nvs);
  uint32_t* tmask = (uint32_t*)(tids + nts * 3);
  uint32_t* tpos = tmask + nts;
  uint32_t* tiles = tpos + nts;
  uint32_t* d_total = tiles + ntile_max;
  hipLaunchKernelGGL(k_marching_cubes, dim3(nb), dim3(512), 0, e->stream, e->tab, e->pool, e->vis,
                     (const McTables*)e->d_mc, e->vs, verts, vprob, vmask, tids, tmask);
  uint32_t nv = 0, nt = 0;
  st = mask_positions(e, vmask, nvs, vpos, tiles, d_total, &nv);
  if (st == RATSDF_OK) st = mask_positions(e, tmask, nts, tpos, tiles, d_total, &nt);
  if (st != RATSDF_OK) {
    (void)hipFree(d);
    return st;
  }
  free(*vertices);
  free(*vertex_prob);
  free(*indices);
  *vertices = (float*)malloc((size_t)nv * 12 + 4);
  *vertex_prob = (float*)malloc((size_t)nv * 4 + 4);
  *indices = (int32_t*)malloc((size_t)nt * 12 + 4);
  uint8_t* o = nullptr;
  hipError_t err = hipMalloc(&o, (size_t)nv * 16 + (size_t)nt * 12 + 64);
  if (err == hipSuccess) {
    float* ov = (float*)o;
    float* op = ov + (size_t)nv * 3;
    int32_t* oi = (int32_t*)(op + nv);
    hipLaunchKernelGGL(k_compact_vertices, dim3(2048), dim3(256), 0, e->stream, verts, vprob, vmask,
                       vpos, nvs, ov, op);
    hipLaunchKernelGGL(k_compact_triangles, dim3(2048), dim3(256), 0, e->stream, tids, tmask, tpos,
                       vpos, nts, oi);
    if (nv) err = hipMemcpyAsync(*vertices, ov, (size_t)nv * 12, hipMemcpyDeviceToHost, e->stream);
    if (err == hipSuccess && nv)
      err = hipMemcpyAsync(*vertex_prob, op, (size_t)nv * 4, hipMemcpyDeviceToHost, e->stream);
    if (err == hipSuccess && nt)
      err = hipMemcpyAsync(*indices, oi, (size_t)nt * 12, hipMemcpyDeviceToHost, e->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
    (void)hipFree(o);
  }
  (void)hipFree(d);
  if (err != hipSuccess) return RATSDF_ERR_DEVICE;
  *n_vertices = nv;
  *n_triangles = nt;
  return RATSDF_OK;
}

int ratsdf_download_all_mesh(ratsdf_engine* e, const char* vp, const char* ip, const char* pp) {
  DeviceGuard guard(e ? e->device : -1);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  if (!e || !vp || !ip || !pp) return RATSDF_ERR_BAD_ARGUMENT;
  float *v = nullptr, *pr = nullptr;
  int32_t* idx = nullptr;
  size_t nv = 0, nt = 0;
  int st = ratsdf_gather_valid_mesh(e, &v, &nv, &idx, &nt, &pr);
  if (st == RATSDF_OK) {  // modules/tsdf_module.cc:66-86
    FILE* fv = fopen(vp, "wb");
    FILE* fp = fopen(pp, "wb");
    FILE* fi = fopen(ip, "wb");
    if (fv && fp && fi) {
      fwrite(v, 12, nv, fv);
      fwrite(pr, 4, nv, fp);
      fwrite(idx, 12, nt, fi);
    } else {
      st = RATSDF_ERR_BAD_ARGUMENT;
    }
    if (fv) fclose(fv);
    if (fp) fclose(fp);
    if (fi) fclose(fi);
  }
  free(v);
  free(pr);
  free(idx);
  return st;
}

int ratsdf_export_directory_device(ratsdf_engine* e, void* d_blocks, int32_t capacity,
                                   void* d_count) {
  DeviceGuard guard(e ? e->device : -1);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  if (!e || !d_blocks || capacity < 0) return RATSDF_ERR_BAD_ARGUMENT;
  { const int st0 = e->settle(); if (st0 != RATSDF_OK) return st0; }
  int st = e->select(kSelValid, GridBounds{}, &e->ctl->n_sel);
  if (st != RATSDF_OK) return st;
  hipLaunchKernelGGL(k_export_entries, dim3(256), dim3(256), 0, e->stream, e->vis, &e->ctl->n_sel,
                     (Entry*)d_blocks, (int32_t*)nullptr, (uint32_t)capacity, (int32_t*)d_count, e->ctl);
  HIPCHK(hipGetLastError());
  return RATSDF_OK;
}

// What the directory gained, changed and lost since the previous call (or since creation): the engine keeps a
// dirty bit per entry and a log of deleted positions (device_types.h: Table::dirty / del_log), so the delta costs
// two small kernels instead of a sort of the whole directory on the caller's side.  d_payload receives the
// added / changed entries first, then one entry {position, offset 0, idx -1} per deleted position; d_counts
// (int32[2]) the TRUE numbers of both -- more than `capacity` together means the payload was too small, and
// 0x7FFFFFFF deleted positions that the log overflowed: either way the caller takes a whole directory
// (ratsdf_export_directory_device) next.  A position deleted and inserted again is in both lists: drop, then add.
// d_payload == NULL: forget the changes so far (after a whole-directory export).  Asynchronous on the engine's stream.
int ratsdf_export_directory_delta_device(ratsdf_engine* e, void* d_payload, int32_t capacity, void* d_counts) {
  DeviceGuard guard(e ? e->device : -1);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  if (!e || capacity < 0 || (d_payload && !d_counts)) return RATSDF_ERR_BAD_ARGUMENT;
  { const int st0 = e->settle(); if (st0 != RATSDF_OK) return st0; }
  const uint32_t occ_words = (e->tab.num_entry + 63) / 64;
  if (!e->tab.delta_on) {
    // The first call starts the bookkeeping (an engine nobody asks for deltas keeps none: a dirty-bit atomic per
    // commit and the delete log cost the frame 0.9 us).  Nothing has been recorded so far, so this call cannot
    // deliver a delta: a payload call reports the overflow value and the caller takes a whole directory.
    HIPCHK(hipStreamSynchronize(e->stream));
    e->tab.delta_on = 1;
    const int st1 = e->upload_record();
    if (st1 != RATSDF_OK) return st1;
    HIPCHK(hipMemsetAsync(e->tab.occ + occ_words, 0, (size_t)occ_words * 8, e->stream));
    HIPCHK(hipMemsetAsync(e->tab.del_count, 0, 4, e->stream));
    if (d_payload) {
      const int32_t unusable[2] = {0, 0x7FFFFFFF};
      HIPCHK(hipMemcpyAsync(d_counts, unusable, 8, hipMemcpyHostToDevice, e->stream));
      HIPCHK(hipStreamSynchronize(e->stream));
    }
    return RATSDF_OK;
  }
  if (!d_payload) {
    HIPCHK(hipMemsetAsync(e->tab.occ + occ_words, 0, (size_t)occ_words * 8, e->stream));
    HIPCHK(hipMemsetAsync(e->tab.del_count, 0, 4, e->stream));
    return RATSDF_OK;
  }
  HIPCHK(hipMemsetAsync(d_counts, 0, 8, e->stream));
  hipLaunchKernelGGL(k_delta_added, dim3(e->nwg), dim3(kVisWG), 0, e->stream, e->tab, (Entry*)d_payload,
                     (uint32_t)capacity, (uint32_t*)d_counts);
  hipLaunchKernelGGL(k_delta_deleted, dim3(64), dim3(256), 0, e->stream, e->tab, (Entry*)d_payload,
                     (uint32_t)capacity, (uint32_t*)d_counts);
  hipLaunchKernelGGL(k_delta_reset, dim3(1), dim3(1), 0, e->stream, e->tab);
  HIPCHK(hipGetLastError());
  return RATSDF_OK;
}

// ---- test hooks ------------------------------------------------------------------------------
static int upload_s3(ratsdf_engine* e, const int16_t* src, int32_t n, int16_t** dev) {
  *dev = nullptr;
  if (n == 0) return RATSDF_OK;
  HIPCHK(hipMalloc(dev, (size_t)n * 6));
  HIPCHK(hipMemcpyAsync(*dev, src, (size_t)n * 6, hipMemcpyHostToDevice, e->stream));
  return RATSDF_OK;
}

int ratsdf_test_allocate(ratsdf_engine* e, const int16_t* bp, int32_t n) {
  DeviceGuard guard(e ? e->device : -1);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  if (!e || (!bp && n > 0) || n < 0) return RATSDF_ERR_BAD_ARGUMENT;
  { const int st0 = e->settle(); if (st0 != RATSDF_OK) return st0; }
  if (n == 0) return e->sticky();
  int st = e->ensure_image(0, (size_t)n);
  if (st != RATSDF_OK) return st;
  int16_t* d = nullptr;
  st = upload_s3(e, bp, n, &d);
  if (st != RATSDF_OK) return st;
  FrameParams P = e->base_params();
  const uint32_t par = e->parity;  // an allocation pass of its own in the next frame's counters
  hipLaunchKernelGGL(k_alloc_list, dim3((n + 255) / 256), dim3(256), 0, e->stream, e->tab, P, d, n,
                     e->req, e->req_cap, e->slow, kSlowCap, e->ctl, par);
  st = e->alloc_rank((uint32_t)n, par);
  hipLaunchKernelGGL(k_commit_only, dim3(256), dim3(256), 0, e->stream, e->tab, e->pool, e->req,
                     e->req_cap, e->req_k, e->win_ranks, e->ctl, par);
  // no deletes in this pass; k_settle just zeroes the counters again
  hipLaunchKernelGGL(k_settle, dim3(1), dim3(1024), 0, e->stream, e->tab, e->pool, e->carve_bufs(par),
                     e->ctl, par, (ratsdf_frame_stats*)nullptr);
  const int st2 = e->sticky();
  (void)hipFree(d);
  return st != RATSDF_OK ? st : st2;
}

// The blocks at d_pos (device, n x 3 int16) into the directory -- whatever the engine's shard filter says -- and
// their voxels from device arrays laid out as k_import_voxels describes.  Synchronises the engine's stream (the number
// of blocks the directory still lacks after a pass is read back: control data, 4 bytes per pass).
static int import_from_device(ratsdf_engine* e, int32_t n, const int16_t* d_pos, const float* d_tsdf,
                              const uint32_t* d_rgbw, const float* d_prob, uint32_t stride) {
  e->ever_sem = true;  // (the blocks come with their probabilities: FrameParams::segm_live)
  int st = e->ensure_image(0, (size_t)n);
  if (st != RATSDF_OK) return st;
  uint32_t* d_missing = nullptr;
  if (hipMalloc(&d_missing, 4) != hipSuccess) return RATSDF_ERR_DEVICE;
  auto cleanup = [&](int status) {
    (void)hipStreamSynchronize(e->stream);
    (void)hipFree(d_missing);
    return status;
  };
  FrameParams P = e->base_params();
  P.shard_count = 1;  // whatever the engine's shard filter says
  uint32_t missing = (uint32_t)n;
  // an insertion can lose its bucket to another one of the same pass (one per bucket and pass,
  // voxel_hash.cu:67-78): allocate, copy, and go again for whatever the directory still lacks
  for (int pass = 0; pass < 8 && missing != 0; ++pass) {
    const uint32_t par = e->parity;
    hipLaunchKernelGGL(k_alloc_list, dim3((n + 255) / 256), dim3(256), 0, e->stream, e->tab, P, d_pos, n, e->req,
                       e->req_cap, e->slow, kSlowCap, e->ctl, par);
    st = e->alloc_rank((uint32_t)n, par);
    if (st != RATSDF_OK) return cleanup(st);
    hipLaunchKernelGGL(k_commit_only, dim3(256), dim3(256), 0, e->stream, e->tab, e->pool, e->req, e->req_cap,
                       e->req_k, e->win_ranks, e->ctl, par);
    hipLaunchKernelGGL(k_settle, dim3(1), dim3(1024), 0, e->stream, e->tab, e->pool, e->carve_bufs(par), e->ctl, par,
                       (ratsdf_frame_stats*)nullptr);
    if (hipMemsetAsync(d_missing, 0, 4, e->stream) != hipSuccess) return cleanup(RATSDF_ERR_DEVICE);
    hipLaunchKernelGGL(k_import_voxels, dim3((n + 3) / 4), dim3(256), 0, e->stream, e->tab, e->pool, d_pos, n, d_tsdf,
                       d_rgbw, d_prob, stride, d_missing);
    if (hipMemcpyAsync(&missing, d_missing, 4, hipMemcpyDeviceToHost, e->stream) != hipSuccess ||
        hipStreamSynchronize(e->stream) != hipSuccess)
      return cleanup(RATSDF_ERR_DEVICE);
  }
  st = e->sticky();
  if (st == RATSDF_OK && missing != 0) st = RATSDF_ERR_CAPACITY;
  return cleanup(st);
}

int ratsdf_import_blocks(ratsdf_engine* e, int32_t n, const int16_t* bp, const float* tsdf, const ratsdf_rgbw* rgbw,
                         const float* prob) {
  DeviceGuard guard(e ? e->device : -1);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  if (!e || n < 0 || (n > 0 && (!bp || !tsdf || !rgbw || !prob))) return RATSDF_ERR_BAD_ARGUMENT;
  { const int st0 = e->settle(); if (st0 != RATSDF_OK) return st0; }
  if (n == 0) return e->sticky();
  int16_t* d_pos = nullptr;
  uint8_t* d_vox = nullptr;
  const size_t per = (size_t)n * 512 * 4;
  int st = upload_s3(e, bp, n, &d_pos);
  if (st != RATSDF_OK) return st;
  auto cleanup = [&](int status) {
    (void)hipStreamSynchronize(e->stream);
    if (d_pos) (void)hipFree(d_pos);
    if (d_vox) (void)hipFree(d_vox);
    return status;
  };
  if (hipMalloc(&d_vox, per * 3) != hipSuccess) return cleanup(RATSDF_ERR_DEVICE);
  if (hipMemcpyAsync(d_vox, tsdf, per, hipMemcpyHostToDevice, e->stream) != hipSuccess ||
      hipMemcpyAsync(d_vox + per, rgbw, per, hipMemcpyHostToDevice, e->stream) != hipSuccess ||
      hipMemcpyAsync(d_vox + 2 * per, prob, per, hipMemcpyHostToDevice, e->stream) != hipSuccess)
    return cleanup(RATSDF_ERR_DEVICE);
  return cleanup(import_from_device(e, n, d_pos, (const float*)d_vox, (const uint32_t*)(d_vox + per),
                                    (const float*)(d_vox + 2 * per), 512u));
}

int ratsdf_import_blocks_device(ratsdf_engine* e, int32_t n, const void* d_block_pos, const void* d_voxels) {
  DeviceGuard guard(e ? e->device : -1);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  if (!e || n < 0 || (n > 0 && (!d_block_pos || !d_voxels))) return RATSDF_ERR_BAD_ARGUMENT;
  { const int st0 = e->settle(); if (st0 != RATSDF_OK) return st0; }
  if (n == 0) return e->sticky();
  const uint32_t* rec = (const uint32_t*)d_voxels;
  return import_from_device(e, n, (const int16_t*)d_block_pos, (const float*)rec, rec + 512, (const float*)(rec + 1024),
                            1536u);
}

int ratsdf_export_blocks_device(ratsdf_engine* e, int32_t n, const void* d_block_pos, void* d_voxels,
                                void* d_missing) {
  DeviceGuard guard(e ? e->device : -1);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  if (!e || n < 0 || !d_missing || (n > 0 && (!d_block_pos || !d_voxels))) return RATSDF_ERR_BAD_ARGUMENT;
  { const int st0 = e->settle(); if (st0 != RATSDF_OK) return st0; }
  HIPCHK(hipMemsetAsync(d_missing, 0, 4, e->stream));
  if (n == 0) return RATSDF_OK;
  hipLaunchKernelGGL(k_export_blocks, dim3((n + 3) / 4), dim3(256), 0, e->stream, e->tab, e->pool,
                     (const int16_t*)d_block_pos, n, (uint32_t*)d_voxels, (uint32_t*)d_missing);
  HIPCHK(hipGetLastError());
  return RATSDF_OK;
}

int ratsdf_test_delete(ratsdf_engine* e, const int16_t* bp, int32_t n) {
  DeviceGuard guard(e ? e->device : -1);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  if (!e || (!bp && n > 0) || n < 0) return RATSDF_ERR_BAD_ARGUMENT;
  { const int st0 = e->settle(); if (st0 != RATSDF_OK) return st0; }
  // keep the first occurrence of every position (a repeated Delete is a no-op in list order)
  std::vector<int16_t> uniq;
  uniq.reserve((size_t)n * 3);
  for (int i = 0; i < n; ++i) {
    bool dup = false;
    for (size_t j = 0; j + 2 < uniq.size() && !dup; j += 3)
      dup = uniq[j] == bp[3 * i] && uniq[j + 1] == bp[3 * i + 1] && uniq[j + 2] == bp[3 * i + 2];
    if (!dup) uniq.insert(uniq.end(), bp + 3 * i, bp + 3 * i + 3);
  }
  const int32_t m = (int32_t)(uniq.size() / 3);
  if (m == 0) return e->sticky();
  if (m > e->tab.num_block) return RATSDF_ERR_BAD_ARGUMENT;
  int16_t* d = nullptr;
  int st = upload_s3(e, uniq.data(), m, &d);
  if (st != RATSDF_OK) return st;
  const uint32_t par = e->parity;
  hipLaunchKernelGGL(k_delete_list, dim3((m + 255) / 256), dim3(256), 0, e->stream, e->tab, d, m,
                     e->carve_bufs(par), e->ctl, par);
  hipLaunchKernelGGL(k_settle, dim3(1), dim3(1024), 0, e->stream, e->tab, e->pool, e->carve_bufs(par),
                     e->ctl, par, (ratsdf_frame_stats*)nullptr);
  const int st2 = e->sticky();
  (void)hipFree(d);
  return st != RATSDF_OK ? st : st2;
}

int ratsdf_test_retrieve(ratsdf_engine* e, const int16_t* pts, int32_t n, ratsdf_rgbw* rgbw,
                         float* tsdf, float* prob, ratsdf_block* blocks) {
  DeviceGuard guard(e ? e->device : -1);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  if (!e || (!pts && n > 0) || n < 0) return RATSDF_ERR_BAD_ARGUMENT;
  { const int st0 = e->settle(); if (st0 != RATSDF_OK) return st0; }
  if (n == 0) return RATSDF_OK;
  int16_t* d = nullptr;
  int st = upload_s3(e, pts, n, &d);
  if (st != RATSDF_OK) return st;
  uint8_t* o = nullptr;
  HIPCHK(hipMalloc(&o, (size_t)n * 24));
  uint32_t* o_rgbw = (uint32_t*)o;
  float* o_tsdf = (float*)(o + (size_t)n * 4);
  float* o_prob = (float*)(o + (size_t)n * 8);
  Entry* o_blk = (Entry*)(o + (size_t)n * 12);
  hipLaunchKernelGGL(k_retrieve, dim3((n + 255) / 256), dim3(256), 0, e->stream, e->tab, e->pool, d,
                     n, o_rgbw, o_tsdf, o_prob, o_blk);
  std::vector<uint8_t> h((size_t)n * 24);
  HIPCHK(hipMemcpyAsync(h.data(), o, h.size(), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  if (rgbw) memcpy(rgbw, h.data(), (size_t)n * 4);
  if (tsdf) memcpy(tsdf, h.data() + (size_t)n * 4, (size_t)n * 4);
  if (prob) memcpy(prob, h.data() + (size_t)n * 8, (size_t)n * 4);
  if (blocks) memcpy(blocks, h.data() + (size_t)n * 12, (size_t)n * 12);
  (void)hipFree(o);
  (void)hipFree(d);
  return RATSDF_OK;
}

int ratsdf_test_assign_rgbw(ratsdf_engine* e, const int16_t* pts, const ratsdf_rgbw* vals,
                            int32_t n) {
  DeviceGuard guard(e ? e->device : -1);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  if (!e || ((!pts || !vals) && n > 0) || n < 0) return RATSDF_ERR_BAD_ARGUMENT;
  { const int st0 = e->settle(); if (st0 != RATSDF_OK) return st0; }
  if (n == 0) return RATSDF_OK;
  int16_t* d = nullptr;
  int st = upload_s3(e, pts, n, &d);
  if (st != RATSDF_OK) return st;
  uint32_t* v = nullptr;
  HIPCHK(hipMalloc(&v, (size_t)n * 4));
  HIPCHK(hipMemcpyAsync(v, vals, (size_t)n * 4, hipMemcpyHostToDevice, e->stream));
  hipLaunchKernelGGL(k_assign_rgbw, dim3((n + 255) / 256), dim3(256), 0, e->stream, e->tab, e->pool,
                     d, v, n);
  HIPCHK(hipStreamSynchronize(e->stream));
  (void)hipFree(v);
  (void)hipFree(d);
  return RATSDF_OK;
}

int ratsdf_dump_directory(ratsdf_engine* e, int32_t** entry_index, ratsdf_block** blocks,
                          size_t* n) {
  DeviceGuard guard(e ? e->device : -1);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  if (!e || !entry_index || !blocks || !n) return RATSDF_ERR_BAD_ARGUMENT;
  { const int st0 = e->settle(); if (st0 != RATSDF_OK) return st0; }
  int st = e->select(kSelValid, GridBounds{}, &e->ctl->n_sel);
  if (st != RATSDF_OK) return st;
  uint32_t cnt = 0;
  HIPCHK(hipMemcpyAsync(&cnt, &e->ctl->n_sel, 4, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  int32_t* ei = (int32_t*)malloc(cnt ? (size_t)cnt * 4 : 1);
  ratsdf_block* bl = (ratsdf_block*)malloc(cnt ? (size_t)cnt * 12 : 1);
  if (cnt) {
    Entry* d_b = nullptr;
    int32_t* d_e = nullptr;
    HIPCHK(hipMalloc(&d_b, (size_t)cnt * 12));
    HIPCHK(hipMalloc(&d_e, (size_t)cnt * 4));
    hipLaunchKernelGGL(k_export_entries, dim3(256), dim3(256), 0, e->stream, e->vis, &e->ctl->n_sel,
                       d_b, d_e, cnt, (int32_t*)nullptr, (Ctl*)nullptr);
    HIPCHK(hipMemcpyAsync(bl, d_b, (size_t)cnt * 12, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemcpyAsync(ei, d_e, (size_t)cnt * 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    (void)hipFree(d_b);
    (void)hipFree(d_e);
  }
  *entry_index = ei;
  *blocks = bl;
  *n = cnt;
  return RATSDF_OK;
}

int ratsdf_dump_voxels(ratsdf_engine* e, const int32_t* pool_idx, int32_t n, float* tsdf,
                       ratsdf_rgbw* rgbw, float* prob) {
  DeviceGuard guard(e ? e->device : -1);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  if (!e || (!pool_idx && n > 0) || n < 0) return RATSDF_ERR_BAD_ARGUMENT;
  { const int st0 = e->settle(); if (st0 != RATSDF_OK) return st0; }
  if (n == 0) return RATSDF_OK;
  for (int i = 0; i < n; ++i)
    if (pool_idx[i] < 0 || pool_idx[i] >= e->tab.num_block) return RATSDF_ERR_BAD_ARGUMENT;
  int32_t* d_idx = nullptr;
  uint8_t* d_out = nullptr;
  const size_t per = (size_t)n * 512 * 4;
  HIPCHK(hipMalloc(&d_idx, (size_t)n * 4));
  HIPCHK(hipMalloc(&d_out, per * 3));
  HIPCHK(hipMemcpyAsync(d_idx, pool_idx, (size_t)n * 4, hipMemcpyHostToDevice, e->stream));
  hipLaunchKernelGGL(k_gather_voxels, dim3((n + 3) / 4), dim3(256), 0, e->stream, e->pool, d_idx, n,
                     (float*)d_out, (uint32_t*)(d_out + per), (float*)(d_out + 2 * per));
  if (tsdf) HIPCHK(hipMemcpyAsync(tsdf, d_out, per, hipMemcpyDeviceToHost, e->stream));
  if (rgbw) HIPCHK(hipMemcpyAsync(rgbw, d_out + per, per, hipMemcpyDeviceToHost, e->stream));
  if (prob) HIPCHK(hipMemcpyAsync(prob, d_out + 2 * per, per, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  (void)hipFree(d_idx);
  (void)hipFree(d_out);
  return RATSDF_OK;
}

int ratsdf_dump_heap(ratsdf_engine* e, int32_t* num_free, int32_t* heap) {
  DeviceGuard guard(e ? e->device : -1);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  if (!e) return RATSDF_ERR_BAD_ARGUMENT;
  { const int st0 = e->settle(); if (st0 != RATSDF_OK) return st0; }
  if (num_free)
    HIPCHK(hipMemcpyAsync(num_free, &e->ctl->num_free, 4, hipMemcpyDeviceToHost, e->stream));
  if (heap)
    HIPCHK(hipMemcpyAsync(heap, e->pool.heap, (size_t)e->tab.num_block * 4, hipMemcpyDeviceToHost,
                          e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return RATSDF_OK;
}

const char* ratsdf_status_string(int s) {
  switch (s) {
    case RATSDF_OK: return "ok";
    case RATSDF_ERR_BAD_ARGUMENT: return "bad argument";
    case RATSDF_ERR_DEVICE: return "device / allocation error";
    case RATSDF_ERR_POOL_EXHAUSTED: return "voxel block pool exhausted";
    case RATSDF_ERR_CAPACITY: return "internal work list overflow";
    case RATSDF_ERR_NO_DEVICE: return "no HIP device";
    case RATSDF_ERR_NOT_IMPLEMENTED: return "not implemented";
    case RATSDF_ERR_TIMEOUT: return "in-launch wait between workgroups timed out";
    default: return "unknown status";
  }
}
const char* ratsdf_backend(void) { return "hip-gfx950"; }

}  // extern "C"

// ================================== groups =====================================================
// Several engines (maps) of one device stepped together: frame i of every member stream goes through
// ONE k_front / k_alloc_rank / k_integrate triple whose grids have one slice per engine (blockIdx.y).
// A single 640x480 frame leaves most of the chip waiting on memory round trips and launch ramps; S
// frames per launch fill it.  Operands come from device tables: the engine records (device_types.h:
// EngineDev) and a per-batch table of FrameJob {parameters, image pointers, parity} per frame and slot.
struct ratsdf_group {
  int device = 0;
  int S = 0;
  std::vector<ratsdf_engine*> eng;
  hipStream_t stream = nullptr;
  std::vector<hipEvent_t> ev_member;  // member stream -> group stream
  hipEvent_t ev_done = nullptr;       // group stream -> member streams
  EngineDev* d_engs = nullptr;
  FrameJob* d_jobs = nullptr;
  size_t jobs_cap = 0;
  // page-locked staging of the tables, two of each, used alternately (a copy may still be pending
  // when the next batch is being prepared)
  EngineDev* h_engs[2] = {nullptr, nullptr};
  FrameJob* h_jobs[2] = {nullptr, nullptr};
  size_t h_jobs_cap = 0;
  hipEvent_t ev_stage[2] = {nullptr, nullptr};
  unsigned batch_no = 0;
  int split_a = 100, split_b = 0;  // look-ahead share of k_front / k_alloc_rank (rest: k_integrate)
  bool profiling = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
  size_t prof_used = 0;
  uint64_t prof_frame = 0;
  double prof_ms = 0;
  int64_t prof_n = 0;

  int drain_profile() {
    if (!prof_used) return RATSDF_OK;
    HIPCHK(hipStreamSynchronize(stream));
    for (size_t i = 0; i < prof_used; ++i) {
      float ms = 0;
      HIPCHK(hipEventElapsedTime(&ms, prof_events[i].first, prof_events[i].second));
      prof_ms += ms;
      ++prof_n;
    }
    prof_used = 0;
    return RATSDF_OK;
  }
  void free_all() {
    if (stream) (void)hipStreamSynchronize(stream);
    if (d_engs) (void)hipFree(d_engs);
    if (d_jobs) (void)hipFree(d_jobs);
    for (int i = 0; i < 2; ++i) {
      if (h_engs[i]) (void)hipHostFree(h_engs[i]);
      if (h_jobs[i]) (void)hipHostFree(h_jobs[i]);
      if (ev_stage[i]) (void)hipEventDestroy(ev_stage[i]);
    }
    for (auto& ev : ev_member)
      if (ev) (void)hipEventDestroy(ev);
    if (ev_done) (void)hipEventDestroy(ev_done);
    for (auto& ev : prof_events) {
      (void)hipEventDestroy(ev.first);
      (void)hipEventDestroy(ev.second);
    }
    if (stream) (void)hipStreamDestroy(stream);
  }
};

extern "C" {

int ratsdf_group_create(ratsdf_engine* const* engines, int n, ratsdf_group** out) {
  if (!engines || !out || n < 1 || n > 64) return RATSDF_ERR_BAD_ARGUMENT;
  for (int i = 0; i < n; ++i) {
    const ratsdf_engine* a = engines[i];
    if (!a) return RATSDF_ERR_BAD_ARGUMENT;
    const ratsdf_engine* b = engines[0];
    // one launch geometry for all members
    if (a->device != b->device || a->vs != b->vs || a->trunc != b->trunc ||
        a->block_bits != b->block_bits || a->bucket_bits != b->bucket_bits || a->vpl != b->vpl)
      return RATSDF_ERR_BAD_ARGUMENT;
    for (int j = 0; j < i; ++j)
      if (engines[j] == a) return RATSDF_ERR_BAD_ARGUMENT;
  }
  DeviceGuard guard(engines[0]->device);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  ratsdf_group* g = new (std::nothrow) ratsdf_group();
  if (!g) return RATSDF_ERR_DEVICE;
  g->device = engines[0]->device;
  g->S = n;
  g->eng.assign(engines, engines + n);
  g->ev_member.assign((size_t)n, nullptr);
#ifdef RATSDF_STAMPS
  if (const char* v = getenv("RATSDF_GROUP_SPLIT")) {  // "a[,b]" like RATSDF_CAND_SPLIT
    const int x = atoi(v);
    if (x >= 0 && x <= 100) {
      g->split_a = x;
      g->split_b = 0;
      if (const char* c = strchr(v, ',')) {
        const int y = atoi(c + 1);
        if (y >= 0 && x + y <= 100) g->split_b = y;
      }
    }
  }
#endif
#define GROUP_CHK(expr)                                                  \
  do {                                                                   \
    if ((expr) != hipSuccess) {                                          \
      fprintf(stderr, "[ratsdf] group create failed: %s\n", #expr);      \
      g->free_all();                                                     \
      delete g;                                                          \
      return RATSDF_ERR_DEVICE;                                          \
    }                                                                    \
  } while (0)
  GROUP_CHK(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
  GROUP_CHK(hipMalloc(&g->d_engs, (size_t)n * sizeof(EngineDev)));
  for (int i = 0; i < 2; ++i) {
    GROUP_CHK(hipHostMalloc(&g->h_engs[i], (size_t)n * sizeof(EngineDev), hipHostMallocDefault));
    GROUP_CHK(hipEventCreateWithFlags(&g->ev_stage[i], hipEventDisableTiming));
  }
  for (int i = 0; i < n; ++i)
    GROUP_CHK(hipEventCreateWithFlags(&g->ev_member[(size_t)i], hipEventDisableTiming));
  GROUP_CHK(hipEventCreateWithFlags(&g->ev_done, hipEventDisableTiming));
#undef GROUP_CHK
  *out = g;
  return RATSDF_OK;
}

int ratsdf_group_destroy(ratsdf_group* g) {
  if (!g) return RATSDF_ERR_BAD_ARGUMENT;
  DeviceGuard guard(g->device);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  g->free_all();
  delete g;
  return RATSDF_OK;
}

int ratsdf_group_size(ratsdf_group* g, int32_t* out) {
  if (!g || !out) return RATSDF_ERR_BAD_ARGUMENT;
  *out = g->S;
  return RATSDF_OK;
}

// Frame f of member s is element [f * S + s] of every array.
int ratsdf_group_integrate_device_batch(ratsdf_group* g, int n, const void* const* d_rgb,
                                        const void* const* d_depth, const void* const* d_ht,
                                        const void* const* d_lt, int height, int width,
                                        float max_depth, const ratsdf_intrinsics* K,
                                        const ratsdf_pose* T) {
  if (!g || n < 0 || (n > 0 && (!d_rgb || !d_depth || !K || !T)) || height <= 0 || width <= 0)
    return RATSDF_ERR_BAD_ARGUMENT;
  if (n == 0) return RATSDF_OK;
  const int S = g->S;
  const size_t npix = (size_t)height * width;
  for (size_t i = 0; i < (size_t)n * S; ++i)
    if (!d_rgb[i] || !d_depth[i] || !finite_frame(K[i], T[i], max_depth)) return RATSDF_ERR_BAD_ARGUMENT;
  DeviceGuard guard(g->device);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  ratsdf_engine* e0 = g->eng[0];
  if (npix * (size_t)e0->S >= 0xFFFFFFFFull) return RATSDF_ERR_BAD_ARGUMENT;
  for (ratsdf_engine* e : g->eng) {
    if (e->cand_ready) return RATSDF_ERR_BAD_ARGUMENT;  // cannot happen between complete calls
    const int st = e->ensure_image(npix, npix * (size_t)e->S);
    if (st != RATSDF_OK) return st;
  }
  // ---- tables ----
  const unsigned slot = g->batch_no++ & 1u;
  const size_t njobs = (size_t)n * S;
  if (njobs > g->jobs_cap) {
    HIPCHK(hipStreamSynchronize(g->stream));
    if (g->d_jobs) (void)hipFree(g->d_jobs);
    g->d_jobs = nullptr;
    g->jobs_cap = 0;
    HIPCHK(hipMalloc(&g->d_jobs, njobs * sizeof(FrameJob)));
    g->jobs_cap = njobs;
  }
  if (njobs > g->h_jobs_cap) {
    HIPCHK(hipStreamSynchronize(g->stream));
    for (int i = 0; i < 2; ++i) {
      if (g->h_jobs[i]) (void)hipHostFree(g->h_jobs[i]);
      g->h_jobs[i] = nullptr;
      HIPCHK(hipHostMalloc(&g->h_jobs[i], njobs * sizeof(FrameJob), hipHostMallocDefault));
    }
    g->h_jobs_cap = njobs;
  }
  HIPCHK(hipEventSynchronize(g->ev_stage[slot]));  // the copy that last used this staging pair is done
  for (int s = 0; s < S; ++s) g->h_engs[slot][s] = g->eng[(size_t)s]->record();
  FrameJob* hj = g->h_jobs[slot];
  for (int f = 0; f < n; ++f)
    for (int s = 0; s < S; ++s) {
      const size_t i = (size_t)f * S + s;
      ratsdf_engine* e = g->eng[(size_t)s];
      const void* ht = (d_ht && d_lt) ? d_ht[i] : nullptr;
      const void* lt = (d_ht && d_lt) ? d_lt[i] : nullptr;
      if (!ht || !lt) ht = lt = nullptr;
      const ratsdf_engine::FrameIn in{d_rgb[i], d_depth[i], ht, lt, &K[i], &T[i]};
      FrameJob& j = hj[i];
      j.P = e->frame_params(in, height, width, max_depth);
      j.depth = (const float*)in.depth;
      j.rgb = (const uint8_t*)in.rgb;
      j.ht = (const float*)in.ht;
      j.lt = (const float*)in.lt;
      j.par = (e->parity + (unsigned)f) & 1u;
      j.pad = 0;
    }
  // ---- ordering with the members' own streams (queries, single-engine frames) ----
  for (int s = 0; s < S; ++s) {
    HIPCHK(hipEventRecord(g->ev_member[(size_t)s], g->eng[(size_t)s]->stream));
    HIPCHK(hipStreamWaitEvent(g->stream, g->ev_member[(size_t)s], 0));
  }
  HIPCHK(hipMemcpyAsync(g->d_engs, g->h_engs[slot], (size_t)S * sizeof(EngineDev),
                        hipMemcpyHostToDevice, g->stream));
  HIPCHK(hipMemcpyAsync(g->d_jobs, hj, njobs * sizeof(FrameJob), hipMemcpyHostToDevice, g->stream));
  HIPCHK(hipEventRecord(g->ev_stage[slot], g->stream));

  // ---- launches ----
  EnginePtr engs = (EnginePtr)g->d_engs;
  const bool fused = e0->fused_serial && e0->vpl != 1;
  const uint32_t n_serial_wg = fused ? 8u : 0u;
  ratsdf_engine::Geom g1 = e0->geometry(height, width, true, e0->vpl == 1 ? 100 : g->split_a,
                                        (fused || e0->vpl == 1) ? 0 : g->split_b);
  ratsdf_engine::Geom g0 = e0->geometry(height, width, false, 0, 0);
  // Several members at VGA-sized images: a member's slice of 1 536 update workgroups (two blocks each at 640x480 /
  // 5 mm) instead of 4 096 -- a quarter of those are idle and still have to be dispatched, slice after slice
  // (4 members, round 4: 46.3 k frames/s at 4 096, 47.1 k at 3 072, 48.0 k at 2 048, 48.9 k at 1 536 and 1 024;
  // a single stream measures the same from 1 536 to 4 096)
  if (!e0->grid_from_env && S >= 2 && g0.grid == 4096u) g0.grid = g1.grid = 1536u;
  const unsigned grid0 = g0.grid;
  const uint32_t commit_rot = fused ? e0->commit_rotation(grid0, grid0 * (unsigned)S) : 0u;
  {  // nobody looked ahead for the first frame: its candidate pass runs in line
    AheadGeom all = g0.a;
    all.first_tile = 0;
    all.n_tiles = g0.n_cand_wg * 4;
    hipLaunchKernelGGL(k_cand_g, dim3(g0.n_cand_wg, S), dim3(256), 0, g->stream, engs,
                       (JobPtr)g->d_jobs, all);
  }
  // A failure from here on leaves launches queued on the group's stream on behalf of members that do
  // not know about them: the members are brought to a consistent state before the error is returned
  // (the counterpart of ratsdf_engine::abandon_pipeline for a group).
  auto abandon = [&](int frames_launched) {
    (void)hipStreamSynchronize(g->stream);
    for (ratsdf_engine* e : g->eng) {
      if (frames_launched > 0) {
        e->parity = (e->parity + (unsigned)frames_launched) & 1u;
        e->pending = true;   // the last launched frame owes its carve tail
      }
      // the look-ahead pass of a frame that will not come may have filled the next candidate lists
      if (e->cand[e->parity].count)
        (void)hipMemsetAsync(e->cand[e->parity].count, 0, (size_t)kCandSegs * kCandCountStride * 4, e->stream);
      e->cand_ready = false;
      (void)e->settle();
      (void)hipStreamSynchronize(e->stream);
    }
  };
  if (g->profiling) {  // events for every frame that will be timed, created before anything is launched
    const size_t need = g->prof_used + (size_t)n / 4 + 2;
    while (g->prof_events.size() < need) {
      hipEvent_t a, b;
      HIPCHK(hipEventCreate(&a));
      if (hipEventCreate(&b) != hipSuccess) {
        (void)hipEventDestroy(a);
        return RATSDF_ERR_DEVICE;
      }
      g->prof_events.emplace_back(a, b);
    }
  }
  if (hipGetLastError() != hipSuccess) {  // k_cand_g
    abandon(0);
    return RATSDF_ERR_DEVICE;
  }
  for (int f = 0; f < n; ++f) {
    const bool has_next = f + 1 < n;
    const ratsdf_engine::Geom& gg = has_next ? g1 : g0;
    JobPtr cur = (JobPtr)(g->d_jobs + (size_t)f * S);
    JobPtr nxt = (JobPtr)(g->d_jobs + (size_t)(has_next ? f + 1 : f) * S);
#ifdef RATSDF_STAMPS
    if (e0->tab.tail_on)
      hipLaunchKernelGGL(k_front_g<true>, dim3(gg.n_front_wg, S), dim3(256), 0, g->stream, engs, cur, nxt,
                         (uint32_t)gg.n_vis_wg, (uint32_t)gg.parts, 1u | e0->front_prio, gg.a);
    else
#endif
      hipLaunchKernelGGL(k_front_g<false>, dim3(gg.n_front_wg, S), dim3(256), 0, g->stream, engs, cur, nxt,
                         (uint32_t)gg.n_vis_wg, (uint32_t)gg.parts, 0u, gg.a);
#ifdef RATSDF_STAMPS
    if (!fused) {
      const unsigned extra_b = gg.b.n_tiles ? (gg.b.n_tiles + gg.b.tiles_per_wg - 1) / gg.b.tiles_per_wg : 0;
      hipLaunchKernelGGL(k_alloc_rank_g, dim3(1 + extra_b, S), dim3(1024), kSerialLdsBytes, g->stream,
                         engs, cur, nxt, gg.b);
    }
#endif
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (g->profiling && (g->prof_frame++ % 4 == 0) && g->prof_used < g->prof_events.size()) {
      ev0 = g->prof_events[g->prof_used].first;
      ev1 = g->prof_events[g->prof_used].second;
      ++g->prof_used;
    }
    const unsigned extra_c = ((gg.c.n_tiles + 3) / 4 + 7u) & ~7u;
#define RATSDF_LAUNCH_INTEGRATE_G(V, T, NT)                                                            \
  hipExtLaunchKernelGGL((k_integrate_g<V, T>), dim3(gg.grid + n_serial_wg + extra_c, S), dim3(NT), 0,  \
                        g->stream, ev0, ev1, 0, engs, cur, nxt, (uint32_t)gg.grid, n_serial_wg,        \
                        (uint32_t)extra_c, commit_rot, gg.c)
#ifdef RATSDF_STAMPS
    switch (e0->vpl) {
      case 1: RATSDF_LAUNCH_INTEGRATE_G(1, false, 512); break;
      case 8: RATSDF_LAUNCH_INTEGRATE_G(8, false, RATSDF_INTEG_NT); break;
      case 4: RATSDF_LAUNCH_INTEGRATE_G(4, false, RATSDF_INTEG_NT); break;
      default:
        if (e0->tab.tail_on) RATSDF_LAUNCH_INTEGRATE_G(2, true, RATSDF_INTEG_NT);
        else RATSDF_LAUNCH_INTEGRATE_G(2, false, RATSDF_INTEG_NT);
    }
#else
    RATSDF_LAUNCH_INTEGRATE_G(2, false, RATSDF_INTEG_NT);
#endif
#undef RATSDF_LAUNCH_INTEGRATE_G
    if (hipGetLastError() != hipSuccess) {  // a launch of this frame was refused
      abandon(f);
      return RATSDF_ERR_DEVICE;
    }
    if (g->profiling && g->prof_used >= 4096) {
      const int st = g->drain_profile();
      if (st != RATSDF_OK) {
        abandon(f + 1);
        return st;
      }
    }
  }
  if (hipEventRecord(g->ev_done, g->stream) != hipSuccess) {
    abandon(n);
    return RATSDF_ERR_DEVICE;
  }
  int st_all = RATSDF_OK;
  for (ratsdf_engine* e : g->eng) {
    if (hipStreamWaitEvent(e->stream, g->ev_done, 0) != hipSuccess) st_all = RATSDF_ERR_DEVICE;
    e->parity = (e->parity + (unsigned)n) & 1u;
    e->cand_ready = false;
    e->pending = true;
  }
  if (st_all != RATSDF_OK) (void)hipStreamSynchronize(g->stream);  // ordering by waiting instead
  return st_all;
}

int ratsdf_group_synchronize(ratsdf_group* g) {
  if (!g) return RATSDF_ERR_BAD_ARGUMENT;
  DeviceGuard guard(g->device);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  HIPCHK(hipStreamSynchronize(g->stream));
  int worst = RATSDF_OK;
  for (ratsdf_engine* e : g->eng) {
    const int st = ratsdf_synchronize(e);
    if (st != RATSDF_OK && worst == RATSDF_OK) worst = st;
  }
  return worst;
}

int ratsdf_group_profile_enable(ratsdf_group* g, int enable) {
  if (!g) return RATSDF_ERR_BAD_ARGUMENT;
  DeviceGuard guard(g->device);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  const int st = g->drain_profile();
  g->profiling = enable != 0;
  return st;
}

int ratsdf_group_profile_read(ratsdf_group* g, double* ms, int64_t* launches) {
  if (!g) return RATSDF_ERR_BAD_ARGUMENT;
  DeviceGuard guard(g->device);
  if (!guard.ok()) return RATSDF_ERR_DEVICE;
  const int st = g->drain_profile();
  if (ms) *ms = g->prof_ms;
  if (launches) *launches = g->prof_n;
  g->prof_ms = 0;
  g->prof_n = 0;
  return st;
}

}  // extern "C"
