// device_types.h -- PODs shared by the gfx950 kernels and the host engine.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ratsdf.h"

namespace ratsdf {

constexpr uint32_t kInf = 0xFFFFFFFFu;      // "no claim" in the per-bucket claim table
constexpr int32_t kPlaceholderIdx = 0x40000000;  // entry placed by the resolver, pool idx pending

// VoxelBlock (utils/tsdf/voxel_mem.cuh:75-95): 12-byte hash directory entry.
struct Entry {
  int16_t x, y, z;
  int16_t offset;
  int32_t idx;
};
static_assert(sizeof(Entry) == 12, "directory entry is 12 bytes");

// one element of the per-frame visible list: directory entry copy + where it lives
struct VisItem {
  int16_t x, y, z;
  int16_t offset;
  int32_t idx;
  uint32_t entry;
};
static_assert(sizeof(VisItem) == 16, "visible item is 16 bytes");

// allocation request that survived the claim filter (or was placed by the resolver)
struct Request {
  int16_t x, y, z;
  uint16_t flags;
  uint32_t rank;
  uint32_t entry;
};
static_assert(sizeof(Request) == 16, "request is 16 bytes");
constexpr uint16_t kReqWinner = 1;
constexpr uint16_t kReqPlaced = 2;
constexpr uint16_t kReqSlot1 = 4;  // `entry` is the second entry of the home bucket (filed requests; lets a reader
                                   // that has only the request's first three words rebuild `entry` from the hash)

struct SlowRequest {
  int16_t x, y, z;
  uint16_t pad;
  uint32_t rank;
};
static_assert(sizeof(SlowRequest) == 12, "slow request is 12 bytes");

struct SlowDelete {
  int16_t x, y, z;
  uint16_t state;  // 0 = lost the bucket, 1 = winner, 2 = winner and block released
  uint32_t entry;  // where the block sat when the list was built (= its rank in the carve pass)
  int32_t freed;   // pool index released
};

// what carve_resolve_gate (kernels_carve.h) tells a directory reader
enum GateResult : uint32_t { kGateOpen = 0, kGateWaited = 1, kGateExpired = 2 };

// one successful or pending simple delete of the carve pass: where the block sat, what it held
struct DelItem {
  uint32_t entry;
  int32_t idx;
};

// Per-frame counters.  Two copies, used alternately: frame f works in fr[f & 1] while the end of
// frame f-1 (pool releases, statistics; carve_finalize) is still being settled from fr[(f-1) & 1].
// carve_finalize zeroes the copy it has consumed.
struct FrameCtl {
  uint32_t n_req;          // requests appended this pass
  uint32_t n_slow;         // slow (chained-bucket) requests appended this pass
  uint32_t n_win;          // winners of the allocation pass
  uint32_t alloc_base;     // num_free at the start of the allocation pass
  uint32_t pending;        // this frame's carve pass has not been finalised yet
  uint32_t n_winlist;      // > 0: the winners' raster ranks are listed in win_ranks[0, n_winlist) and
                           // k_integrate derives each winner's order from the list (few winners);
                           // 0: req_k holds the order (many winners, rank bitmap path)
  // new blocks of the frame per XCD list, filed by front_tail_role (kernels_frame.h) behind the visible
  // lists; written once, before k_integrate starts
  uint32_t n_fresh[8];
  uint32_t pad[18];
  // visible blocks per XCD list (image-tile buckets): list l counts in n_list[l * kListStride], one
  // 128-byte line per counter (they take ~2000 atomics per frame; sharing a line serialises them)
  uint32_t n_list[8 * 32];
  // The frame's serial role has published its results (1; 2 = through the general paths, whose plain
  // stores the reader must acquire).  Only consulted when that role runs inside k_integrate.  On a
  // line of its own: the committing workgroups POLL it, and waves polling the line of the counters
  // above slow down the very atomics and loads (requests, deletes) that share it.
  uint32_t serial_done;
  // A frame with thousands of requests (a new view): the serial workgroup shares its pass over the
  // requests with the seven workgroups dispatched beside it (kernels_integrate.h: claim_pass).
  uint32_t help_go;        // its decision: 1 = help, 2 = stay out
  uint32_t help_winners;   // winners listed so far (append cursor of win_ranks)
  uint32_t help_done;      // helpers that have finished and drained their stores
  uint32_t pad2[28];
  // Counters of the carve pass, which k_integrate bumps with atomics while every one of its
  // workgroups reads the first line when it starts: a line of their own as well.
  uint32_t n_delcand;      // slot-0 deletes done by k_integrate (pool release pending)
  uint32_t n_slow_del;     // head / chain deletes waiting for carve_resolve_slow
  uint32_t slow_resolved;  // carve_resolve_slow has run for this frame
  uint32_t pad3[29];
  // The frame's serial role at the TAIL of k_front (kernels_frame.h: front_tail_role): every workgroup of the
  // launch's directory roles (visible list, candidate consumers, pool releases) reports here when its stores
  // have drained; the one that arrives last decides the winners, commits the new blocks and files them in
  // the work lists -- the launch boundary is the hand-off, nothing in k_integrate waits or polls.
  // Two levels (a single address takes ~90 atomics per microsecond and hundreds of workgroups finish
  // together): workgroup b bumps arrive[(b % kArriveSubs) * kListStride], the last one of a sub-counter
  // bumps arrive_top.  One 128-byte line per counter.
  uint32_t arrive[32 * 32];
  uint32_t arrive_top;
  uint32_t front_done;     // 1: the tail role did the frame's allocation pass (k_integrate has no commits to
                           // make and its serial group nothing to do); 0: the role runs inside k_integrate
  uint32_t pad4[30];
};
constexpr int kListStride = 32;
constexpr uint32_t kArriveSubs = 32;
static_assert(sizeof(FrameCtl) == 128 + 1024 + 2 * 128 + 32 * 128 + 128,
              "frame counters: one line + one line per list counter + flag lines + arrival counters");
// New blocks a frame's tail role can file (kernels_frame.h: kTailWinMax), per work list.  Segment l of the
// frame's work lists is seg_cap items long (EngineDev::seg_cap = the segment stride): kFreshCap items for the
// list's new blocks IN FRONT of the segment's start (items -kFreshCap .. -1 of &vis[l * seg_cap]; EngineDev::vis
// points kFreshCap items into its allocation), then up to seg_cap - kFreshCap visible blocks.  k_integrate
// reaches both through one base address (its scalar registers are all in use).
constexpr uint32_t kFreshCap = 768;

// Device-resident control block.
struct Ctl {
  FrameCtl fr[2];
  // --- persistent ---
  int32_t num_free;       // VoxelMemPool::num_free_blocks_
  uint32_t error;         // sticky ratsdf_status
  uint32_t n_sel;         // selected blocks of a query / export
  int32_t free_low;       // lowest num_free ever: pool indices below it have never been handed out (Table::active)
  unsigned long long totals[5];  // frames, sum V, sum U, sum allocated, sum deleted
  // frames by where their serial role ran: tail of k_front | in k_integrate: ordinary, resolver, general path
  unsigned long long paths[4];
  uint32_t* err_flag;     // page-locked host word: non-zero once any error has been recorded (kernels_alloc.h: set_error)
  uint32_t pad1[2];
  unsigned long long stamps[32];  // diagnostic build only
  unsigned long long tstamps[16];  // diagnostic build only: wall-clock timeline of k_front's tail (kernels_frame.h)
  unsigned long long* debug_buf;  // diagnostic build only: per-wave stamps of k_integrate
  unsigned long long dbg[8];      // diagnostic build only: counters of RATSDF_DEBUG=30 (integrate_body.inc)
};
constexpr int kNumLists = 8;  // one block list per XCD; list 8 (the 9th segment) holds this frame's new blocks

// Diagnostic build only (-DRATSDF_STAMPS): thread 0 of the single-workgroup kernels accumulates
// shader-clock stamps per phase into Ctl-adjacent memory; never compiled into the product library.
#ifdef RATSDF_STAMPS
#define RATSDF_DBG(P, n) ((P).debug == (n))  // run-time ablation switches (RATSDF_DEBUG=n)
#define RATSDF_STAMP(buf, i)                                            \
  do {                                                                  \
    if (threadIdx.x == 0) (buf)[i] += (unsigned long long)clock64();    \
  } while (0)
#else
#define RATSDF_DBG(P, n) false
#define RATSDF_STAMP(buf, i) \
  do {                       \
  } while (0)
#endif

struct Quat {
  float x, y, z, w;
};
struct V3 {
  float x, y, z;
};
struct Se3 {
  Quat q;
  V3 t;
};
struct Intr {
  float fx, fy, cx, cy;
};

// everything a frame's kernels need, passed by value
// Threads of the workgroup.  Every launch of this engine runs whole workgroups, so this is the implicit argument
// hidden_group_size_x (code object v5: a 16-bit word 12 bytes into the implicit arguments), fetched with a SCALAR
// load.  `blockDim.x` compiles to "block index < block count ? group size : remainder", and in the large kernels
// the compiler does not fold that: a scalar load, a select of the address, then `global_load_ushort` + wait --
// a vector-memory round trip in front of the first real load of every role that strides by the workgroup size.
__device__ __forceinline__ uint32_t block_threads() {
  const uint16_t* ia = (const uint16_t*)__builtin_amdgcn_implicitarg_ptr();
  return ia[6];
}

struct FrameParams {
  Se3 T;        // cam_T_world
  Se3 Ti;       // world_T_cam (host-computed, voxel_tsdf.cu:459)
  Intr K;
  Intr Ki;
  float vs, trunc, md;
  int W, H;
  int S;        // rank stride: max ray samples per pixel
  int has_sem;
  // 0: no frame with ht / lt has ever been integrated into this map (and no block imported): every probability in it
  // is exactly the initial 0.5 and a TSDF-only frame leaves it there -- log(0.5 / 0.5) = 0, exp(0) = 1, 1 / 2 -- so
  // the update neither loads, computes nor stores the probability of existing blocks (new blocks are still
  // initialised): SURVEY 8d's TSDF-only bytes, 16 instead of 24 per updated voxel
  int segm_live;
  int shard_rank, shard_count, shard_slab_bits;
  int shard_bias;          // multiple of shard_count, >= 32768 (device_math.h: shard_owned)
  uint32_t shard_magic;    // floor(2^32 / shard_count) + 1
  int debug;    // diagnostic switches (0 in production)
};

struct Table {
  Entry* entries;
  uint32_t* claim;          // per-bucket claim of the allocation pass (kInf = free)
  uint32_t* dclaim;         // per-bucket claim of the carve pass (its own table: k_integrate both
                            // releases allocation claims and places carve claims)
  unsigned long long* occ;  // occupancy bitmap of the directory: bit e set <=> entries[e].idx >= 0
  // The live blocks BY POOL INDEX (round 5): active[idx] = {position, idx, hash entry} while pool block idx is in
  // the directory, idx = -1 otherwise.  Written where an entry's pool index is written (commit_request,
  // carve_candidate, carve_resolve_slow).  The per-frame visible list is a dense scan of the slots that have ever
  // been in use -- pool indices are handed out from the top, so those are [Ctl::free_low, num_block) -- instead of a
  // scan of the 512 KiB occupancy bitmap followed by a gather of 12-byte entries out of the 48 MiB table
  // (visible_append_role; SURVEY 7 step 6: "persistent compact active-block list").
  VisItem* active;
  uint32_t num_bucket, num_entry, bucket_mask, entry_mask;
  int32_t num_block;
  // What changed since the last directory-delta export (ratsdf_export_directory_delta_device):
  //   dirty    one bit per entry, in the words right behind the occupancy bitmap (occ[(num_entry >> 6) + (e >> 6)]:
  //            no pointer of its own -- k_integrate has no scalar register to spare), set whenever a live block's
  //            entry is written (commit, chain link edited, a chain element moved into its bucket's head by a
  //            delete): the export lists those entries as they are THEN -- a position lives in one entry, so the
  //            list has no duplicates whatever happened in between
  //   del_log  positions deleted, appended when a frame's deletes are finalised (carve_release_role /
  //            carve_finalize: one returning atomic per frame); a position deleted and inserted again is in both
  //            lists, the receiver drops before it adds; duplicates are harmless
  uint2* del_log;           // {x | y << 16, z}
  uint32_t* del_count;      // entries appended (may exceed del_cap: counted, not stored -- the export says so)
  uint32_t del_cap;
  // Switches (uniform; same-box A/B, round 4: with both features compiled in unconditionally the frame took 27.2
  // instead of 25.4 us): delta_on -- somebody consumes directory deltas (set by the first
  // ratsdf_export_directory_delta_device call): dirty bits and the delete log are kept; tail_on -- the serial role
  // may run at the tail of k_front (RATSDF_FRONT_TAIL=1): requests and heap pushes leave k_front's workgroups as
  // write-through stores, the arrival counters are reset with the frame's other counters
  uint32_t delta_on, tail_on;
};

struct Pool {
  uint32_t* rgbw;
  float* tsdf;
  float* segm;
  int32_t* heap;
};

// a bucket lock taken by the chained-bucket resolver (kernels_alloc.h)
struct XLock {
  uint32_t bucket, time;
};

// ---- buffer bundles of the frame kernels ------------------------------------------------------
struct CarveBufs {
  DelItem* del;        // slot-0 deletes of the frame
  uint32_t del_cap;
  SlowDelete* slow;    // head / chain deletes of the frame
  uint32_t slow_cap;
  uint32_t* upd_wg;    // voxels updated: kUpdCounters counters shared by k_integrate's workgroups
  uint32_t* bitmap;    // delete bitmap indexed by hash entry (many-deletes path)
  uint32_t* summary;
  uint32_t* prefix;
};

// candidate set of a frame (kernels_cand.h)
struct CandSet {
  uint4* list;               // [kCandSegs][seg_cap] {x | y << 16, z, rank, -}
  uint32_t* count;           // [kCandSegs * kCandCountStride]
  uint32_t seg_cap;
};

// buffers of the serial allocation-order role (kernels_alloc.h / kernels_frame.h)
struct RankBufs {
  Request* req;
  uint32_t req_cap;
  uint32_t* req_k;
  const SlowRequest* slow;
  uint32_t slow_cap;
  XLock* xlocks;
  uint32_t* win_ranks;  // raster ranks of the winners (few-winners path, kSmallRank entries)
  uint32_t* bitmap;   // rank bitmap (many-requests path)
  uint32_t* summary;
  uint32_t* prefix;
  uint32_t nwords;
  unsigned long long* sort_scratch;  // kSlowSortCap keys: resolver's sort beyond what fits LDS
};

// Everything about one engine that is constant between (re)allocations, resident in device memory:
// the kernels' rarely taken paths (space carving, commit of new blocks) read their operands from
// here instead of holding them in scalar registers for the whole launch, and launches that serve
// several engines at once (one engine per blockIdx.y) index an array of these.
struct EngineDev {
  Table tab;
  Pool pool;
  CarveBufs cb[2];  // per frame parity
  RankBufs rb;
  Ctl* ctl;
  ratsdf_frame_stats* stats;
  SlowRequest* slow;
  uint32_t slow_cap;
  uint32_t seg_cap;
  VisItem* vis;
  uint32_t* serial_scratch;  // 40 KiB: scratch of the serial role's rare paths when it has no LDS to spare
  float4* texA[2];
  uint32_t* texB[2];
  CandSet cand[2];
};
// scalar (SMEM) loads whatever the surrounding code stores: constant address space
typedef const EngineDev __attribute__((address_space(4))) * EnginePtr;
// copy of one member (or the whole record) into registers
template <typename T>
__device__ inline T ld_const(const T __attribute__((address_space(4))) * p) {
  T r;
  __builtin_memcpy(&r, p, sizeof(T));
  return r;
}

}  // namespace ratsdf
