// kernels_alloc.h -- voxel-block allocation pass for gfx950.
//
// Replaces block_allocate_kernel + VoxelHashTable::Allocate + VoxelMemPool::AquireBlock +
// ResetLocks (utils/tsdf/voxel_tsdf.cu:120-168,454-463; voxel_hash.cu:46-108; voxel_mem.cu:37-54).
//
// The reference serialises insertions with per-bucket try-locks that are only released after the
// pass, so which thread wins a bucket and which pool block it pops are timing dependent.  Here the
// outcome is made a pure function of the input: every candidate carries its raster rank
// (pixel * S + sample) and the pass computes exactly what a sequential, rank-ordered execution of
// the reference code would:
//   cand_pixels_role   per pixel: candidate blocks -> the frame's candidate lists {block, smallest
//                      rank per workgroup}; no directory access (kernels_cand.h; for batched frames
//                      it runs one frame ahead, inside the previous frame's launches)
//   cand_consume_role  per candidate: directory lookup, frustum test, atomicMin(claim[bucket], rank)
//                      for ordinary buckets ("first requester in raster order wins the bucket
//                      lock"), append to a small slow list for chained / full buckets (k_front)
//   serial_frame_role  one workgroup (workgroup 0 of k_integrate as serial_role256, kernels_integrate.h;
//                      k_alloc_rank, kernels_frame.h, as a launch of its own): (a) replays the slow list in
//                      rank order against the claim table (time-dependent lock / fill queries), so
//                      chain appends lock, link and defeat later claims exactly as the sequential
//                      code would; (b) winners = requests whose rank equals their bucket's claim;
//                      (c) their ranks are listed (few) or marked in a rank-indexed bitmap with a
//                      popcount prefix (many) = order of the AquireBlock calls
//   commit_request     (inside k_integrate, or k_commit_only for the test hook) one wave per winner:
//                      pool index heap[free-1-k], directory entry, occupancy bit
#pragma once
#include "device_math.h"

namespace ratsdf {

// (the operands are pinned to the call site: hoisted out of a hot loop as loop invariants they were
// the first thing the register allocator spilled to scratch memory, for a path that is never taken)
// The first error sticks in Ctl::error (device memory).  Every error is ALSO flagged in a word of page-locked host
// memory (Ctl::err_flag, system-scope store): a synchronising call reads that word after the stream has drained and
// fetches the device word only when it is set -- the ordinary call pays for no device-to-host copy (engine: sticky()).
__device__ inline void set_error(Ctl* ctl, uint32_t code) {
  uint32_t expect = 0u, value = code;
  asm volatile("" : "+v"(expect), "+v"(value));
  atomicCAS(&ctl->error, expect, value);
  __hip_atomic_store(ctl->err_flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// one thread: the words of a FrameCtl that are in use
__device__ inline void zero_frame_ctl(FrameCtl* F, bool tail_on) {
  F->n_req = 0;
  F->n_slow = 0;
  F->n_win = 0;
  F->alloc_base = 0;
  F->pending = 0;
  F->n_winlist = 0;
  F->serial_done = 0;
  F->help_go = 0;
  F->help_winners = 0;
  F->help_done = 0;
  F->n_delcand = 0;
  F->n_slow_del = 0;
  F->slow_resolved = 0;
#pragma unroll
  for (int l = 0; l < kNumLists; ++l) F->n_list[l * kListStride] = 0;
  if (tail_on) {  // (uniform) the words only front_tail_role's frames use: 42 more lines
    F->arrive_top = 0;
    F->front_done = 0;
#pragma unroll
    for (uint32_t l = 0; l < kArriveSubs; ++l) F->arrive[l * kListStride] = 0;
#pragma unroll
    for (int l = 0; l < kNumLists; ++l) F->n_fresh[l] = 0;
  }
}
// visible blocks of a frame for the statistics: the blocks listed by the visible role + the frame's new blocks
__device__ inline uint32_t frame_visible_blocks(const FrameCtl* F) {
  uint32_t nv = F->n_win;
#pragma unroll
  for (int l = 0; l < kNumLists; ++l) nv += F->n_list[l * kListStride];
  return nv;
}
__device__ inline uint32_t ld_agent_u32(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct EntryWords {
  uint32_t w0, w1;  // x | y << 16,  z | offset << 16
  int32_t idx;
};

__device__ inline EntryWords load_entry(const Entry* entries, uint32_t e) {
  const uint32_t* p = reinterpret_cast<const uint32_t*>(entries + e);
  EntryWords r;
  r.w0 = p[0];
  r.w1 = p[1];
  r.idx = (int32_t)p[2];
  return r;
}
__device__ inline int entry_offset(const EntryWords& w) { return (int16_t)(w.w1 >> 16); }
__device__ inline bool entry_matches(const EntryWords& w, uint32_t k0, uint32_t k1) {
  return w.idx >= 0 && w.w0 == k0 && (w.w1 & 0xFFFFu) == k1;
}
__device__ inline uint32_t key0(int x, int y) { return ((uint32_t)x & 0xFFFFu) | ((uint32_t)y << 16); }
__device__ inline uint32_t key1(int z) { return (uint32_t)z & 0xFFFFu; }

// directory-delta bookkeeping (device_types.h: Table, "what changed since the last directory-delta export")
__device__ inline void mark_dirty(const Table& t, uint32_t e) {
  if (t.delta_on) atomicOr(&t.occ[(t.num_entry >> 6) + (e >> 6)], 1ull << (e & 63));
}

// VoxelHashTable::GetBlock(pos, out), voxel_hash.cu:190-218.  Returns the entry index or kInf.
__device__ inline uint32_t find_block(const Table& t, int x, int y, int z, EntryWords* out) {
  const uint32_t k0 = key0(x, y), k1 = key1(z);
  const uint32_t e0 = block_hash(x, y, z, t.bucket_mask) << 1;
  const EntryWords a = load_entry(t.entries, e0);      // both home entries are fetched together
  EntryWords w = load_entry(t.entries, e0 + 1);         // (24 contiguous bytes)
  if (entry_matches(a, k0, k1)) { *out = a; return e0; }
  if (entry_matches(w, k0, k1)) { *out = w; return e0 + 1; }
  uint32_t last = e0 + 1;
  int off = entry_offset(w);
  uint32_t guard = 0;
  while (off && guard++ < t.num_entry) {
    last = (last + (uint32_t)off) & t.entry_mask;
    w = load_entry(t.entries, last);
    if (entry_matches(w, k0, k1)) { *out = w; return last; }
    off = entry_offset(w);
  }
  out->w0 = k0;
  out->w1 = k1 | 0xFFFF0000u;  // offset -1
  out->idx = -1;
  return kInf;
}

// One allocation request with raster rank `rank` for block (x,y,z), against the pre-pass directory.
__device__ inline void alloc_request(const Table& t, int x, int y, int z, uint32_t rank, Request* req,
                                     uint32_t req_cap, SlowRequest* slow, uint32_t slow_cap,
                                     Ctl* ctl, FrameCtl* F) {
  const uint32_t k0 = key0(x, y), k1 = key1(z);
  const uint32_t bucket = block_hash(x, y, z, t.bucket_mask);
  const uint32_t e0 = bucket << 1;
  const EntryWords a = load_entry(t.entries, e0);
  const EntryWords b = load_entry(t.entries, e0 + 1);
  if (entry_matches(a, k0, k1) || entry_matches(b, k0, k1)) return;  // voxel_hash.cu:50-56
  uint32_t last = e0 + 1;
  int off = entry_offset(b);
  uint32_t guard = 0;
  while (off && guard++ < t.num_entry) {                               // voxel_hash.cu:58-65
    last = (last + (uint32_t)off) & t.entry_mask;
    const EntryWords w = load_entry(t.entries, last);
    if (entry_matches(w, k0, k1)) return;
    off = entry_offset(w);
  }
  // Buckets whose two home entries are full, or that head a chain, can reach the list-append code
  // (voxel_hash.cu:79-106), which touches other buckets' locks: those go to the resolver.
  const bool special = (a.idx >= 0 && b.idx >= 0) || entry_offset(b) != 0;
  if (!special) {
    const uint32_t old = atomicMin(&t.claim[bucket], rank);
    if (rank < old) {
      // the leader of an ordinary bucket fills its first empty home entry (voxel_hash.cu:67-78);
      // nothing else can take that slot during the pass, so it is fixed here
      const uint32_t e = (bucket << 1) + (a.idx < 0 ? 0u : 1u);
      const uint32_t slot = atomicAdd(&F->n_req, 1u);
      if (slot < req_cap) {
        req[slot] = Request{(int16_t)x, (int16_t)y, (int16_t)z, (uint16_t)(a.idx < 0 ? 0 : kReqSlot1), rank, e};
      } else {
        set_error(ctl, RATSDF_ERR_CAPACITY);
      }
    }
  } else {
    const uint32_t slot = atomicAdd(&F->n_slow, 1u);
    if (slot < slow_cap) {
      slow[slot] = SlowRequest{(int16_t)x, (int16_t)y, (int16_t)z, 0, rank};
    } else {
      set_error(ctl, RATSDF_ERR_CAPACITY);
    }
  }
}

// Presence test with the two home entries already loaded (chain walk only when the head links on).
__device__ inline bool block_present_pre(const Table& t, uint32_t k0, uint32_t k1, uint32_t e0,
                                         const EntryWords& a, const EntryWords& b) {
  if (entry_matches(a, k0, k1) || entry_matches(b, k0, k1)) return true;
  uint32_t last = e0 + 1;
  int off = entry_offset(b);
  uint32_t guard = 0;
  while (off && guard++ < t.num_entry) {
    last = (last + (uint32_t)off) & t.entry_mask;
    const EntryWords w = load_entry(t.entries, last);
    if (entry_matches(w, k0, k1)) return true;
    off = entry_offset(w);
  }
  return false;
}

// Requests of a consumer workgroup are collected in LDS and appended to the frame's request list with
// ONE returning atomic per workgroup at the end (a per-lane or per-wave atomicAdd on that single
// address, ~90 ops/us, was the longest step of k_front for large or finely resolved images).
constexpr uint32_t kReqBufCap = 512;
struct ReqBuf {
  Request item[kReqBufCap];
  uint32_t n, base;
};
// A request goes to the frame's list with write-through stores: the workgroup of the same launch that runs
// the frame's serial role (front_tail_role, kernels_frame.h) reads the list past the caches.
__device__ inline void st_agent_request(Request* p, const Request& r) {
  unsigned long long w[2];
  __builtin_memcpy(w, &r, sizeof(r));
  unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
  __hip_atomic_store(q, w[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(q + 1, w[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// alloc_request_absent for a whole wave (every lane calls it; `want` selects the lanes that have an
// absent, visible block).
__device__ inline void alloc_request_absent_wave(bool want, const Table& t, int x, int y, int z,
                                                 uint32_t rank, const EntryWords& a,
                                                 const EntryWords& b, Request* req, uint32_t req_cap,
                                                 SlowRequest* slow, uint32_t slow_cap, Ctl* ctl,
                                                 FrameCtl* F, ReqBuf& B) {
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t bucket = block_hash(x, y, z, t.bucket_mask);
  const bool special = (a.idx >= 0 && b.idx >= 0) || entry_offset(b) != 0;
  bool app = false;
  if (want && !special) app = rank < atomicMin(&t.claim[bucket], rank);
  const unsigned long long m = __ballot(app);
  if (m) {  // uniform
    const uint32_t cnt = (uint32_t)__popcll(m);
    uint32_t base = 0;
    const int leader = __ffsll((long long)m) - 1;
    if ((int)lane == leader) base = atomicAdd(&B.n, cnt);
    base = __shfl(base, leader);
    const bool in_lds = base + cnt <= kReqBufCap;  // uniform
    if (!in_lds) {  // buffer full: straight to the global list
      if ((int)lane == leader) {
        atomicSub(&B.n, cnt);
        base = atomicAdd(&F->n_req, cnt);
      }
      base = __shfl(base, leader);
    }
    if (app) {
      const uint32_t slot = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      // the leader of an ordinary bucket fills its first empty home entry (voxel_hash.cu:67-78);
      // nothing else can take that slot during the pass, so it is fixed here
      const uint32_t e = (bucket << 1) + (a.idx < 0 ? 0u : 1u);
      const Request r{(int16_t)x, (int16_t)y, (int16_t)z, (uint16_t)(a.idx < 0 ? 0 : kReqSlot1), rank, e};
      if (in_lds) {
        B.item[slot] = r;
      } else if (slot < req_cap) {
        if (t.tail_on) st_agent_request(req + slot, r);
        else req[slot] = r;
      } else {
        set_error(ctl, RATSDF_ERR_CAPACITY);
      }
    }
  }
  if (want && special) {
    const uint32_t slot = atomicAdd(&F->n_slow, 1u);
    if (slot < slow_cap) {
      slow[slot] = SlowRequest{(int16_t)x, (int16_t)y, (int16_t)z, 0, rank};
    } else {
      set_error(ctl, RATSDF_ERR_CAPACITY);
    }
  }
}

// all threads of the workgroup: append the collected requests to the frame's list
__device__ inline void req_buf_flush(ReqBuf& B, Request* req, uint32_t req_cap, Ctl* ctl, FrameCtl* F,
                                     bool write_through) {
  __syncthreads();
  const uint32_t n = B.n < kReqBufCap ? B.n : kReqBufCap;
  if (n == 0) return;  // uniform
  if (threadIdx.x == 0) B.base = atomicAdd(&F->n_req, n);
  __syncthreads();
  const uint32_t base = B.base;
  for (uint32_t i = threadIdx.x; i < n; i += block_threads()) {
    if (base + i < req_cap) {
      if (write_through) st_agent_request(req + base + i, B.item[i]);
      else req[base + i] = B.item[i];
    } else {
      set_error(ctl, RATSDF_ERR_CAPACITY);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// test hook: explicit request list, rank = list index (utils/tests/voxel_hash_test.cu:36-39)
__global__ void k_alloc_list(Table tab, FrameParams P, const int16_t* pos, int n, Request* req,
                             uint32_t req_cap, SlowRequest* slow, uint32_t slow_cap, Ctl* ctl,
                             uint32_t par) {
  const int i = blockIdx.x * block_threads() + threadIdx.x;
  if (i >= n) return;
  const int x = pos[3 * i], y = pos[3 * i + 1], z = pos[3 * i + 2];
  if (!shard_owned(x, P)) return;
  alloc_request(tab, x, y, z, (uint32_t)i, req, req_cap, slow, slow_cap, ctl, &ctl->fr[par]);
}

// ---------------------------------------------------------------------------------------------
// Exact rank-ordered replay of VoxelHashTable::Allocate (voxel_hash.cu:46-108) for requests whose
// home bucket is full or heads a chain.  Runs in ONE workgroup; thread 0 does the serial part after
// an LDS bitonic sort of the slow list by rank.  Everything an ordinary bucket does during the pass
// is summarised by its claim (min rank): bucket x is locked from time claim[x] on, and its leader
// fills the first empty home entry at that time unless an earlier slow request locked x first.
// ---------------------------------------------------------------------------------------------
constexpr int kSlowSortCap = 16384;   // slow requests per pass (bitonic sort by rank)
constexpr int kSlowLdsCap = 4096;     // ... sorted in LDS up to this many, in global scratch beyond
// dynamic LDS of the serial role: sort keys | rank lists (alloc_rank_role) | carve_finalize scratch
constexpr int kSerialLdsBytes = 34 * 1024;
constexpr int kSlowDistinctCap = 1024;
constexpr int kXLockCap = 2048;


// What Allocate would do for one chained request against the directory as it stands before the pass
// (resolve_slow_requests): 64 bytes, written by the workgroup, replayed by one thread.
struct SlowPlan {
  uint32_t flags;       // kPlan* | chain entries walked << 8
  uint32_t last, last_w1;  // the chain's tail (entry index, its second word)
  uint32_t next;        // free slot-0 entry found after the tail
  uint32_t c_home, c_last, c_next;  // claims of the three buckets
  uint32_t k0, k1, rank, bucket;    // the request
  uint32_t chain_b[3];  // buckets of the chain entries walked
  uint32_t idx;         // the request's place in the slow list
  uint32_t pad;
};
static_assert(sizeof(SlowPlan) == 64, "four 16-byte words");
enum : uint32_t {
  kPlanDup = 1,        // same block as an earlier request of the pass
  kPlanPresent = 2,    // the block exists
  kPlanHome = 4,       // an empty home entry: slot 0, or slot 1 with kPlanHomeSlot1
  kPlanHomeSlot1 = 8,
  kPlanFound = 16,     // append behind `last`, into `next`
  kPlanComplex = 32,   // long chain or long probe: replayed from memory
};
constexpr uint32_t kPlanSpan = 8;  // buckets probed for a plan
constexpr uint32_t kSlowPlanCap = 2048;  // requests of a pass that get a plan (their own 128 KiB behind the sort keys)

// Pointer to T in LDS (address space 3: ds_read / ds_write) or anywhere (flat).  The resolver's tables sit
// in LDS on its ordinary path; through generic pointers every access was a FLAT instruction, which waits
// for the vector-memory counter as well and cost the replaying thread ~300 cycles apiece.
template <bool Lds, typename T>
struct PtrOf {
  using type = T*;
};
template <typename T>
struct PtrOf<true, T> {
  using type = __attribute__((address_space(3))) T*;
};
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// A set of bucket numbers (open addressing, 0 = empty slot, bucket + 1 stored): the buckets the pass's
// chained requests have locked so far.
template <bool Lds>
struct LockSet {
  typename PtrOf<Lds, uint32_t>::type slot;
  uint32_t mask;  // slots - 1 (a power of two); at most half of them are ever filled
};
// (a set in device memory is read and written past the CU's L1: its slots are claimed with atomics, which
// act in L2)
template <bool Lds>
__device__ inline uint32_t lockset_slot(const LockSet<Lds>& L, uint32_t h) {
  if constexpr (Lds) return L.slot[h];
  else return __hip_atomic_load(&L.slot[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool Lds>
__device__ inline bool lockset_has(const LockSet<Lds>& L, uint32_t bucket) {
  for (uint32_t h = (bucket * 2654435761u) & L.mask;; h = (h + 1) & L.mask) {
    const uint32_t v = lockset_slot(L, h);
    if (v == 0) return false;
    if (v == bucket + 1) return true;
  }
}
// (several lanes may add at once -- distinct buckets: the slot is claimed with a compare-and-swap)
template <bool Lds>
__device__ inline void lockset_add(const LockSet<Lds>& L, uint32_t bucket) {
  for (uint32_t h = (bucket * 2654435761u) & L.mask;; h = (h + 1) & L.mask) {
    uint32_t expected = 0;
    if (__hip_atomic_compare_exchange_strong(&L.slot[h], &expected, bucket + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                             Lds ? __HIP_MEMORY_SCOPE_WORKGROUP : __HIP_MEMORY_SCOPE_AGENT) ||
        expected == bucket + 1)
      return;
  }
}

// Exact rank-ordered replay of VoxelHashTable::Allocate for the pass's chained requests (see above).
// All threads of one workgroup.  Round 3 (at 1280x720 / 2 mm a map past 100 k blocks files 100-250 of them
// per frame; one thread replaying them against memory, with linear scans of the locks and of the blocks
// already seen, took 8-10 us per request, 1-4 ms per frame) splits the work in four:
//  1. order and duplicates, whole workgroup: every request counts the keys (rank, index) below its own
//     -- its place in rank order -- and looks for an earlier-ranked request naming the same block, which
//     makes it irrelevant to the replay (same block, later rank);
//  2. one PLAN per request, whole workgroup: the replay's reads (home entries, chain, the probe for a
//     free slot and the claims beside them) done against the directory as it stands BEFORE the pass, with
//     their answer and the buckets that answer came from;
//  3. the replay proper, the first wave, 64 requests per batch and one per lane, in rank order: a plan
//     holds unless one of its buckets has been locked by an earlier request of the pass -- every edit of
//     the pass (an entry placed, a tail linked, a claim cleared) is made under that bucket's lock -- and
//     what a holding plan does is known beforehand (its claims against its rank).  The wave steps through
//     the batch; the request whose turn it is announces its locks (readlane), every later lane tests them
//     against its plan's buckets.  A stale plan (one pass in ten has one) has the batch's outcomes so far
//     stored, then reads the directory again from memory as the reference would (one lane);
//  4. locks into the set and outcomes into memory, by all lanes of the batch at once.  (No store is issued
//     inside the step loop: on this hardware a wave's loads and stores share one counter, and a loop that
//     waited for its next plan also waited for the write-through stores of the last request.)
// `xlocks`: kXLockCap * 8 bytes of device memory, the lock set unless the caller has LDS for it
// (`lds_locks`, `lds_lock_slots` words, a power of two).
// `global_keys`: kSlowSortCap * 8 bytes for the sort keys of passes beyond `lds_cap` requests, followed by
// kSlowPlanCap plans.
// kLds: keys and lock set in LDS (the pass has at most lds_cap requests); else keys in LDS up to lds_cap
// requests, in `global_keys` beyond, and the lock set in `xlocks`.
// Returns false (uniform, nothing edited yet) when the LDS lock set is too small for the pass's distinct
// requests: the caller then runs the <false> instantiation.
template <bool kLds>
__device__ inline bool resolve_slow_requests(const Table& tab, Request* req, uint32_t req_cap,
                                             const SlowRequest* slow, uint32_t slow_cap,
                                             XLock* xlocks, Ctl* ctl,
                                             FrameCtl* F, unsigned long long* lds_keys,
                                             unsigned long long* global_keys,
                                             uint32_t lds_cap,
                                             uint32_t* lds_locks = nullptr, uint32_t lds_lock_slots = 0) {
  using KeyPtr = typename PtrOf<kLds, unsigned long long>::type;
  using WordPtr = typename PtrOf<kLds, uint32_t>::type;
  const uint32_t tid = threadIdx.x, nt = block_threads();
  uint32_t n = F->n_slow;
  if (n > slow_cap) n = slow_cap;
  if (n > (uint32_t)kSlowSortCap) {
    if (tid == 0) set_error(ctl, RATSDF_ERR_CAPACITY);
    n = kSlowSortCap;
  }
  // few keys (always, with the default directory size): LDS; many: a global scratch buffer, which is
  // coherent inside one workgroup (same CU, write-through L1, __syncthreads() drains the stores)
  const KeyPtr skeys = (KeyPtr)(kLds || n <= lds_cap ? lds_keys : global_keys);
  RATSDF_STAMP(ctl->stamps, 20);
  // the lock set: every distinct request takes at most two locks
  static_assert(sizeof(XLock) == 8 && (kXLockCap & (kXLockCap - 1)) == 0 && kXLockCap >= 2 * kSlowDistinctCap,
                "xlocks holds 2 * kXLockCap set slots, at most half filled");
  LockSet<kLds> locks{(WordPtr) reinterpret_cast<uint32_t*>(xlocks), 2u * (uint32_t)kXLockCap - 1u};
  if (kLds) {  // as many slots as the pass can need (4 per request), at most what the caller has
    uint32_t slots = 64;
    while (slots < 4u * n && slots < lds_lock_slots) slots <<= 1;
    locks = LockSet<kLds>{(WordPtr)lds_locks, slots - 1u};
  }
  constexpr uint32_t kDup = 0x80000000u;  // in the index half of a sorted key
  auto block_key = [](const SlowRequest& s) -> unsigned long long {
    return (unsigned long long)(uint16_t)s.x | ((unsigned long long)(uint16_t)s.y << 16) |
           ((unsigned long long)(uint16_t)s.z << 32);
  };
  // the blocks' names side by side in the lock set's memory (not in use yet) when they fit
  const bool staged = 2u * n <= locks.mask + 1u;  // uniform
  const KeyPtr bk = (KeyPtr)locks.slot;

  // ---- 1. order and duplicates -------------------------------------------------------------------
  if (n <= lds_cap && n <= 2u * nt && staged) {  // uniform: two requests per thread, by counting
    unsigned long long k[2] = {~0ull, ~0ull}, b[2] = {0, 0};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const uint32_t i = tid + (uint32_t)u * nt;
      if (i < n) {
        const SlowRequest s = slow[i];
        k[u] = ((unsigned long long)s.rank << 32) | i;
        b[u] = block_key(s);
        skeys[i] = k[u];
        bk[i] = b[u];
      }
    }
    __syncthreads();
    uint32_t pos[2] = {0, 0};
    bool dup[2] = {false, false};
    if (n <= nt) {  // uniform: one request per thread (the usual case)
#pragma unroll 8
      for (uint32_t j = 0; j < n; ++j) {  // (every lane reads the same words: LDS broadcasts)
        const unsigned long long kj = skeys[j], bj = bk[j];
        const bool before = kj < k[0];
        pos[0] += before;
        dup[0] |= before & (bj == b[0]);
      }
    } else {
#pragma unroll 4
      for (uint32_t j = 0; j < n; ++j) {
        const unsigned long long kj = skeys[j], bj = bk[j];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const bool before = kj < k[u];
          pos[u] += before;
          dup[u] |= before & (bj == b[u]);
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u)
      if (tid + (uint32_t)u * nt < n) skeys[pos[u]] = k[u] | (dup[u] ? kDup : 0u);
  } else {  // many requests: bitonic sort, then every request looks at the ones before it
    uint32_t m = 1;
    while (m < n) m <<= 1;
    for (uint32_t i = tid; i < m; i += nt)
      skeys[i] = i < n ? (((unsigned long long)slow[i].rank << 32) | i) : ~0ull;
    __syncthreads();
    for (uint32_t k = 2; k <= m; k <<= 1) {
      for (uint32_t j = k >> 1; j > 0; j >>= 1) {
        for (uint32_t i = tid; i < m; i += nt) {
          const uint32_t l = i ^ j;
          if (l > i) {
            const unsigned long long a = skeys[i], b = skeys[l];
            const bool up = (i & k) == 0;
            if ((a > b) == up) {
              skeys[i] = b;
              skeys[l] = a;
            }
          }
        }
        __syncthreads();
      }
    }
    if (staged) {
      for (uint32_t i = tid; i < n; i += nt) bk[i] = block_key(slow[(uint32_t)skeys[i]]);
      __syncthreads();
    }
    uint32_t dup_bits = 0;  // of this thread's requests i = tid, tid + nt, ... (staged: at most 2048 / 256 = 8 of them)
    for (uint32_t i = tid, t = 0; i < n; i += nt, ++t) {
      const unsigned long long mine = staged ? bk[i] : block_key(slow[(uint32_t)skeys[i] & ~kDup]);
      bool dup = false;
      if (staged) {
#pragma unroll 8
        for (uint32_t j = 0; j < i; ++j) dup |= bk[j] == mine;
      } else {
        for (uint32_t j = 0; j < i && !dup; ++j) dup = block_key(slow[(uint32_t)skeys[j] & ~kDup]) == mine;
        if (dup) ((WordPtr)(skeys + i))[0] |= kDup;  // (little endian: the index half)
      }
      if (staged && dup) dup_bits |= 1u << (t & 31);
    }
    if (staged) {  // flags into the keys
      __syncthreads();
      for (uint32_t i = tid, t = 0; i < n; i += nt, ++t)
        if ((dup_bits >> (t & 31)) & 1u) ((WordPtr)(skeys + i))[0] |= kDup;
    }
  }
  __syncthreads();  // (the duplicate flags are in the keys)
  if constexpr (kLds) {
    // every distinct request takes at most two locks and the set is to stay at most half full
    if (4u * n > locks.mask + 1u) {  // uniform: more requests than a quarter of the slots -- count the distinct ones
      uint32_t mine = 0;
      for (uint32_t i = tid; i < n; i += nt) mine += !((uint32_t)skeys[i] & kDup);
      uint32_t distinct = 0;
      for (uint32_t r = 0; r < 4; ++r) distinct += (uint32_t)__syncthreads_count(mine > r);  // (n <= 4 * block_threads())
      if (4u * distinct > locks.mask + 1u) return false;  // uniform
    }
  }
  // (nobody reads a block name past the barriers above: the memory becomes the lock set)
  for (uint32_t i = tid; i <= locks.mask; i += nt) locks.slot[i] = 0;
  __syncthreads();
  RATSDF_STAMP(ctl->stamps, 22);

  // ---- 2. plans ----------------------------------------------------------------------------------
  // (in the sort scratch, which a pass of at most lds_cap requests leaves unused; a pass without plans
  // still does the loads: they warm the caches for the replay from memory)
  const bool planned = n <= kSlowPlanCap;  // uniform
  SlowPlan* plans = reinterpret_cast<SlowPlan*>(global_keys + kSlowSortCap);
  for (uint32_t i = tid; i < n; i += nt) {
    const uint32_t key_lo = (uint32_t)skeys[i];
    SlowPlan pl;
    pl.flags = kPlanDup;
    pl.last = pl.last_w1 = pl.next = 0;
    pl.c_home = pl.c_last = pl.c_next = kInf;
    pl.k0 = pl.k1 = pl.rank = pl.bucket = 0;
    pl.chain_b[0] = pl.chain_b[1] = pl.chain_b[2] = 0;
    pl.idx = key_lo & ~kDup;
    pl.pad = 0;
    uint32_t touched = 0;
    if (!(key_lo & kDup)) {
      const SlowRequest s = slow[key_lo];
      const uint32_t bucket = block_hash(s.x, s.y, s.z, tab.bucket_mask), e0 = bucket << 1;
      const uint32_t k0 = key0(s.x, s.y), k1 = key1(s.z);
      const EntryWords a = load_entry(tab.entries, e0);
      const EntryWords b = load_entry(tab.entries, e0 + 1);
      pl.c_home = tab.claim[bucket];
      pl.k0 = k0;
      pl.k1 = k1;
      pl.rank = s.rank;
      pl.bucket = bucket;
      uint32_t flags = 0, nchain = 0;
      bool present = entry_matches(a, k0, k1) || entry_matches(b, k0, k1);
      uint32_t last = e0 + 1, last_w1 = b.w1;
      int off = entry_offset(b);
      while (off && !present) {
        if (nchain == 3) {
          flags |= kPlanComplex;
          break;
        }
        last = (last + (uint32_t)off) & tab.entry_mask;
        const EntryWords w = load_entry(tab.entries, last);
        if (nchain == 0) pl.chain_b[0] = last >> 1;  // (constant indices: the record stays in registers)
        else if (nchain == 1) pl.chain_b[1] = last >> 1;
        else pl.chain_b[2] = last >> 1;
        ++nchain;
        present = entry_matches(w, k0, k1);
        last_w1 = w.w1;
        off = entry_offset(w);
      }
      if (present) {
        flags = kPlanPresent;
      } else if (!(flags & kPlanComplex)) {
        if (a.idx < 0 || b.idx < 0) {
          flags |= kPlanHome | (a.idx < 0 ? 0u : kPlanHomeSlot1);
        } else {
          pl.c_last = tab.claim[last >> 1];
          // the probe looks at slot 0 of the buckets after the tail's, one by one
          bool found = false;
          uint32_t bn = last >> 1;
          for (uint32_t g = 0; g < kPlanSpan && !found; ++g) {
            bn = (bn + 1) & tab.bucket_mask;
            const int32_t idx = load_entry(tab.entries, bn << 1).idx;
            const uint32_t c = tab.claim[bn];
            if (idx >= 0 || (c != kInf && c < s.rank)) continue;
            found = true;
            pl.c_next = c;
          }
          pl.next = bn << 1;
          flags |= found ? kPlanFound : kPlanComplex;
        }
      }
      pl.last = last;
      pl.last_w1 = last_w1;
      pl.flags = flags | (nchain << 8);
      touched = pl.c_home;
    }
    if (planned) {
      uint4* q = reinterpret_cast<uint4*>(plans + i);
      const uint4* v = reinterpret_cast<const uint4*>(&pl);
      q[0] = v[0];
      q[1] = v[1];
      q[2] = v[2];
      q[3] = v[3];
    }
    asm volatile("" ::"v"(touched));
  }
  __syncthreads();
  RATSDF_STAMP(ctl->stamps, 23);

  // ---- the stores of the pass ----
  // (agent scope: when this role runs inside k_integrate, the committing and carving workgroups of the
  // same launch read these words past their own L2)
  auto write_placed = [&](uint32_t e, uint32_t k0, uint32_t k1, uint32_t rank, uint32_t slot) {
    uint32_t* p = reinterpret_cast<uint32_t*>(tab.entries + e);
    __hip_atomic_store(&p[0], k0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&p[1], k1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // offset 0
    __hip_atomic_store(&p[2], (uint32_t)kPlaceholderIdx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    mark_dirty(tab, e);  // (the commit fills in the pool index; the export lists the entry as it is then)
    if (slot < req_cap) {
      const Request r{(int16_t)(k0 & 0xFFFFu), (int16_t)(k0 >> 16), (int16_t)(k1 & 0xFFFFu),
                      (uint16_t)(kReqWinner | kReqPlaced), rank, e};
      unsigned long long w[2];
      __builtin_memcpy(w, &r, sizeof(r));
      unsigned long long* q = reinterpret_cast<unsigned long long*>(req + slot);
      __hip_atomic_store(q, w[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(q + 1, w[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      set_error(ctl, RATSDF_ERR_CAPACITY);
    }
  };
  auto write_link = [&](uint32_t last, uint32_t last_w1, uint32_t next) {         // :98-99
    const uint32_t wrap = next > last ? 0u : tab.num_entry;
    uint32_t* pl = reinterpret_cast<uint32_t*>(tab.entries + last);
    const int16_t link = (int16_t)(next + wrap - last);
    __hip_atomic_store(&pl[1], (last_w1 & 0xFFFFu) | ((uint32_t)(uint16_t)link << 16), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    mark_dirty(tab, last);  // (the tail's entry has changed: its link)
  };
  // outcome of a request whose plan held: [1:0] 1 = fill the home slot, 2 = link the tail and fill `next`;
  // [2] clear the claim of the first bucket locked (home / tail), [3] of `next`'s; [63:32] request slot
  // (a claim reset is read by the OTHER workgroups of the serial group -- claim_pass, kernels_integrate.h, one
  // per XCD -- in a frame that has both chained requests and thousands of ordinary ones: write-through like
  // every other store of the pass, or a helper would read the old claim from memory and list a leader whose
  // claim this pass has defeated)
  auto clear_claim = [&](uint32_t bucket) {
    __hip_atomic_store(&tab.claim[bucket], kInf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto apply = [&](const SlowPlan& pl, unsigned long long act) {
    const uint32_t kind = (uint32_t)act & 3u, slot = (uint32_t)(act >> 32);
    if (act & 4u) clear_claim((pl.flags & kPlanHome) ? pl.bucket : pl.last >> 1);
    if (act & 8u) clear_claim(pl.next >> 1);
    if (kind == 1u) {
      write_placed((pl.bucket << 1) + ((pl.flags & kPlanHomeSlot1) ? 1u : 0u), pl.k0, pl.k1, pl.rank, slot);
    } else if (kind == 2u) {
      write_link(pl.last, pl.last_w1, pl.next);
      write_placed(pl.next, pl.k0, pl.k1, pl.rank, slot);
    }
  };
  auto load_plan = [&](uint32_t i) -> SlowPlan {
    SlowPlan pl;
    const uint4* q = reinterpret_cast<const uint4*>(plans + i);
    uint4* v = reinterpret_cast<uint4*>(&pl);
    v[0] = q[0];
    v[1] = q[1];
    v[2] = q[2];
    v[3] = q[3];
    return pl;
  };

  // ---- 3. replay ---------------------------------------------------------------------------------
  // The first wave replays, 64 requests at a time, one per lane, in rank order.  What a request decides
  // from its plan is known beforehand (its claims against its rank); what the serial order adds is only
  // this: a request whose plan was read from a bucket that an EARLIER request has locked is stale.  So the
  // wave steps through the batch in order; the request whose turn it is announces the locks it takes
  // (readlane) and every later lane tests them against the buckets its plan was read from -- two readlanes
  // and ten compares per request, no branch, instead of ~300 instructions in a single lane each of which
  // waited for its own LDS round trips.  The outcome stays in the lane; locks and stores follow for the
  // whole batch at once -- or earlier, before a stale request reads the directory and the lock set as they
  // stand (replay_from_memory: the reference's code path, one lane).
  // The pass's only writer of the request list (the frame's ordinary requests were filed by the launch
  // before): the list's counter lives in a register and is published once at the end.
  uint32_t n_x = 0, n_req = 0, n_req_before = 0;
#ifdef RATSDF_STAMPS
  uint32_t n_stale = 0;
#endif
  // Allocate's try-lock of `bucket` at time `time` (voxel_hash.cu:67-70,93-94) given the bucket's claim
  // `c`: fails when the bucket's leader (claim earlier than `time`) or an earlier chained request of
  // the pass holds it; a later leader must find the lock taken (*clear: its claim is to be reset).
  auto try_lock = [&](uint32_t bucket, uint32_t c, uint32_t time, bool* clear) -> bool {
    if (c != kInf && c < time) return false;
    if (lockset_has(locks, bucket)) return false;  // every recorded lock is earlier than `time`
    if (2u * n_x <= locks.mask) {
      lockset_add(locks, bucket);
      ++n_x;
    } else {
      set_error(ctl, RATSDF_ERR_CAPACITY);
    }
    *clear = c != kInf && c > time;
    return true;
  };
  // Allocate for one request, everything read from the directory as it stands now, stores at once
  auto replay_from_memory = [&](const SlowRequest& s, uint32_t* took_a, uint32_t* took_b) {
    const uint32_t time = s.rank;
    const uint32_t bucket = block_hash(s.x, s.y, s.z, tab.bucket_mask);
    const uint32_t e0 = bucket << 1;
    // :48-65 -- present already?  (the walk ends on the chain's tail, which :80-84 needs below)
    const uint32_t k0 = key0(s.x, s.y), k1 = key1(s.z);
    const EntryWords a = load_entry(tab.entries, e0);
    const EntryWords b = load_entry(tab.entries, e0 + 1);
    const uint32_t c_home = tab.claim[bucket];  // (rides with the entries)
    if (entry_matches(a, k0, k1) || entry_matches(b, k0, k1)) return;
    uint32_t last = e0 + 1, last_w1 = b.w1;
    {
      int off = entry_offset(b);
      for (uint32_t g = 0; off && g < tab.num_entry; ++g) {
        last = (last + (uint32_t)off) & tab.entry_mask;
        const EntryWords w = load_entry(tab.entries, last);
        if (entry_matches(w, k0, k1)) return;
        last_w1 = w.w1;
        off = entry_offset(w);
      }
    }
    bool clear = false;
    if (a.idx < 0 || b.idx < 0) {                                        // :67-78
      if (try_lock(bucket, c_home, time, &clear)) {
        *took_a = bucket;
        if (clear) clear_claim(bucket);
        write_placed(e0 + (a.idx < 0 ? 0u : 1u), k0, k1, time, n_req++);
      }
      return;
    }
    const uint32_t bucket_last = last >> 1;                              // :80-84
    const uint32_t c_last = bucket_last == bucket ? c_home : tab.claim[bucket_last];
    uint32_t next = last, c_next = kInf;
    bool found = false;
    for (uint32_t g = 0; g < tab.num_entry && !found; ++g) {            // :86-91
      next = (next + 1) & tab.entry_mask;
      if ((next & 1u) == 1u) continue;  // never the last slot of a bucket
      const int32_t idx = load_entry(tab.entries, next).idx;
      c_next = tab.claim[next >> 1];    // (both requested before either is looked at)
      if (idx >= 0) continue;
      if (c_next != kInf && c_next < time) continue;  // that bucket's leader has filled its slot 0 by now
      found = true;
    }
    if (!found) return;
    // :93-94 (short-circuit &&: the tail's lock is taken, and kept, even when the second one fails).
    // c_last / c_next are what the claims hold now unless the first lock of THIS request changed the
    // second's: only when both buckets are the same one, and then the second try fails on the lock set.
    if (!try_lock(bucket_last, c_last, time, &clear)) return;
    *took_a = bucket_last;
    if (clear) clear_claim(bucket_last);
    if (!try_lock(next >> 1, c_next, time, &clear)) return;
    *took_b = next >> 1;
    if (clear) clear_claim(next >> 1);
    write_link(last, last_w1, next);
    write_placed(next, k0, k1, time, n_req++);
  };
  bool by_wave = planned;  // uniform
  if (planned && !kLds) {  // (the LDS path has counted already) at most kSlowDistinctCap distinct requests
    uint32_t mine = 0;
    for (uint32_t i = tid; i < n; i += nt) mine += !((uint32_t)skeys[i] & kDup);
    uint32_t distinct = 0;
    for (uint32_t r = 0; r < (kSlowPlanCap + 255u) / 256u; ++r) distinct += (uint32_t)__syncthreads_count(mine > r);
    by_wave = distinct <= (uint32_t)kSlowDistinctCap;
  }
  if (by_wave) {
    if (tid < 64) {
      constexpr uint32_t kNone = 0xFFFFFFFFu;
      const uint32_t lane = tid;
      n_req = n_req_before = F->n_req;  // (uniform)
      for (uint32_t base = 0; base < n; base += 64) {  // uniform
        const uint32_t i = base + lane;
        // (what the loop below needs of the plan, in a dozen registers: the record itself is read again
        // when its outcome is applied -- the update kernel this role rides in has none to spare)
        constexpr uint32_t kAbsent = 0xFFFFFFFEu;  // no such chain entry (never equal to a bucket or to kNone)
        uint32_t bucket, cb0, cb1, cb2, lo, span1, lock_a, lock_b, bits, pidx, places;
        bool work, stale;
        {
          SlowPlan pl = load_plan(i < n ? i : base);
          if (i >= n) pl.flags = kPlanDup;
          const uint32_t flags = pl.flags, nchain = (flags >> 8) & 3u;
          work = !(flags & (kPlanDup | kPlanPresent));
          const bool home = (flags & kPlanHome) != 0;
          bucket = pl.bucket;
          cb0 = nchain > 0 ? pl.chain_b[0] : kAbsent;
          cb1 = nchain > 1 ? pl.chain_b[1] : kAbsent;
          cb2 = nchain > 2 ? pl.chain_b[2] : kAbsent;
          pidx = pl.idx;
          // buckets the probe looked at: lo .. lo + span1 - 1 (circular); none: span1 = 0
          lo = ((pl.last >> 1) + 1) & tab.bucket_mask;
          span1 = (flags & kPlanFound) ? (((pl.next >> 1) - lo) & tab.bucket_mask) + 1u : 0u;
          stale = (flags & kPlanComplex) != 0;
          // what the plan does when it holds: try-lock A (home / tail), then B (`next`), voxel_hash.cu:67-70,93-94
          const uint32_t A = home ? pl.bucket : pl.last >> 1, B = pl.next >> 1;
          const uint32_t c_a = home ? pl.c_home : pl.c_last;
          const bool ok_a = work && !(c_a != kInf && c_a < pl.rank);
          const bool ok_b = ok_a && !home && B != A && !(pl.c_next != kInf && pl.c_next < pl.rank);
          lock_a = ok_a ? A : kNone;  // the locks it takes
          lock_b = ok_b ? B : kNone;
          const uint32_t kind = home ? (ok_a ? 1u : 0u) : (ok_b ? 2u : 0u);
          places = kind != 0;
          bits = kind | (ok_a && c_a != kInf && c_a > pl.rank ? 4u : 0u) |
                 (ok_b && pl.c_next != kInf && pl.c_next > pl.rank ? 8u : 0u);
        }
        // was the plan read from bucket x?  (x: a bucket number; no branches: this runs in every step)
        auto reads = [&](uint32_t x) -> bool {
          return (x == bucket) | (x == cb0) | (x == cb1) | (x == cb2) | (((x - lo) & tab.bucket_mask) < span1);
        };
        // stale already?  (locks of the batches before)
        if (work && !stale) {
          stale = lockset_has(locks, bucket);
          if (!stale && cb0 != kAbsent) stale = lockset_has(locks, cb0);
          if (!stale && cb1 != kAbsent) stale = lockset_has(locks, cb1);
          if (!stale && cb2 != kAbsent) stale = lockset_has(locks, cb2);
          for (uint32_t g = 0; g < span1 && !stale; ++g) stale = lockset_has(locks, (lo + g) & tab.bucket_mask);
        }
        // A request whose plan holds keeps its outcome in its lane (`taken`, `slot`); locks and stores
        // follow for the whole batch at once -- or earlier, before a stale request reads the directory
        // and the lock set as they stand.
        bool taken = false;  // this lane's turn has come and its plan held
        uint32_t slot = 0;   // its place in the request list (when it places)
        auto settle = [&]() {
          if (taken && lock_a != kNone) {
            lockset_add(locks, lock_a);
            if (lock_b != kNone) lockset_add(locks, lock_b);
            apply(load_plan(i), bits | ((unsigned long long)slot << 32));
          }
          taken = false;
        };
        unsigned long long todo = __builtin_amdgcn_ballot_w64(work);
#ifdef RATSDF_STAMPS
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (lane == 0) { ctl->stamps[30] -= (unsigned long long)clock64(); ctl->stamps[31] += __popcll(todo); }
#endif
        while (todo) {  // uniform
          const uint32_t j = (uint32_t)__ffsll((long long)todo) - 1u;
          todo &= todo - 1;
          uint32_t la, lb;  // the buckets request j locks (uniform)
          if (__builtin_expect(!((__builtin_amdgcn_ballot_w64(stale) >> j) & 1ull), 1)) {
            la = __builtin_amdgcn_readlane(lock_a, j);
            lb = __builtin_amdgcn_readlane(lock_b, j);
            const bool me = lane == j;
            taken |= me;
            slot = me ? n_req : slot;
            n_req += __builtin_amdgcn_readlane(places, j);
          } else {
#ifdef RATSDF_STAMPS
            if (lane == 0) ++n_stale;
#endif
            settle();
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            uint32_t ta = kNone, tb = kNone;
            if (lane == j) replay_from_memory(slow[pidx], &ta, &tb);
            la = __builtin_amdgcn_readlane(ta, j);
            lb = __builtin_amdgcn_readlane(tb, j);
            n_req = __builtin_amdgcn_readlane(n_req, j);
          }
          // every later request whose plan was read from a bucket locked just now is stale (earlier ones
          // and j itself may set the flag too: their turn has passed)
          bool hit = false;
          if (la != kNone) hit = reads(la);  // uniform
          if (lb != kNone) hit |= reads(lb);
          stale |= hit;
        }
#ifdef RATSDF_STAMPS
        if (lane == 0) ctl->stamps[30] += (unsigned long long)clock64();
#endif
        settle();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the next batch probes the lock set)
      }
      if (lane == 0) {
        if (n_req != n_req_before)
          __hip_atomic_store(&F->n_req, n_req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        RATSDF_STAMP(ctl->stamps, 24);
#ifdef RATSDF_STAMPS
        ctl->stamps[25] += n;
        ctl->stamps[27] += n_stale;
        ctl->stamps[28] += n_req - n_req_before;
        ctl->stamps[29] += 1;
        ctl->stamps[21] += (unsigned long long)clock64();
#endif
      }
    }
    return true;
  }
  // more requests than plans (or than locks): one thread, everything from memory, as in round 2
  if (tid == 0) {
    n_req = n_req_before = F->n_req;
    uint32_t n_d = 0;
    for (uint32_t si = 0; si < n; ++si) {
      const uint32_t key_lo = (uint32_t)skeys[si];
      if (key_lo & kDup) continue;  // same block, later rank: irrelevant
      if (n_d++ >= (uint32_t)kSlowDistinctCap) {
        set_error(ctl, RATSDF_ERR_CAPACITY);
        break;
      }
      uint32_t ta, tb;
      replay_from_memory(slow[key_lo], &ta, &tb);
    }
    if (n_req != n_req_before)
      __hip_atomic_store(&F->n_req, n_req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    RATSDF_STAMP(ctl->stamps, 24);
  }
  return true;
}

// Exclusive scan of one value per thread across the workgroup: wave-level scan with cross-lane
// shuffles, then the (<= 16) wave totals through LDS.  Returns the exclusive prefix; *total receives
// the workgroup sum.  lds: at least block_threads() / 64 words.  Two barriers.
__device__ inline uint32_t block_exclusive_scan(uint32_t v, uint32_t* lds, uint32_t* total) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (block_threads() + 63) >> 6;
  uint32_t x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = __shfl_up(x, o);
    if (lane >= (uint32_t)o) x += t;
  }
  __syncthreads();  // lds may still be in use by a previous scan
  if (lane == 63) lds[wave] = x;
  __syncthreads();
  uint32_t base = 0, tot = 0;
  for (uint32_t i = 0; i < nw; ++i) {
    const uint32_t w = lds[i];
    if (i < wave) base += w;
    tot += w;
  }
  *total = tot;
  return base + x - v;
}

// Rank bitmaps are organised in groups of 32 words (1024 bits) with a summary bitmap (one bit per
// group) so that scans and clean-ups only touch groups that contain set bits.
constexpr uint32_t kGroupWords = 32;

// Words per thread-chunk when `nt` threads scan a bitmap of `nwords` words (whole groups).
__host__ __device__ inline uint32_t bitmap_chunk(uint32_t nwords, uint32_t nt) {
  const uint32_t ngroups = (nwords + kGroupWords - 1) / kGroupWords;
  uint32_t gpt = (ngroups + nt - 1) / nt;
  if (gpt == 0) gpt = 1;
  return gpt * kGroupWords;
}

__device__ inline void bitmap_set(uint32_t* bitmap, uint32_t* summary, uint32_t pos) {
  const uint32_t w = pos >> 5;
  atomicOr(&bitmap[w], 1u << (pos & 31));
  const uint32_t g = w / kGroupWords;
  atomicOr(&summary[g >> 5], 1u << (g & 31));
}

// Popcount of this thread's chunk, skipping groups whose summary bit is clear.  The bits were set
// with atomics earlier in the same kernel by this workgroup; neither the summary nor the bitmap
// words have been read before in this kernel (and the vector L1 is invalidated at kernel start), so
// plain loads after the barrier fetch them from L2.  `bitmap` must be padded to whole groups.
__device__ inline uint32_t chunk_popcount(const uint32_t* bitmap, const uint32_t* summary,
                                          uint32_t nwords, uint32_t chunk) {
  const uint32_t lo = threadIdx.x * chunk;
  uint32_t sum = 0;
  for (uint32_t w = lo; w < lo + chunk && w < nwords; w += kGroupWords) {
    const uint32_t g = w / kGroupWords;
    if (!((summary[g >> 5] >> (g & 31)) & 1u)) continue;
    const uint4* p = reinterpret_cast<const uint4*>(bitmap + w);
    uint4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = p[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) sum += __popc(v[i].x) + __popc(v[i].y) + __popc(v[i].z) + __popc(v[i].w);
  }
  return sum;
}

// Zero every group of `bitmap` whose summary bit is set, then the summary itself (whole workgroup).
__device__ inline void bitmap_clean(uint32_t* bitmap, uint32_t* summary, uint32_t nwords) {
  const uint32_t ngroups = (nwords + kGroupWords - 1) / kGroupWords;
  for (uint32_t g = threadIdx.x; g < ngroups; g += block_threads()) {
    if (!((summary[g >> 5] >> (g & 31)) & 1u)) continue;
    uint4* p = reinterpret_cast<uint4*>(bitmap + g * kGroupWords);
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = make_uint4(0, 0, 0, 0);
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < (ngroups + 31) / 32; i += block_threads()) summary[i] = 0;
}

// Per-word exclusive prefix, written only for groups that contain set bits (the only ones anyone
// looks up): `excl` = number of set bits before this thread's chunk, `sum` = set bits inside it.
__device__ inline void bitmap_write_prefix(const uint32_t* bitmap, const uint32_t* summary,
                                           uint32_t* prefix, uint32_t nwords, uint32_t chunk,
                                           uint32_t sum, uint32_t excl) {
  if (!sum) return;
  const uint32_t lo = threadIdx.x * chunk;
  uint32_t run = excl;
  for (uint32_t w = lo; w < lo + chunk && w < nwords; w += kGroupWords) {
    const uint32_t g = w / kGroupWords;
    if (!((summary[g >> 5] >> (g & 31)) & 1u)) continue;
    const uint4* p = reinterpret_cast<const uint4*>(bitmap + w);
    uint4* q = reinterpret_cast<uint4*>(prefix + w);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint4 v = p[i];
      uint4 o;
      o.x = run; run += __popc(v.x);
      o.y = run; run += __popc(v.y);
      o.z = run; run += __popc(v.z);
      o.w = run; run += __popc(v.w);
      q[i] = o;
    }
  }
}

// rank of set bit `pos` among the set bits (prefix written by bitmap_write_prefix)
__device__ inline uint32_t bitmap_rank(const uint32_t* bitmap, const uint32_t* prefix, uint32_t pos) {
  const uint32_t w = pos >> 5;
  return prefix[w] + __popc(bitmap[w] & ((1u << (pos & 31)) - 1u));
}

// ---------------------------------------------------------------------------------------------
// k_alloc_rank: one workgroup.  resolve -> winners -> rank among winners -> free-list bookkeeping.
// The order of the AquireBlock calls is the raster order of the winners: winner i gets
// req_k[i] = number of winners with a smaller rank.
//   * few requests (the steady state, <= kSmallRank): winners' ranks go to an LDS list and every
//     winner counts the smaller ones -- no global atomics, no bitmap
//   * many requests (first frames of a scene): rank-indexed bitmap + popcount prefix (self-cleaning)
// ---------------------------------------------------------------------------------------------
// Rank of every key of an LDS list among the list's (distinct) keys = number of smaller keys,
// handed to emit(position, rank).  Four lanes share a key and each scans a quarter of the list with
// 16-byte LDS reads (a scalar loop over a few hundred keys per thread is a chain of LDS latencies and
// was most of the serial kernels' time).  keys[] must have room for 16 entries of padding.
// All threads of the workgroup; two barriers.
template <typename Emit>
__device__ inline void lds_rank_all(uint32_t* keys, uint32_t n, Emit emit) {
  const uint32_t tid = threadIdx.x, nt = block_threads();
  const uint32_t n16 = (n + 15u) & ~15u;
  __syncthreads();
  if (tid < n16 - n) keys[n + tid] = kInf;
  __syncthreads();
  const uint4* v = reinterpret_cast<const uint4*>(keys);
  const uint32_t nchunks = n16 >> 2, sub = tid & 3u;
  for (uint32_t base = 0; base < n; base += nt >> 2) {  // uniform
    const uint32_t item = base + (tid >> 2);
    const uint32_t mine = item < n ? keys[item] : 0u;
    uint32_t k = 0;
#pragma unroll 4
    for (uint32_t c = sub; c < nchunks; c += 4) {
      const uint4 x = v[c];
      k += (x.x < mine) + (x.y < mine) + (x.z < mine) + (x.w < mine);
    }
    k += __shfl_xor(k, 1);
    k += __shfl_xor(k, 2);
    if (sub == 0 && item < n) emit(item, k);
  }
}

constexpr uint32_t kSmallRank = 4096;  // LDS list capacity (16 KiB of the resolver's sort buffer)
// The serial role inside k_integrate (kernels_integrate.h) lists the winners' ranks in device memory
// (RankBufs::win_ranks) and every committing wave counts the smaller ones itself, so its ordinary path
// is not limited by LDS: it takes frames of up to kFusedRank requests -- a whole new view at 1280x720 /
// 2 mm files ~30 k -- which until round 3 fell into the general path (a rank-indexed bitmap scanned by
// one 256-thread workgroup with its scratch in device memory while 2 048 workgroups poll: 0.8-1.4 ms).
constexpr uint32_t kFusedRank = 32768;

// All threads of one workgroup; `skeys` = the workgroup's dynamic LDS (kSerialLdsBytes), `nf` = the
// pool's free count at the start of this pass.
__device__ inline void alloc_rank_role(const Table& tab, Request* req, uint32_t req_cap,
                                       uint32_t* req_k, const SlowRequest* slow, uint32_t slow_cap,
                                       XLock* xlocks, uint32_t* bitmap,
                                       uint32_t* summary, uint32_t* prefix, uint32_t nwords,
                                       unsigned long long* sort_scratch, Ctl* ctl, FrameCtl* F,
                                       int32_t nf, unsigned long long* skeys, bool resolved = false) {
  uint32_t* lds = reinterpret_cast<uint32_t*>(skeys);  // [0,32): scan scratch, [32]: counter
  uint32_t* lds_rank = lds + 64;                       // kSmallRank words
  const uint32_t tid = threadIdx.x, nt = block_threads();
  RATSDF_STAMP(ctl->stamps, 8);
  const uint32_t n_slow = F->n_slow;
  if (n_slow != 0 && !resolved) {  // uniform
    (void)resolve_slow_requests<false>(tab, req, req_cap, slow, slow_cap, xlocks, ctl, F, skeys, sort_scratch,
                                       (uint32_t)kSlowLdsCap);
    __syncthreads();
  }
  uint32_t n = n_slow ? ld_agent_u32(&F->n_req) : F->n_req;  // the resolver appends requests
  if (n > req_cap) n = req_cap;
  RATSDF_STAMP(ctl->stamps, 12);
  uint32_t total = 0;
  if (n <= kSmallRank) {
    uint32_t* lds_req = lds_rank + kSmallRank + 16;  // request index of each listed winner
    if (tid == 0) lds[32] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < n; i += nt) {
      const Request r = req[i];
      bool win = (r.flags & kReqWinner) != 0;
      if (!(r.flags & kReqPlaced)) {
        const uint32_t bucket = block_hash(r.x, r.y, r.z, tab.bucket_mask);
        win = tab.claim[bucket] == r.rank;
        if (win) {
          req[i].flags = kReqWinner;
          mark_dirty(tab, r.entry);  // (directory delta: the entry the commit will fill)
        }
      }
      if (win) {
        const uint32_t slot = atomicAdd(&lds[32], 1u);
        lds_rank[slot] = r.rank;
        lds_req[slot] = i;
      }
    }
    __syncthreads();
    RATSDF_STAMP(ctl->stamps, 9);
    total = lds[32];
    // winner w: count the winners with smaller rank
    lds_rank_all(lds_rank, total, [&](uint32_t w, uint32_t k) { req_k[lds_req[w]] = k; });
    RATSDF_STAMP(ctl->stamps, 10);
  } else {
    // Thousands of requests (the first frames of a view) on one workgroup: every request costs two
    // DEPENDENT loads (the request, then its bucket's claim), so the loops below keep kU requests per
    // thread in flight -- taken one at a time, 30 k requests on 256 threads were ~230 back-to-back
    // memory round trips and most of the 0.85 ms such a frame took in round 2.
    constexpr int kU = 4;
    for (uint32_t base = tid; base < n; base += nt * kU) {  // uniform
      Request r[kU];
      uint32_t c[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const uint32_t i = base + (uint32_t)u * nt;
        r[u] = req[i < n ? i : 0];
      }
#pragma unroll
      for (int u = 0; u < kU; ++u)
        c[u] = tab.claim[block_hash(r[u].x, r[u].y, r[u].z, tab.bucket_mask)];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const uint32_t i = base + (uint32_t)u * nt;
        if (i >= n) continue;
        bool win = (r[u].flags & kReqWinner) != 0;
        if (!(r[u].flags & kReqPlaced)) {
          win = c[u] == r[u].rank;
          if (win) {
            req[i].flags = kReqWinner;
            mark_dirty(tab, r[u].entry);
          }
        }
        if (win) bitmap_set(bitmap, summary, r[u].rank);
      }
    }
    __syncthreads();
    RATSDF_STAMP(ctl->stamps, 9);
    const uint32_t chunk = bitmap_chunk(nwords, nt);
    const uint32_t sum = chunk_popcount(bitmap, summary, nwords, chunk);
    const uint32_t excl = block_exclusive_scan(sum, lds, &total);
    bitmap_write_prefix(bitmap, summary, prefix, nwords, chunk, sum, excl);
    __syncthreads();
    for (uint32_t base = tid; base < n; base += nt * kU) {  // uniform
      Request r[kU];
      uint32_t pw[kU], bw[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const uint32_t i = base + (uint32_t)u * nt;
        r[u] = req[i < n ? i : 0];
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {  // bitmap_rank's two loads (rank < 32 * nwords by construction)
        const uint32_t w = r[u].rank >> 5;
        const bool need = (r[u].flags & kReqWinner) && w < nwords;
        pw[u] = prefix[need ? w : 0];
        bw[u] = bitmap[need ? w : 0];
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const uint32_t i = base + (uint32_t)u * nt;
        if (i < n && (r[u].flags & kReqWinner))
          req_k[i] = pw[u] + __popc(bw[u] & ((1u << (r[u].rank & 31)) - 1u));
      }
    }
    __syncthreads();
    bitmap_clean(bitmap, summary, nwords);
    RATSDF_STAMP(ctl->stamps, 10);
  }
  if (tid == 0) {
    uint32_t take = total;
    if ((int64_t)total > (int64_t)nf) {  // voxel_mem.cu:39 assert(idx >= 1)
      set_error(ctl, RATSDF_ERR_POOL_EXHAUSTED);
      take = (uint32_t)nf;
    }
    F->alloc_base = (uint32_t)nf;
    F->n_win = take;
    F->pending = 1;  // the frame (or test pass) now owes a carve_finalize
    ctl->num_free = nf - (int32_t)take;
    atomicMin(&ctl->free_low, nf - (int32_t)take);  // (Table::active: the slots ever in use; no reply awaited)
  }
  RATSDF_STAMP(ctl->stamps, 11);
}

// Commit of one allocation request by one wave (all lanes call it; only lane 0 of a wave with
// `writer` set touches the directory).  A winner with rank k among the winners takes
// heap[alloc_base - 1 - k] (AquireBlock, voxel_mem.cu:37-41) and gets its directory entry written
// (voxel_hash.cu:72-74,101-103); every request releases its bucket's claim (ResetLocks).
// Returns true for a winner that received a block; *out_k / *out_idx / *out_entry describe it.
// kMark: the entry's dirty bit for the directory delta is set here (stand-alone commit).  Inside k_integrate the
// frame's serial role marks every winner's entry instead (claim_pass / write_placed): the update loop has no
// register to spare for one more rarely-taken atomic.
template <bool kMark = false>
__device__ inline bool commit_request(const Table& tab, const Pool& pool, const Request& r,
                                      uint32_t k, uint32_t alloc_base, uint32_t n_win, bool writer,
                                      int32_t* out_idx, uint32_t* out_entry) {
  const uint32_t bucket = block_hash(r.x, r.y, r.z, tab.bucket_mask);
  const bool placed = (r.flags & kReqPlaced) != 0;
  if (!(r.flags & kReqWinner)) {
    if (writer && !placed) tab.claim[bucket] = kInf;
    return false;
  }
  const uint32_t e = r.entry;  // chosen when the request was filed / placed
  uint32_t* pe = reinterpret_cast<uint32_t*>(tab.entries + e);
  if (k >= n_win) {  // pool exhausted (voxel_mem.cu:39): this insertion does not happen
    if (writer) {
      if (placed) pe[2] = (uint32_t)-1;
      else tab.claim[bucket] = kInf;
    }
    return false;
  }
  // agent-scope load: in a fused frame the serial role may have pushed this index moments ago
  const int32_t idx = __hip_atomic_load(&pool.heap[alloc_base - 1 - k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (writer) {
    if (!placed) {
      pe[0] = key0(r.x, r.y);
      pe[1] = key1(r.z);
      tab.claim[bucket] = kInf;
    }
    pe[2] = (uint32_t)idx;
    atomicOr(&tab.occ[e >> 6], 1ull << (e & 63));
    if (kMark) mark_dirty(tab, e);
    reinterpret_cast<uint4*>(tab.active)[idx] = make_uint4(key0(r.x, r.y), key1(r.z), (uint32_t)idx, e);
  }
  *out_idx = idx;
  *out_entry = e;
  return true;
}

// Stand-alone commit (test hook: allocation passes without a frame): one wave per request, block
// initialised to weight 1 / tsdf -1 / probability .5 with rgb left untouched (voxel_mem.cu:43-51).
__global__ __launch_bounds__(256) void k_commit_only(Table tab, Pool pool, const Request* req,
                                                     uint32_t req_cap, const uint32_t* req_k,
                                                     const uint32_t* win_ranks, Ctl* ctl,
                                                     uint32_t par) {
  const FrameCtl* F = &ctl->fr[par];
  uint32_t n = F->n_req;
  if (n > req_cap) n = req_cap;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * block_threads() + threadIdx.x) >> 6;
  const uint32_t nwaves = (gridDim.x * block_threads()) >> 6;
  const uint32_t base = F->alloc_base, n_win = F->n_win, n_winlist = F->n_winlist;
  for (uint32_t i = wave; i < n; i += nwaves) {
    const Request r = req[i];
    uint32_t e;
    int32_t idx;
    uint32_t k = 0;
    if (r.flags & kReqWinner) {
      if (n_winlist) {  // few winners: position in raster order = winners with a smaller rank
        for (uint32_t j = lane; j < n_winlist; j += 64) k += win_ranks[j] < r.rank;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) k += __shfl_xor(k, o);
      } else {
        k = req_k[i];
      }
    }
    if (!commit_request<true>(tab, pool, r, k, base, n_win, lane == 0, &idx, &e)) continue;
    const size_t v = ((size_t)idx << 9) + lane * 8;
    float4* pt = reinterpret_cast<float4*>(pool.tsdf + v);
    float4* ps = reinterpret_cast<float4*>(pool.segm + v);
    uint4* pc = reinterpret_cast<uint4*>(pool.rgbw + v);
    const float4 m1 = make_float4(-1.f, -1.f, -1.f, -1.f);
    const float4 hf = make_float4(.5f, .5f, .5f, .5f);
    pt[0] = m1;
    pt[1] = m1;
    ps[0] = hf;
    ps[1] = hf;
    uint4 c0 = pc[0], c1 = pc[1];
    const uint32_t keep = 0x00FFFFFFu, one = 0x01000000u;
    c0.x = (c0.x & keep) | one; c0.y = (c0.y & keep) | one;
    c0.z = (c0.z & keep) | one; c0.w = (c0.w & keep) | one;
    c1.x = (c1.x & keep) | one; c1.y = (c1.y & keep) | one;
    c1.z = (c1.z & keep) | one; c1.w = (c1.w & keep) | one;
    pc[0] = c0;
    pc[1] = c1;
  }
}

}  // namespace ratsdf
