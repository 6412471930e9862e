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
//   alloc_pixels_role  per pixel: candidates, directory lookup, atomicMin(claim[bucket], rank) for
//                      ordinary buckets ("first requester in raster order wins the bucket lock"),
//                      append to a small slow list for chained / full buckets (part of k_front)
//   k_alloc_rank       one workgroup: (a) replays the slow list in rank order against the claim
//                      table (time-dependent lock / fill queries), so chain appends lock, link and
//                      defeat later claims exactly as the sequential code would; (b) winners
//                      (claim == own rank) set their bit in a rank-indexed bitmap; (c) popcount
//                      prefix of the bitmap = order of the AquireBlock calls
//   commit_request     (inside k_integrate, or k_commit_only for the test hook) one wave per winner:
//                      pool index heap[free-1-k], directory entry, occupancy bit
#pragma once
#include "device_math.h"

namespace ratsdf {

__device__ inline void set_error(Ctl* ctl, uint32_t code) { atomicCAS(&ctl->error, 0u, code); }
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
                                     Ctl* ctl) {
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
      const uint32_t slot = atomicAdd(&ctl->n_req, 1u);
      if (slot < req_cap) {
        req[slot] = Request{(int16_t)x, (int16_t)y, (int16_t)z, 0, rank, 0};
      } else {
        set_error(ctl, RATSDF_ERR_CAPACITY);
      }
    }
  } else {
    const uint32_t slot = atomicAdd(&ctl->n_slow, 1u);
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

// alloc_request for a block already known to be absent, home entries already loaded.
__device__ inline void alloc_request_absent(const Table& t, int x, int y, int z, uint32_t rank,
                                            const EntryWords& a, const EntryWords& b, Request* req,
                                            uint32_t req_cap, SlowRequest* slow, uint32_t slow_cap,
                                            Ctl* ctl) {
  const uint32_t bucket = block_hash(x, y, z, t.bucket_mask);
  const bool special = (a.idx >= 0 && b.idx >= 0) || entry_offset(b) != 0;
  if (!special) {
    const uint32_t old = atomicMin(&t.claim[bucket], rank);
    if (rank < old) {
      const uint32_t slot = atomicAdd(&ctl->n_req, 1u);
      if (slot < req_cap) {
        req[slot] = Request{(int16_t)x, (int16_t)y, (int16_t)z, 0, rank, 0};
      } else {
        set_error(ctl, RATSDF_ERR_CAPACITY);
      }
    }
  } else {
    const uint32_t slot = atomicAdd(&ctl->n_slow, 1u);
    if (slot < slow_cap) {
      slow[slot] = SlowRequest{(int16_t)x, (int16_t)y, (int16_t)z, 0, rank};
    } else {
      set_error(ctl, RATSDF_ERR_CAPACITY);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// alloc_pixels_role: block_allocate_kernel, voxel_tsdf.cu:120-168.  One lane per pixel, 64 consecutive
// pixels of a row per wave (coalesced depth / ht / lt reads).  Also writes the packed per-pixel
// texels the integration kernel gathers from: texA = {depth, range, log ht, log lt},
// texB = {rgb, w_new}.  log(ht), log(lt) and w_new = (1 - d/max_depth)*4 are functions of the pixel
// only (voxel_tsdf.cu:226,243,246), so evaluating them once per pixel instead of once per voxel is
// value-identical.
// ---------------------------------------------------------------------------------------------
__device__ inline void alloc_pixels_role(const Table& tab, const FrameParams& P, uint32_t wg,
                                         const float* depth, const uint8_t* rgb, const float* ht,
                                         const float* lt, float4* texA, uint2* texB, Request* req,
                                         uint32_t req_cap, SlowRequest* slow, uint32_t slow_cap,
                                         Ctl* ctl) {
#ifdef RATSDF_STAMPS
  unsigned long long* ws = (ctl->debug_buf && P.debug == 8) ? ctl->debug_buf + (size_t)((wg * 4 + (threadIdx.x >> 6)) & 16383) * 8 : nullptr;
  if (ws && (threadIdx.x & 63) == 0) { ws[0] = (unsigned long long)clock64(); ws[5] = wall_clock64(); }
#define PSTAMP(i) do { if (ws && (threadIdx.x & 63) == 0) ws[i] = (unsigned long long)clock64(); } while (0)
#else
#define PSTAMP(i) do { } while (0)
#endif
  const int npix = P.W * P.H;
  const int pix = (int)wg * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool inb = pix < npix;
  const int px = inb ? pix % P.W : 0;
  const int py = inb ? pix / P.W : 0;
  const float d = inb ? depth[pix] : 0.f;

  const V3 pimg{(float)px, (float)py, 1.f};
  const V3 pc = intr_mul(P.Ki, pimg);                                   // :137
  const float r = sqrtf(pc.x * pc.x + (pc.y * pc.y + pc.z * pc.z));     // :140 (Eigen norm order)
  if (inb) {
    float lh = 0.f, ll = 0.f;
    if (P.has_sem) {
      lh = __logf(ht[pix]);  // same function as the per-voxel log in k_integrate
      ll = __logf(lt[pix]);
    }
    const float wn = (1 - d / P.md) * 4;
    const uint32_t c = (uint32_t)rgb[3 * pix] | ((uint32_t)rgb[3 * pix + 1] << 8) |
                       ((uint32_t)rgb[3 * pix + 2] << 16);
    texA[pix] = make_float4(d, r, lh, ll);
    texB[pix] = make_uint2(c, __float_as_uint(wn));
  }
  const bool valid = inb && !(d == 0 || d > P.md);                      // :141
  PSTAMP(1);

  const V3 pcd{pc.x * d, pc.y * d, pc.z * d};
  const V3 pw = se3_apply(P.Ti, pcd);                                   // :146
  // shared-divisor divisions (device_math.h): r in [1, ~3], voxel size a frame constant
  const Recip rr = make_recip(r), rvs = make_recip(P.vs);
  const bool vs_ok = recip_safe(P.vs);  // uniform
  const V3 dc{div_shared(pc.x, rr), div_shared(pc.y, rr), div_shared(pc.z, rr)};     // :148
  const V3 dw = quat_rotate(P.Ti.q, dc);                                // :150
  const V3 sw{pw.x - dw.x * P.trunc, pw.y - dw.y * P.trunc, pw.z - dw.z * P.trunc};  // :151
  V3 dg, sg;
  if (vs_ok && fabsf(sw.x) < 1e18f && fabsf(sw.y) < 1e18f && fabsf(sw.z) < 1e18f) {
    dg = V3{div_shared(dw.x, rvs), div_shared(dw.y, rvs), div_shared(dw.z, rvs)};     // :153
    sg = V3{div_shared(sw.x, rvs), div_shared(sw.y, rvs), div_shared(sw.z, rvs)};     // :154
  } else {
    dg = V3{dw.x / P.vs, dw.y / P.vs, dw.z / P.vs};
    sg = V3{sw.x / P.vs, sw.y / P.vs, sw.z / P.vs};
  }
  const float two_tr = 2 * P.trunc;
  const V3 rg{two_tr * dg.x, two_tr * dg.y, two_tr * dg.z};             // :155
  int steps = f2i(ceilf(fmaxf(fmaxf(fabsf(rg.x), fabsf(rg.y)), fabsf(rg.z)) / RATSDF_BLOCK_LEN));
  const float den = fmaxf((float)steps, 1);
  // :159 -- dividing by 1, 2, 4, ... is an exact scaling, so multiply by the exact reciprocal then
  V3 st;
  if (steps <= 2 || (steps & (steps - 1)) == 0) {
    const float inv = 1.f / den;  // exact for powers of two
    st = V3{rg.x * inv, rg.y * inv, rg.z * inv};
  } else {
    st = V3{rg.x / den, rg.y / den, rg.z / den};
  }
  if (valid && steps >= P.S) {  // cannot happen for |dir| <= 1; keep ranks unique regardless
    set_error(ctl, RATSDF_ERR_CAPACITY);
    steps = P.S - 1;
  }
  V3 p = sg;
  uint32_t prev0 = kInf, prev1 = kInf;  // this lane's previous sample
  if (P.debug == 1) return;
  if (P.S <= 4) {
    // Batched form (S = 3 for truncation = 6 voxels): derive all candidate blocks first, then put
    // every directory lookup of the lane in flight at once, then evaluate.  Same requests, same
    // ranks as the sequential form below; only the memory latency overlaps.
    int bxs[4], bys[4], bzs[4];
    bool need[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      need[i] = false;
      bxs[i] = bys[i] = bzs[i] = 0;
      if (i < P.S) {  // uniform
        const bool act = valid && i <= steps;
        const int gx = (int16_t)f2i(roundf(p.x)), gy = (int16_t)f2i(roundf(p.y)),
                  gz = (int16_t)f2i(roundf(p.z));                         // :163-164
        const int bx = gx >> 3, by = gy >> 3, bz = gz >> 3;
        const uint32_t k0 = act ? key0(bx, by) : kInf;
        const uint32_t k1 = act ? key1(bz) : kInf;
        const uint32_t n0 = __shfl_up(k0, 1), n1 = __shfl_up(k1, 1);
        const bool dup = (k0 == prev0 && k1 == prev1) || (lane > 0 && k0 == n0 && k1 == n1);
        need[i] = act && !dup && shard_owned(bx, P) && P.debug != 2;
        bxs[i] = bx;
        bys[i] = by;
        bzs[i] = bz;
        if (act) {
          prev0 = k0;
          prev1 = k1;
        }
        p.x += st.x;
        p.y += st.y;
        p.z += st.z;
      }
    }
    PSTAMP(2);
    EntryWords ea[4], eb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ea[i] = EntryWords{0, 0, -1};
      eb[i] = EntryWords{0, 0, -1};
      if (need[i]) {
        const uint32_t e0 = block_hash(bxs[i], bys[i], bzs[i], tab.bucket_mask) << 1;
        ea[i] = load_entry(tab.entries, e0);
        eb[i] = load_entry(tab.entries, e0 + 1);
      }
    }
#ifdef RATSDF_STAMPS
    if (ws) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    PSTAMP(3);
    // lookup first (cheap, usually a hit); the 8-corner frustum test only for absent blocks (both are
    // pure predicates; the reference tests visibility first, :165-166)
    bool absent[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      absent[i] = false;
      if (need[i]) {
        const uint32_t e0 = block_hash(bxs[i], bys[i], bzs[i], tab.bucket_mask) << 1;
        absent[i] = !block_present_pre(tab, key0(bxs[i], bys[i]), key1(bzs[i]), e0, ea[i], eb[i]);
      }
    }
    // Frustum test of the wave's absent candidates, spread over the lanes: 8 candidates x 8 corners
    // per step (is_block_visible<true>, voxel_tsdf.cu:75-96).  A lane that owns an absent candidate
    // would otherwise walk its 8 corners alone while the other 63 lanes idle -- those waves (image
    // border, freshly carved blocks) were the tail of this kernel.
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i >= P.S) break;  // uniform
      unsigned long long todo = __ballot(absent[i]);
      unsigned long long ok = 0;
      while (todo) {  // uniform
        unsigned long long packed = 0;  // up to 8 owner lanes, one byte each, 0xFF = none
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          unsigned long long o = 0xFFull;
          if (todo) {
            o = (unsigned long long)(__ffsll((long long)todo) - 1);
            todo &= todo - 1;
          }
          packed |= o << (8 * k);
        }
        const uint32_t own = (uint32_t)(packed >> (8 * (lane >> 3))) & 0xFFu;
        const int src = own == 0xFFu ? 0 : (int)own;
        const int cbx = __shfl(bxs[i], src), cby = __shfl(bys[i], src), cbz = __shfl(bzs[i], src);
        const int c = lane & 7;
        const int cx = (int16_t)((int16_t)(cbx << 3) + ((c >> 0) & 1) * 7);
        const int cy = (int16_t)((int16_t)(cby << 3) + ((c >> 1) & 1) * 7);
        const int cz = (int16_t)((int16_t)(cbz << 3) + ((c >> 2) & 1) * 7);
        const bool v = own == 0xFFu || voxel_visible(cx, cy, cz, P);
        const unsigned long long b = __ballot(v);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const unsigned long long o = (packed >> (8 * k)) & 0xFFull;
          if (o != 0xFFull && ((b >> (8 * k)) & 0xFFull) == 0xFFull) ok |= 1ull << o;
        }
      }
      if (absent[i] && ((ok >> lane) & 1ull)) {
        alloc_request_absent(tab, bxs[i], bys[i], bzs[i], (uint32_t)pix * (uint32_t)P.S + (uint32_t)i,
                             ea[i], eb[i], req, req_cap, slow, slow_cap, ctl);
      }
    }
#ifdef RATSDF_STAMPS
    if (ws && (threadIdx.x & 63) == 0) { ws[4] = (unsigned long long)clock64(); ws[6] = wall_clock64(); }
#endif
    return;
  }
  for (int i = 0; i < P.S; ++i) {       // uniform trip count: the shuffles below need every lane
    const bool act = valid && i <= steps;
    const int gx = (int16_t)f2i(roundf(p.x)), gy = (int16_t)f2i(roundf(p.y)),
              gz = (int16_t)f2i(roundf(p.z));                           // :163-164
    const int bx = gx >> 3, by = gy >> 3, bz = gz >> 3;
    const uint32_t k0 = act ? key0(bx, by) : kInf;
    const uint32_t k1 = act ? key1(bz) : kInf;
    // A later request for the same block never matters (the earlier one either inserts it or fails
    // on a lock that stays taken), so drop repeats of the previous sample / the previous pixel.
    const uint32_t n0 = __shfl_up(k0, 1), n1 = __shfl_up(k1, 1);
    const bool dup = (k0 == prev0 && k1 == prev1) || (lane > 0 && k0 == n0 && k1 == n1);
    if (act && !dup && shard_owned(bx, P) && P.debug != 2) {
      EntryWords w;
      if (find_block(tab, bx, by, bz, &w) == kInf && block_visible<true>(bx, by, bz, P)) {
        alloc_request(tab, bx, by, bz, (uint32_t)pix * (uint32_t)P.S + (uint32_t)i, req, req_cap,
                      slow, slow_cap, ctl);
      }
    }
    if (act) {
      prev0 = k0;
      prev1 = k1;
    }
    p.x += st.x;
    p.y += st.y;
    p.z += st.z;
  }
}

// test hook: explicit request list, rank = list index (utils/tests/voxel_hash_test.cu:36-39)
__global__ void k_alloc_list(Table tab, FrameParams P, const int16_t* pos, int n, Request* req,
                             uint32_t req_cap, SlowRequest* slow, uint32_t slow_cap, Ctl* ctl) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int x = pos[3 * i], y = pos[3 * i + 1], z = pos[3 * i + 2];
  if (!shard_owned(x, P)) return;
  alloc_request(tab, x, y, z, (uint32_t)i, req, req_cap, slow, slow_cap, ctl);
}

// ---------------------------------------------------------------------------------------------
// Exact rank-ordered replay of VoxelHashTable::Allocate (voxel_hash.cu:46-108) for requests whose
// home bucket is full or heads a chain.  Runs in ONE workgroup; thread 0 does the serial part after
// an LDS bitonic sort of the slow list by rank.  Everything an ordinary bucket does during the pass
// is summarised by its claim (min rank): bucket x is locked from time claim[x] on, and its leader
// fills the first empty home entry at that time unless an earlier slow request locked x first.
// ---------------------------------------------------------------------------------------------
constexpr int kSlowSortCap = 16384;   // slow requests per pass (LDS bitonic sort, 128 KiB)
constexpr int kSlowDistinctCap = 1024;
constexpr int kXLockCap = 2048;

struct XLock {
  uint32_t bucket, time;
};

__device__ inline void resolve_slow_requests(const Table& tab, Request* req, uint32_t req_cap,
                                             const SlowRequest* slow, uint32_t slow_cap,
                                             XLock* xlocks, SlowRequest* distinct, Ctl* ctl,
                                             unsigned long long* skeys) {
  uint32_t n = ctl->n_slow;
  if (n > slow_cap) n = slow_cap;
  if (n > (uint32_t)kSlowSortCap) {
    if (threadIdx.x == 0) set_error(ctl, RATSDF_ERR_CAPACITY);
    n = kSlowSortCap;
  }
  uint32_t m = 1;
  while (m < n) m <<= 1;
  for (uint32_t i = threadIdx.x; i < m; i += blockDim.x)
    skeys[i] = i < n ? (((unsigned long long)slow[i].rank << 32) | i) : ~0ull;
  __syncthreads();
  for (uint32_t k = 2; k <= m; k <<= 1) {
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) {
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
  if (threadIdx.x != 0) return;

  uint32_t n_x = 0, n_d = 0;
  auto locked_at = [&](uint32_t bucket, uint32_t time) -> bool {
    const uint32_t c = tab.claim[bucket];
    if (c != kInf && c < time) return true;
    for (uint32_t i = 0; i < n_x; ++i)
      if (xlocks[i].bucket == bucket) return true;  // every recorded lock is earlier than `time`
    return false;
  };
  auto take_lock = [&](uint32_t bucket, uint32_t time) {
    if (n_x < (uint32_t)kXLockCap) {
      xlocks[n_x++] = XLock{bucket, time};
    } else {
      set_error(ctl, RATSDF_ERR_CAPACITY);
    }
    const uint32_t c = tab.claim[bucket];
    if (c != kInf && c > time) tab.claim[bucket] = kInf;  // a later leader finds the lock taken
  };
  auto place = [&](uint32_t e, const SlowRequest& s) {
    uint32_t* p = reinterpret_cast<uint32_t*>(tab.entries + e);
    p[0] = key0(s.x, s.y);
    p[1] = key1(s.z);  // offset 0
    p[2] = (uint32_t)kPlaceholderIdx;
    const uint32_t slot = atomicAdd(&ctl->n_req, 1u);
    if (slot < req_cap) {
      req[slot] = Request{s.x, s.y, s.z, (uint16_t)(kReqWinner | kReqPlaced), s.rank, e};
    } else {
      set_error(ctl, RATSDF_ERR_CAPACITY);
    }
  };

  for (uint32_t si = 0; si < n; ++si) {
    const SlowRequest s = slow[(uint32_t)(skeys[si] & 0xFFFFFFFFu)];
    bool seen = false;
    for (uint32_t i = 0; i < n_d && !seen; ++i)
      seen = distinct[i].x == s.x && distinct[i].y == s.y && distinct[i].z == s.z;
    if (seen) continue;  // same block, later rank: irrelevant
    if (n_d < (uint32_t)kSlowDistinctCap) {
      distinct[n_d++] = s;
    } else {
      set_error(ctl, RATSDF_ERR_CAPACITY);
      break;
    }
    const uint32_t time = s.rank;
    const uint32_t bucket = block_hash(s.x, s.y, s.z, tab.bucket_mask);
    const uint32_t e0 = bucket << 1;
    EntryWords w;
    if (find_block(tab, s.x, s.y, s.z, &w) != kInf) continue;          // :48-65
    bool handled = false;
    for (uint32_t i = 0; i < 2 && !handled; ++i) {                      // :67-78
      if (load_entry(tab.entries, e0 + i).idx < 0) {
        if (!locked_at(bucket, time)) {
          take_lock(bucket, time);
          place(e0 + i, s);
        }
        handled = true;
      }
    }
    if (handled) continue;
    uint32_t last = e0 + 1;                                             // :80-84
    for (uint32_t g = 0; g < tab.num_entry; ++g) {
      const int off = entry_offset(load_entry(tab.entries, last));
      if (!off) break;
      last = (last + (uint32_t)off) & tab.entry_mask;
    }
    const uint32_t bucket_last = last >> 1;
    uint32_t next = last;
    bool found = false;
    for (uint32_t g = 0; g < tab.num_entry && !found; ++g) {            // :86-91
      next = (next + 1) & tab.entry_mask;
      if ((next & 1u) == 1u) continue;  // never the last slot of a bucket
      if (load_entry(tab.entries, next).idx >= 0) continue;
      const uint32_t c = tab.claim[next >> 1];
      if (c != kInf && c < time) continue;  // that bucket's leader has filled its slot 0 by now
      found = true;
    }
    if (!found) continue;
    const uint32_t bucket_next = next >> 1;
    if (!locked_at(bucket_last, time)) {                                // :93-94 (short-circuit &&)
      take_lock(bucket_last, time);
      if (!locked_at(bucket_next, time)) {
        take_lock(bucket_next, time);
        const uint32_t wrap = next > last ? 0u : tab.num_entry;
        uint32_t* pl = reinterpret_cast<uint32_t*>(tab.entries + last);
        const int16_t link = (int16_t)(next + wrap - last);             // :98-99
        pl[1] = (pl[1] & 0xFFFFu) | ((uint32_t)(uint16_t)link << 16);
        place(next, s);
      }
    }
  }
}

// Exclusive scan of one value per thread across the workgroup: wave-level scan with cross-lane
// shuffles, then the (<= 16) wave totals through LDS.  Returns the exclusive prefix; *total receives
// the workgroup sum.  lds: at least blockDim.x / 64 words.  Two barriers.
__device__ inline uint32_t block_exclusive_scan(uint32_t v, uint32_t* lds, uint32_t* total) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
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
  for (uint32_t g = threadIdx.x; g < ngroups; g += blockDim.x) {
    if (!((summary[g >> 5] >> (g & 31)) & 1u)) continue;
    uint4* p = reinterpret_cast<uint4*>(bitmap + g * kGroupWords);
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = make_uint4(0, 0, 0, 0);
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < (ngroups + 31) / 32; i += blockDim.x) summary[i] = 0;
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
constexpr uint32_t kSmallRank = 4096;  // LDS list capacity (16 KiB of the resolver's sort buffer)

__global__ __launch_bounds__(1024) void k_alloc_rank(Table tab, Request* req, uint32_t req_cap,
                                                     uint32_t* req_k, const SlowRequest* slow,
                                                     uint32_t slow_cap, XLock* xlocks,
                                                     SlowRequest* distinct, uint32_t* bitmap,
                                                     uint32_t* summary, uint32_t* prefix,
                                                     uint32_t nwords, Ctl* ctl) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long skeys[];
  uint32_t* lds = reinterpret_cast<uint32_t*>(skeys);  // [0,32): scan scratch, [32]: counter
  uint32_t* lds_rank = lds + 64;                       // kSmallRank words
  const uint32_t tid = threadIdx.x, nt = blockDim.x;
  RATSDF_STAMP(ctl->stamps, 8);
  const uint32_t n_slow = ctl->n_slow;
  const int32_t nf = ctl->num_free;
  if (n_slow != 0) {  // uniform
    resolve_slow_requests(tab, req, req_cap, slow, slow_cap, xlocks, distinct, ctl, skeys);
    __syncthreads();
  }
  uint32_t n = n_slow ? ld_agent_u32(&ctl->n_req) : ctl->n_req;  // the resolver appends requests
  if (n > req_cap) n = req_cap;
  RATSDF_STAMP(ctl->stamps, 12);
  uint32_t total = 0;
  if (n <= kSmallRank) {
    uint32_t* lds_req = lds_rank + kSmallRank;  // request index of each listed winner
    if (tid == 0) lds[32] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < n; i += nt) {
      const Request r = req[i];
      bool win = (r.flags & kReqWinner) != 0;
      if (!(r.flags & kReqPlaced)) {
        const uint32_t bucket = block_hash(r.x, r.y, r.z, tab.bucket_mask);
        win = tab.claim[bucket] == r.rank;
        if (win) req[i].flags = kReqWinner;
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
    for (uint32_t w = tid; w < total; w += nt) {  // winner w: count the winners with smaller rank
      const uint32_t mine = lds_rank[w];
      uint32_t k = 0;
#pragma unroll 4
      for (uint32_t j = 0; j < total; ++j) k += lds_rank[j] < mine;
      req_k[lds_req[w]] = k;
    }
    RATSDF_STAMP(ctl->stamps, 10);
  } else {
    for (uint32_t i = tid; i < n; i += nt) {
      const Request r = req[i];
      bool win = (r.flags & kReqWinner) != 0;
      if (!(r.flags & kReqPlaced)) {
        const uint32_t bucket = block_hash(r.x, r.y, r.z, tab.bucket_mask);
        win = tab.claim[bucket] == r.rank;
        if (win) req[i].flags = kReqWinner;
      }
      if (win) bitmap_set(bitmap, summary, r.rank);
    }
    __syncthreads();
    RATSDF_STAMP(ctl->stamps, 9);
    const uint32_t chunk = bitmap_chunk(nwords, nt);
    const uint32_t sum = chunk_popcount(bitmap, summary, nwords, chunk);
    const uint32_t excl = block_exclusive_scan(sum, lds, &total);
    bitmap_write_prefix(bitmap, summary, prefix, nwords, chunk, sum, excl);
    __syncthreads();
    for (uint32_t i = tid; i < n; i += nt) {
      const Request r = req[i];
      if (r.flags & kReqWinner) req_k[i] = bitmap_rank(bitmap, prefix, r.rank);
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
    ctl->alloc_base = (uint32_t)nf;
    ctl->n_win = take;
    ctl->num_free = nf - (int32_t)take;
  }
  RATSDF_STAMP(ctl->stamps, 11);
}

// Commit of one allocation request by one wave (all lanes call it; only lane 0 of a wave with
// `writer` set touches the directory).  A winner with rank k among the winners takes
// heap[alloc_base - 1 - k] (AquireBlock, voxel_mem.cu:37-41) and gets its directory entry written
// (voxel_hash.cu:72-74,101-103); every request releases its bucket's claim (ResetLocks).
// Returns true for a winner that received a block; *out_k / *out_idx / *out_entry describe it.
__device__ inline bool commit_request(const Table& tab, const Pool& pool, const Request& r,
                                      uint32_t k, uint32_t alloc_base, uint32_t n_win, bool writer,
                                      int32_t* out_idx, uint32_t* out_entry) {
  const uint32_t bucket = block_hash(r.x, r.y, r.z, tab.bucket_mask);
  const bool placed = (r.flags & kReqPlaced) != 0;
  if (!(r.flags & kReqWinner)) {
    if (writer && !placed) tab.claim[bucket] = kInf;
    return false;
  }
  uint32_t e = r.entry;
  if (writer && !placed) {
    const uint32_t e0 = bucket << 1;
    e = (load_entry(tab.entries, e0).idx < 0) ? e0 : e0 + 1;
  }
  uint32_t* pe = reinterpret_cast<uint32_t*>(tab.entries + e);
  if (k >= n_win) {  // pool exhausted (voxel_mem.cu:39): this insertion does not happen
    if (writer) {
      if (placed) pe[2] = (uint32_t)-1;
      else tab.claim[bucket] = kInf;
    }
    return false;
  }
  const int32_t idx = pool.heap[alloc_base - 1 - k];
  if (writer) {
    if (!placed) {
      pe[0] = key0(r.x, r.y);
      pe[1] = key1(r.z);
      tab.claim[bucket] = kInf;
    }
    pe[2] = (uint32_t)idx;
    atomicOr(&tab.occ[e >> 6], 1ull << (e & 63));
  }
  *out_idx = idx;
  *out_entry = e;
  return true;
}

// Stand-alone commit (test hook: allocation passes without a frame): one wave per request, block
// initialised to weight 1 / tsdf -1 / probability .5 with rgb left untouched (voxel_mem.cu:43-51).
__global__ __launch_bounds__(256) void k_commit_only(Table tab, Pool pool, const Request* req,
                                                     uint32_t req_cap, const uint32_t* req_k,
                                                     Ctl* ctl) {
  uint32_t n = ctl->n_req;
  if (n > req_cap) n = req_cap;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
  const uint32_t base = ctl->alloc_base, n_win = ctl->n_win;
  for (uint32_t i = wave; i < n; i += nwaves) {
    const Request r = req[i];
    uint32_t e;
    int32_t idx;
    const uint32_t k = (r.flags & kReqWinner) ? req_k[i] : 0u;
    if (!commit_request(tab, pool, r, k, base, n_win, lane == 0, &idx, &e)) continue;
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
