// kernels_cand.h -- the directory-independent half of block_allocate_kernel
// (utils/tsdf/voxel_tsdf.cu:120-168) for gfx950.
//
// Which blocks a frame asks for depends only on its depth image and pose: the ray samples of every
// valid pixel, rounded to voxels and shifted to blocks (:137-164).  Whether a request then inserts
// anything depends on the directory (:165-166).  The first part is ~700 VALU instructions per pixel
// wave and touches no map state, so it does not have to sit on the frame's critical path: when the
// caller hands over a batch of frames (ratsdf_integrate_device_batch = the queue of
// TSDFSystem::Run, modules/tsdf_module.cc:88-115), the candidate pass of frame f+1 runs as extra
// workgroups inside frame f's k_front and k_integrate (kernels_frame.h).  A frame nobody looked
// ahead for (a single ratsdf_integrate_device call, the first frame of a batch) runs both halves in the workgroups of
// its own k_front (cand_inline_role / k_front_inline; until round 5 the pixel half was a launch of its own, k_cand).
//
// Output of the pass = the frame's candidate list: (block, raster rank = pixel * S + sample) pairs,
// deduplicated per workgroup with the SMALLEST rank kept.  Only the first request for a block
// matters: under the canonical raster-order execution a later request for the same block finds it
// either inserted by the first one or blocked by a bucket lock that stays taken until the end of the
// pass (voxel_hash.cu:46-108), so it can never change the directory.  A 640x480 frame has ~0.9 M
// samples, ~100 k lane-distinct requests, but only ~15 k workgroup-distinct ones (~5 k distinct
// blocks); the directory-dependent half (cand_consume_role, in k_front) then does ~15 k lookups.
// Duplicates between workgroups are harmless: the per-bucket claim (atomicMin of the rank) picks
// the smallest rank exactly as it did when every lane filed its own request.
//
// Requests of a workgroup meet in a small LDS hash set {block, min rank}; after a barrier the
// occupied slots are compacted and appended to one of 64 global lists with ONE returning atomic per
// workgroup (64 counters on separate cache lines: a single-address atomic sustains only ~90 ops/us
// on this part).  The consumer resets the counters.  Two lists are used alternately (frame parity).
// Also written here: the packed per-pixel texels k_integrate gathers from (texA = {depth, range,
// log ht - log lt, w_new}, texB = rgb); the logs and w_new = (1 - d / max_depth) * 4 are functions of
// the pixel only (voxel_tsdf.cu:226,243,246).
#pragma once
#include "kernels_alloc.h"

namespace ratsdf {

constexpr int kCandSegs = 64;           // candidate lists (and consumer workgroups) per frame
constexpr int kCandCountStride = 32;    // words between list counters (one 128-byte line each)
constexpr uint32_t kCandReserve = 32;   // list entries a 16x16 super-tile reserves up front
constexpr unsigned long long kCandEmpty = ~0ull;


// One candidate pass (or a share of it).  The image is cut into 16x4-pixel tiles, one per wave; four
// vertically stacked tiles form a 16x16 super-tile and tiles are numbered super-tile by super-tile,
// so that the waves of a workgroup cover a compact patch of the image: neighbouring pixels ask for
// the same blocks, and the per-workgroup deduplication is what keeps the candidate list short
// (row-segment workgroups produced ~120 k candidates per 640x480 frame, patches ~20 k).
struct CandJob {
  FrameParams P;
  const float* depth;
  const uint8_t* rgb;
  const float* ht;
  const float* lt;
  float4* texA;
  uint32_t* texB;
  CandSet set;
  uint32_t first_tile, n_tiles;  // this share: tiles [first_tile, first_tile + n_tiles)
  uint32_t tiles_per_wg;         // waves of a workgroup that take a tile; the others only join barriers
  uint32_t tiles_x;              // super-tiles per image row = ceil(W / 16)
  uint32_t tiles_x_magic;        // floor(2^32 / tiles_x) + 1: n / tiles_x = mulhi(n, magic) for n * tiles_x < 2^32
};
inline uint32_t cand_tiles_magic(uint32_t tiles_x) { return tiles_x > 1 ? 0xFFFFFFFFu / tiles_x + 1u : 0u; }

// Workgroup-level aggregation: the pixels of one workgroup ask for the same few dozen blocks over
// and over.  A request that finds the LDS set full goes to the global list directly.
constexpr uint32_t kCandLdsSlots = 512;

struct CandLds {
  unsigned long long keys[kCandLdsSlots];
  uint32_t ranks[kCandLdsSlots];
  uint32_t n, base;
};

// append one candidate straight to a global list (LDS set overflow only)
__device__ inline void cand_append(const CandSet& cs, uint32_t seg, uint32_t k0, uint32_t k1,
                                   uint32_t rank, Ctl* ctl) {
  const uint32_t pos = atomicAdd(&cs.count[seg * kCandCountStride], 1u);
  if (pos < cs.seg_cap) {
    cs.list[(size_t)seg * cs.seg_cap + pos] = make_uint4(k0, k1, rank, 0u);
  } else {
    set_error(ctl, RATSDF_ERR_CAPACITY);
  }
}

__device__ inline bool cand_lds_insert(CandLds& L, unsigned long long key, uint32_t h, uint32_t rank) {
  uint32_t s = h & (kCandLdsSlots - 1);
  for (uint32_t guard = 0; guard < 32; ++guard) {
    unsigned long long k = L.keys[s];
    if (k == kCandEmpty) k = atomicCAS(&L.keys[s], kCandEmpty, key);
    if (k == kCandEmpty || k == key) {
      atomicMin(&L.ranks[s], rank);
      return true;
    }
    s = (s + 1) & (kCandLdsSlots - 1);
  }
  return false;
}

// block_allocate_kernel up to (not including) the directory lookup, one lane per pixel, one 16x4
// tile per wave.  `wg` counts workgroups inside the job.
// `overflow(bx, by, bz, k0, k1, rank)`: what happens to a candidate that finds the workgroup's LDS set full.
template <typename Overflow>
__device__ inline void cand_pixel_work(const CandJob& J, CandLds& L, uint32_t wg, Ctl* ctl, uint32_t reserved_at,
                                       Overflow overflow) {
  const FrameParams& P = J.P;
  // (the tile's place in the image is the same for the whole wave: scalar arithmetic, and the division by the
  // row length as a multiplication -- as vector code with a run-time divisor it was 33 instructions per lane)
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  // Which super-tile a workgroup takes: horizontal neighbours (2k, 2k + 1) go to workgroups 8 apart -- the same
  // XCD (round-robin placement), dispatched back to back.  A 16-pixel row of a float image is HALF a 128-byte
  // line; with neighbours on different XCDs every line of depth / ht / lt was fetched from memory twice (once per
  // L2) and a wave's inputs took 4.3 us to arrive at 1280x720 (2.0 us at 640x480), the longest phase of the pass.
  // (shares start at multiples of 16 super-tiles -- ratsdf_engine::geometry -- so relative = absolute parity)
  if (J.tiles_per_wg == 4 && (wg | 15u) < (J.n_tiles + 3) / 4) wg = (wg & ~15u) | ((wg & 7u) << 1) | ((wg >> 3) & 1u);
  const uint32_t rel = wg * J.tiles_per_wg + wave;       // tile within the share
  const uint32_t tile = J.first_tile + rel;
  const uint32_t sup = tile >> 2, sub = tile & 3u;
  const uint32_t srow = J.tiles_x > 1 ? __umulhi(sup, J.tiles_x_magic) : sup;
  const uint32_t scol = sup - srow * J.tiles_x;
  const int px = (int)scol * 16 + (lane & 15);
  const int py = (int)srow * 16 + (int)sub * 4 + (lane >> 4);
  const bool inb = wave < J.tiles_per_wg && rel < J.n_tiles && px < P.W && py < P.H;
  const int pix = inb ? py * P.W + px : 0;
  // every input of the pixel is requested here, in ONE round of loads (depth, ht, lt, the three colour
  // bytes): read where they are used they made three dependent memory round trips (depth -> ht / lt ->
  // colour) in a workgroup whose whole life is ~6 us
  float d = 0.f, hv = 1.f, lv = 1.f;
  uint32_t c0 = 0, c1 = 0, c2 = 0;
#ifdef RATSDF_STAMPS
  unsigned long long pt[5];
  pt[0] = clock64();
#endif
  if (inb) {
    // (the colour bytes FIRST: the compiler moves the logarithms of ht / lt up into the has_sem branch, right behind
    // their loads; with the colour loads after that branch the "one round" was two -- depth / ht / lt, a wait,
    // the logarithms, then the colour loads and a second wait)
    c0 = J.rgb[3 * pix];
    c1 = J.rgb[3 * pix + 1];
    c2 = J.rgb[3 * pix + 2];
    d = J.depth[pix];
    if (P.has_sem) {
      hv = J.ht[pix];
      lv = J.lt[pix];
    }
  }
  const uint32_t c = c0 | (c1 << 8) | (c2 << 16);

  // (the workgroup's list reservation -- a returning atomic issued before the loads above -- is put where the
  // other waves find it HERE: the wait for it is the wait for the inputs; read at the end of the workgroup it
  // would be a `s_waitcnt vmcnt(0)` behind the texel stores, i.e. a wait for their acknowledgement)
  if (threadIdx.x == 0) L.base = reserved_at;
  const V3 pimg{(float)px, (float)py, 1.f};
  const V3 pc = intr_mul(P.Ki, pimg);                                   // :137
  const float r = sqrtf(pc.x * pc.x + (pc.y * pc.y + pc.z * pc.z));     // :140 (Eigen norm order)
  if (inb) {
    // log ht - log lt: the observation's log-odds, the only form in which ht and lt enter the
    // probability update (kernels_integrate.h); -inf / +inf / NaN for ht = 0 / lt = 0 / both
    float ln = 0.f;
    if (P.has_sem) ln = __logf(hv) - __logf(lv);
    // :226.  d / max_depth with the frame's shared reciprocal: the correctly rounded quotient for every depth the
    // update can use (finite, at most max_depth; a texel's w_new is read only when its voxel updates)
    const Recip rmd = make_recip(P.md);
    const float wn = (1 - (recip_safe(P.md) ? div_shared(d, rmd) : d / P.md)) * 4;
    J.texA[pix] = make_float4(d, r, ln, wn);
    J.texB[pix] = c;
  }
  bool valid = inb && !(d == 0 || d > P.md);                            // :141
  if (RATSDF_DBG(P, 1)) return;  // uniform
#ifdef RATSDF_STAMPS
  pt[1] = clock64();  // (the inputs have arrived: the texel values above needed them)
#endif

  const V3 pcd{pc.x * d, pc.y * d, pc.z * d};
  const V3 pw = se3_apply(P.Ti, pcd);                                   // :146
  // A map split over ranks by block ownership: every sample of this pixel's ray lies within the
  // truncation distance of the surface point, so its blocks' x lies in a short interval; when no slab of that
  // interval belongs to this rank, every request of the pixel would be dropped by shard_owned() further down --
  // the pixel is done, and so is the wave when that holds for all of its pixels (the ray set-up and the sample
  // loop are ~2/3 of the pass's instructions and all of its LDS traffic).  Conservative: 2.5 voxels of slack for
  // the roundings; coordinates outside the shorts' range (which wrap) never skip.
  if (P.shard_count > 1) {  // uniform
    const float lo = (pw.x - P.trunc) / P.vs - 2.5f, hi = (pw.x + P.trunc) / P.vs + 2.5f;
    if (lo > -32000.f && hi < 32000.f) {
      const int s_lo = ((int)floorf(lo) >> 3) >> P.shard_slab_bits, s_hi = ((int)floorf(hi) >> 3) >> P.shard_slab_bits;
      const uint32_t u = (uint32_t)(s_lo + P.shard_bias);
      const uint32_t m_lo = u - __umulhi(u, P.shard_magic) * (uint32_t)P.shard_count;  // floormod(s_lo, count)
      uint32_t k0 = (uint32_t)P.shard_rank + (uint32_t)P.shard_count - m_lo;           // slabs up to the next own one
      if (k0 >= (uint32_t)P.shard_count) k0 -= (uint32_t)P.shard_count;
      if (k0 > (uint32_t)(s_hi - s_lo)) valid = false;
    }
    if (!__any(valid)) return;  // uniform
  }
  // shared-divisor divisions (device_math.h): r in [1, ~3], voxel size a frame constant
  const Recip rr = make_recip(r), rvs = make_recip(P.vs);
  const bool vs_ok = recip_safe(P.vs);  // uniform
  const V3 dc{div_shared(pc.x, rr), div_shared(pc.y, rr), div_shared(pc.z, rr)};     // :148
  const V3 dw = quat_rotate(P.Ti.q, dc);                                // :150
  const V3 sw{pw.x - dw.x * P.trunc, pw.y - dw.y * P.trunc, pw.z - dw.z * P.trunc};  // :151
  V3 dg, sg;
  // (one decision for the wave: as a per-lane choice the compiler evaluated BOTH forms for every lane -- six full
  // IEEE divisions, 66 instructions, beside the six short ones.  The long form is right for every lane.)
  if (vs_ok && __all(fabsf(sw.x) < 1e18f && fabsf(sw.y) < 1e18f && fabsf(sw.z) < 1e18f)) {
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
    // den = 2^k: its reciprocal is the same mantissa with the exponent mirrored (one integer subtraction, exact)
    const float inv = __uint_as_float(0x7F000000u - __float_as_uint(den));
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
  // every sample of every ray of the wave within +-2^30 voxels (always, for a map whose coordinates fit the
  // reference's shorts): the short form of (int)roundf below; else the long one
  const bool small = __all(!valid || (fabsf(sg.x) + fabsf(rg.x) < 1e9f && fabsf(sg.y) + fabsf(rg.y) < 1e9f &&
                                      fabsf(sg.z) + fabsf(rg.z) < 1e9f));  // uniform
#ifdef RATSDF_STAMPS
  pt[2] = clock64();
  pt[4] = 0;
#endif
  for (int i = 0; i < P.S; ++i) {       // uniform trip count: the shuffles below need every lane
    const bool act = valid && i <= steps;
    if (!__any(act)) break;             // (uniform) no ray of the wave reaches this sample
    int gx, gy, gz;                                                      // :163-164
    if (small) {
      gx = (int16_t)round_to_int(p.x);
      gy = (int16_t)round_to_int(p.y);
      gz = (int16_t)round_to_int(p.z);
    } else {
      gx = (int16_t)f2i(roundf(p.x));
      gy = (int16_t)f2i(roundf(p.y));
      gz = (int16_t)f2i(roundf(p.z));
    }
    const int bx = gx >> 3, by = gy >> 3, bz = gz >> 3;
    const uint32_t k0 = act ? key0(bx, by) : kInf;
    const uint32_t k1 = act ? key1(bz) : kInf;
    // repeats of the previous sample / the previous pixel carry a larger rank for the same block
    // (the left neighbour's key by DPP wave_shr:1 -- a register read of the neighbouring lane; __shfl_up
    // is ds_bpermute_b32: an LDS round trip and index arithmetic, twice per sample, on the critical path
    // of a pass whose workgroups live ~6 us)
    const uint32_t n0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)k0, 0x138, 0xF, 0xF, false);
    const uint32_t n1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)k1, 0x138, 0xF, 0xF, false);
    const bool dup = (k0 == prev0 && k1 == prev1) || (lane > 0 && k0 == n0 && k1 == n1);
    if (act && !dup && shard_owned(bx, P) && !RATSDF_DBG(P, 2)) {
      const uint32_t rank = (uint32_t)pix * (uint32_t)P.S + (uint32_t)i;
      const unsigned long long key = (unsigned long long)k0 | ((unsigned long long)k1 << 32);
      // (slot of the workgroup's set: any spreading of neighbouring blocks will do -- two 24-bit multiply-adds
      // instead of the directory hash's three full 32-bit multiplications, which issue at a quarter of the rate)
      if (!cand_lds_insert(L, key, (uint32_t)(bx + by * 17 + bz * 41), rank)) overflow(bx, by, bz, k0, k1, rank);
    }
    if (act) {
      prev0 = k0;
      prev1 = k1;
    }
    p.x += st.x;
    p.y += st.y;
    p.z += st.z;
#ifdef RATSDF_STAMPS
    pt[4] += 1;
#endif
  }
#ifdef RATSDF_STAMPS
  pt[3] = clock64();
  if (threadIdx.x == 0 && (wg & 15) == 0) {  // [cand stamps] of ratsdf_debug_tail_stamps
    atomicAdd(&ctl->tstamps[11], pt[1] - pt[0]);
    atomicAdd(&ctl->tstamps[12], pt[2] - pt[1]);
    atomicAdd(&ctl->tstamps[13], pt[3] - pt[2]);
    atomicAdd(&ctl->tstamps[14], pt[4]);
    atomicAdd(&ctl->tstamps[15], 1ull);
  }
#endif
}

// `L`: LDS of the workgroup (the caller owns it so that a kernel with several roles can share one buffer)
__device__ inline void cand_pixels_role(const CandJob& J, uint32_t wg, Ctl* ctl, CandLds& L) {
  // (the workgroup size in a scalar register: device_types.h, block_threads())
  const uint32_t nthreads = block_threads();
  for (uint32_t i = threadIdx.x; i < kCandLdsSlots; i += nthreads) {
    L.keys[i] = kCandEmpty;
    L.ranks[i] = kInf;
  }
  // Space in the global list is reserved NOW, kCandReserve entries per 16x16 super-tile, so that the
  // returning atomic overlaps the whole pass instead of ending it (it was a full memory round trip
  // between two barriers at the end of every workgroup); unused entries are written as empties.
  const uint32_t seg = (J.first_tile + wg) & (kCandSegs - 1);
  const uint32_t reserve = kCandReserve * ((J.tiles_per_wg + 3) / 4);
  // Thread 0 keeps the answer in a register until the workgroup's inputs have arrived (cand_pixel_work).  The
  // counter's address goes through a vector register the compiler cannot see through: with an address it knows
  // to be the same for the whole wave its atomic optimiser rewrites atomicAdd() into "one lane adds for the wave,
  // the others read its result with v_readfirstlane", which put `s_waitcnt vmcnt(0)` right behind the atomic -- a
  // full round trip to L2 at the top of every workgroup, before its first input load.
  uint32_t reserved_at = 0;
  if (threadIdx.x == 0) {
    L.n = 0;
    uint32_t opaque_zero;
    asm("v_mov_b32 %0, 0" : "=v"(opaque_zero));
    reserved_at = atomicAdd(&J.set.count[seg * kCandCountStride + opaque_zero], reserve);
  }
#ifdef RATSDF_STAMPS
  unsigned long long cs[4];
  cs[0] = clock64();
  unsigned long long* ws = (ctl->debug_buf && RATSDF_DBG(J.P, 8))
                               ? ctl->debug_buf + (size_t)(((J.first_tile / 4 + wg) * 4 + (threadIdx.x >> 6)) & 16383) * 8
                               : nullptr;
  if (ws && (threadIdx.x & 63) == 0) { ws[0] = cs[0]; ws[5] = wall_clock64(); }
#endif
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // LDS-only barrier (the set is ready)
#ifdef RATSDF_STAMPS
  cs[1] = clock64();
#endif
  if ((threadIdx.x >> 6) < J.tiles_per_wg)  // whole waves in or out
    cand_pixel_work(J, L, wg, ctl, reserved_at, [&](int, int, int, uint32_t k0, uint32_t k1, uint32_t rank) {
      cand_append(J.set, (J.first_tile + wg) & (kCandSegs - 1), k0, k1, rank, ctl);  // straight to the global list
    });
#ifdef RATSDF_STAMPS
  cs[2] = clock64();
#endif
  // LDS-only barriers from here on: only the hash set and two counters are handed over, while
  // __syncthreads() would also wait for the texel stores in flight (~1 us)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef RATSDF_STAMPS
  cs[3] = clock64();
  if (threadIdx.x == 0 && (wg & 15) == 0) {
    atomicAdd(&ctl->stamps[14], cs[1] - cs[0]);
    atomicAdd(&ctl->stamps[15], cs[2] - cs[1]);
    atomicAdd(&ctl->stamps[16], cs[3] - cs[2]);
    atomicAdd(&ctl->stamps[17], 1ull);
  }
#endif
  // compact the occupied slots
  constexpr uint32_t kMaxPerThread = kCandLdsSlots / 64;  // block_threads() >= 64
  uint32_t pos[kMaxPerThread];
  // (a 256-thread workgroup covers the set in two rounds: the other six are skipped by a scalar branch -- as
  // predicated straight-line code they were 72 vector instructions per wave that did nothing)
#pragma unroll
  for (uint32_t k = 0; k < kMaxPerThread; ++k) {
    pos[k] = kInf;
    if (k * nthreads < kCandLdsSlots) {  // uniform
      const uint32_t i = threadIdx.x + k * nthreads;
      if (i < kCandLdsSlots && L.keys[i] != kCandEmpty) pos[k] = atomicAdd(&L.n, 1u);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  const uint32_t n = L.n;
  const uint32_t base = L.base;
  uint32_t base2 = 0;  // a second piece for a workgroup with more candidates than it reserved
  if (n > reserve) {   // uniform
    __syncthreads();
    if (threadIdx.x == 0) L.base = atomicAdd(&J.set.count[seg * kCandCountStride], n - reserve);
    __syncthreads();
    base2 = L.base;
  }
  if (base + reserve > J.set.seg_cap || (n > reserve && base2 + (n - reserve) > J.set.seg_cap)) {
    if (threadIdx.x == 0) set_error(ctl, RATSDF_ERR_CAPACITY);  // uniform
    return;
  }
  uint4* out = J.set.list + (size_t)seg * J.set.seg_cap;
#pragma unroll
  for (uint32_t k = 0; k < kMaxPerThread; ++k) {
    const uint32_t i = threadIdx.x + k * nthreads;
    if (pos[k] != kInf) {
      const unsigned long long key = L.keys[i];
      const uint32_t at = pos[k] < reserve ? base + pos[k] : base2 + (pos[k] - reserve);
      out[at] = make_uint4((uint32_t)key, (uint32_t)(key >> 32), L.ranks[i], 0u);
    }
  }
  for (uint32_t i = n + threadIdx.x; i < reserve; i += nthreads)
    out[base + i] = make_uint4(kInf, kInf, kInf, 0u);  // empty
#ifdef RATSDF_STAMPS
  if (threadIdx.x == 0 && (wg & 15) == 0) atomicAdd(&ctl->stamps[18], (unsigned long long)clock64() - cs[3]);
  if (ws && (threadIdx.x & 63) == 0) {
    ws[1] = cs[1]; ws[2] = cs[2]; ws[3] = cs[3]; ws[4] = clock64(); ws[6] = wall_clock64();
  }
#endif
}

// is_block_visible<true> (voxel_tsdf.cu:75-96) for the lanes of a wave that hold an absent
// candidate, spread over the lanes: 8 candidates x 8 corners per step.  Every lane of the wave must
// call it; returns the answer for the calling lane's own candidate.
__device__ inline bool wave_block_visible_full(bool want, int bx, int by, int bz,
                                               const FrameParams& P) {
  const int lane = threadIdx.x & 63;
  unsigned long long todo = __ballot(want);
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
    const int cbx = __shfl(bx, src), cby = __shfl(by, src), cbz = __shfl(bz, src);
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
  return want && ((ok >> lane) & 1ull);
}

// The directory-dependent half (voxel_tsdf.cu:165-166 + VoxelHashTable::Allocate): workgroup `seg`
// takes slot list `seg` of the frame's candidate set; one lane per candidate looks the block up,
// absent blocks take the full-frustum test and file their allocation request with the rank the set
// recorded.  Every slot read is emptied again.
// `gate()` makes sure the previous frame's queued deletes have happened before the directory is read
// (carve_resolve_gate); the list itself does not depend on it, so its loads are issued first.
template <typename Gate>
__device__ inline void cand_consume_role(const Table& tab, const FrameParams& P, const CandSet& cs,
                                         uint32_t seg, uint32_t part, uint32_t parts, Request* req,
                                         uint32_t req_cap, SlowRequest* slow, uint32_t slow_cap,
                                         Ctl* ctl, FrameCtl* F, Gate gate, ReqBuf& B) {
  if (threadIdx.x == 0) B.n = 0;
  __syncthreads();
  // `parts` workgroups share a list (part = 0 .. parts-1): the lists of a large or finely resolved
  // image hold thousands of candidates each.  The counters are reset by the serial role afterwards.
  const uint4* list = cs.list + (size_t)seg * cs.seg_cap;
  const uint32_t stride = parts * block_threads();
  // the count and the first batch of items are fetched together (list memory is always readable)
  const uint32_t i0 = part * block_threads() + threadIdx.x;
  uint4 item = list[i0 < cs.seg_cap ? i0 : 0];
  uint32_t n = cs.count[seg * kCandCountStride];
  if (gate() == kGateExpired) return;  // uniform: the directory may be half-edited (sticky error set)
  if (n > cs.seg_cap) n = cs.seg_cap;
  for (uint32_t base = part * block_threads(); base < n; base += stride) {  // uniform
    const uint32_t i = base + threadIdx.x;
    bool have = i < n;
    if (base != part * block_threads() && have) item = list[i];
    have = have && item.y != kInf;  // reserved but unused entry
    int bx = 0, by = 0, bz = 0;
    EntryWords ea{0, 0, -1}, eb{0, 0, -1};
    bool absent = false;
    if (have) {
      const uint32_t k0 = item.x, k1 = item.y;
      bx = (int16_t)(k0 & 0xFFFFu);
      by = (int16_t)(k0 >> 16);
      bz = (int16_t)(k1 & 0xFFFFu);
      const uint32_t e0 = block_hash(bx, by, bz, tab.bucket_mask) << 1;
      ea = load_entry(tab.entries, e0);
      eb = load_entry(tab.entries, e0 + 1);
      absent = !block_present_pre(tab, k0, k1, e0, ea, eb);
    }
    const bool want = wave_block_visible_full(absent, bx, by, bz, P);
    alloc_request_absent_wave(want, tab, bx, by, bz, item.z, ea, eb, req, req_cap, slow, slow_cap, ctl,
                              F, B);
  }
  req_buf_flush(B, req, req_cap, ctl, F, tab.tail_on != 0);
}

// Both halves in ONE workgroup, for a frame nobody looked ahead for (a single ratsdf_integrate_device call, the first
// frame of a batch): the workgroup's pixels fill its LDS set as above, and the same workgroup then looks its OWN
// candidates up and files their requests -- no candidate list, no consumer workgroups, no launch of its own for the
// pixel half (until round 5: k_cand, then k_front; a caller that synchronises after every frame -- TSDFGrid::Integrate's
// convention -- paid three launches per frame).  Duplicates between workgroups are settled by the claims, as ever.
// A candidate that finds the LDS set full (never seen; 512 slots for the few dozen blocks a 16x16-pixel patch asks
// for) files its request on the spot.  `lds`: sizeof(CandLds) rounded up to 16 bytes + sizeof(ReqBuf) bytes.
constexpr uint32_t kInlineReqOffsetWords = ((uint32_t)sizeof(CandLds) + 15u) / 16u * 4u;
template <typename Gate>
__device__ inline void cand_inline_role(const CandJob& J, uint32_t wg, const Table& tab, Request* req, uint32_t req_cap,
                                        SlowRequest* slow, uint32_t slow_cap, Ctl* ctl, FrameCtl* F, Gate gate,
                                        uint32_t* lds) {
  CandLds& L = *reinterpret_cast<CandLds*>(lds);
  ReqBuf& B = *reinterpret_cast<ReqBuf*>(lds + kInlineReqOffsetWords);
  const FrameParams& P = J.P;
  for (uint32_t i = threadIdx.x; i < kCandLdsSlots; i += 256) {
    L.keys[i] = kCandEmpty;
    L.ranks[i] = kInf;
  }
  if (threadIdx.x == 0) {
    L.n = 0;
    B.n = 0;
  }
  // the previous frame's queued deletes have happened before anything here reads the directory (the overflow path
  // below may, in the middle of the pixel half)
  const bool open = gate() != kGateExpired;  // uniform; expired: the sticky error is set, the frame is incomplete
  __syncthreads();
  if ((threadIdx.x >> 6) < J.tiles_per_wg)  // whole waves in or out
    cand_pixel_work(J, L, wg, ctl, 0u, [&](int bx, int by, int bz, uint32_t, uint32_t, uint32_t rank) {
      if (open && block_visible<true>(bx, by, bz, P)) alloc_request(tab, bx, by, bz, rank, req, req_cap, slow, slow_cap, ctl, F);
    });
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // LDS-only barrier: the set is complete
  if (!open) return;
  // the directory half over the workgroup's own set: one lane per slot (cand_consume_role's loop body)
  for (uint32_t base = 0; base < kCandLdsSlots; base += 256) {  // uniform
    const uint32_t i = base + threadIdx.x;
    const unsigned long long key = L.keys[i];
    const bool have = key != kCandEmpty;
    int bx = 0, by = 0, bz = 0;
    EntryWords ea{0, 0, -1}, eb{0, 0, -1};
    bool absent = false;
    if (have) {
      const uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
      bx = (int16_t)(k0 & 0xFFFFu);
      by = (int16_t)(k0 >> 16);
      bz = (int16_t)(k1 & 0xFFFFu);
      const uint32_t e0 = block_hash(bx, by, bz, tab.bucket_mask) << 1;
      ea = load_entry(tab.entries, e0);
      eb = load_entry(tab.entries, e0 + 1);
      absent = !block_present_pre(tab, k0, k1, e0, ea, eb);
    }
    const bool want = wave_block_visible_full(absent, bx, by, bz, P);
    alloc_request_absent_wave(want, tab, bx, by, bz, L.ranks[i], ea, eb, req, req_cap, slow, slow_cap, ctl, F, B);
  }
  req_buf_flush(B, req, req_cap, ctl, F, tab.tail_on != 0);
}

// stand-alone candidate pass (single frames, first frame of a batch)
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80))) void k_cand(CandJob job,
                                                                                   Ctl* ctl) {
  __shared__ CandLds L;
  cand_pixels_role(job, blockIdx.x, ctl, L);
}

}  // namespace ratsdf
