// kernels_integrate.h -- per-frame voxel update and space carving for gfx950.
//
// k_integrate replaces tsdf_integrate_kernel AND the read half of space_carving_kernel
// (utils/tsdf/voxel_tsdf.cu:170-276).  An 8x8x8 voxel block is 512 consecutive voxels in each of the
// three SoA pools; a lane owns VPL consecutive voxels of an x-row (VPL = 2, 4 or 8), so a wave
// reads/writes one contiguous 64*VPL*4-byte span per pool with 8- or 16-byte vector accesses.
// The kernel is latency bound (gathers of per-pixel data through L2 / Infinity Cache), so the order
// inside a lane is: issue the pool loads, project all VPL voxels, issue ALL texel gathers, then do
// the arithmetic -- every lane has 3 + 2*VPL independent memory operations in flight.  The block's
// min |tsdf| (space carving) is reduced from registers with cross-lane shuffles (+ LDS across the
// waves of a block), so the reference's second pass over the block is gone.
//
// k_integrate also performs the COMMIT of this frame's allocation winners: the wave that commits a
// new block (pool index, directory entry, occupancy bit) integrates it straight away from the
// initial values held in registers (weight 1 / tsdf -1 / probability .5, rgb read from the pool as
// the reference leaves it untouched, voxel_mem.cu:43-51), so a new block costs no separate
// initialisation pass.
//
// The carve pass (VoxelHashTable::Delete, voxel_hash.cu:110-159 + ReleaseBlock, voxel_mem.cu:56-61)
// is made deterministic the same way as allocation: deletions happen in ascending hash-entry order
// (the order of the reference's visible list); deletes of a block sitting in slot 0 of its home
// bucket are lock-free and independent; head / chain deletes are serialised per home bucket by the
// bucket lock, i.e. the first one in entry order wins (atomicMin claim), and the released pool
// indices are pushed on the free list in entry order via a popcount prefix over an entry-indexed
// bitmap.  One workgroup does the whole pass.
#pragma once
#include "kernels_visible.h"

namespace ratsdf {

__device__ inline float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
  return v;
}
__device__ inline uint32_t wave_sum(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <int N>
struct VecIO;
template <>
struct VecIO<8> {
  static __device__ inline void load(const uint32_t* p, uint32_t (&v)[8]) {
    const uint4 a = reinterpret_cast<const uint4*>(p)[0], b = reinterpret_cast<const uint4*>(p)[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
  static __device__ inline void store(uint32_t* p, const uint32_t (&v)[8]) {
    reinterpret_cast<uint4*>(p)[0] = make_uint4(v[0], v[1], v[2], v[3]);
    reinterpret_cast<uint4*>(p)[1] = make_uint4(v[4], v[5], v[6], v[7]);
  }
};
template <>
struct VecIO<4> {
  static __device__ inline void load(const uint32_t* p, uint32_t (&v)[4]) {
    const uint4 a = reinterpret_cast<const uint4*>(p)[0];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  }
  static __device__ inline void store(uint32_t* p, const uint32_t (&v)[4]) {
    reinterpret_cast<uint4*>(p)[0] = make_uint4(v[0], v[1], v[2], v[3]);
  }
};
template <>
struct VecIO<2> {
  static __device__ inline void load(const uint32_t* p, uint32_t (&v)[2]) {
    const uint2 a = reinterpret_cast<const uint2*>(p)[0];
    v[0] = a.x; v[1] = a.y;
  }
  static __device__ inline void store(uint32_t* p, const uint32_t (&v)[2]) {
    reinterpret_cast<uint2*>(p)[0] = make_uint2(v[0], v[1]);
  }
};

// Update of one voxel block by the waves that own it (WPB waves, `part` = which one).  Returns the
// number of voxels this lane updated and the lane's min |tsdf| after the update.
#ifdef RATSDF_STAMPS
#define WSTAMP(i) do { if (wstamps && (threadIdx.x & 63) == 0) wstamps[i] = (unsigned long long)clock64(); } while (0)
#else
#define WSTAMP(i) do { } while (0)
#endif
template <int VPL>
__device__ inline void integrate_block(const Pool& pool, const FrameParams& P, const VisItem& item,
                                       bool fresh_in, uint32_t vi0, const float4* texA,
                                       const uint2* texB, uint32_t* out_nupd, float* out_min,
                                       unsigned long long* wstamps = nullptr) {
  bool fresh = fresh_in;
  const int tx0 = vi0 & 7, ty = (vi0 >> 3) & 7, tz = vi0 >> 6;
  const size_t v = ((size_t)item.idx << 9) + vi0;
  uint32_t tv[VPL], sv[VPL], cv[VPL];
  if (P.debug == 5) fresh = true;  // diagnostic: no voxel loads / stores
  if (P.debug != 5) VecIO<VPL>::load(pool.rgbw + v, cv);
  else
    for (int j = 0; j < VPL; ++j) cv[j] = 0;
  if (!fresh) {
    VecIO<VPL>::load(reinterpret_cast<const uint32_t*>(pool.tsdf + v), tv);
    VecIO<VPL>::load(reinterpret_cast<const uint32_t*>(pool.segm + v), sv);
  } else {  // AquireBlock initial values, voxel_mem.cu:43-51 (rgb stays as found)
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
      tv[j] = __float_as_uint(-1.f);
      sv[j] = __float_as_uint(.5f);
      cv[j] = (cv[j] & 0x00FFFFFFu) | 0x01000000u;
    }
  }
  const int gy = (int16_t)((int16_t)(item.y << 3) + ty);
  const int gz = (int16_t)((int16_t)(item.z << 3) + tz);
  const float wy = (float)gy * P.vs, wz = (float)gz * P.vs;
  float phz[VPL];
  int kk[VPL];
  bool inb[VPL];
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    const int gx = (int16_t)((int16_t)(item.x << 3) + tx0 + j);         // :183-184
    const V3 pw{(float)gx * P.vs, wy, wz};                              // :187
    const V3 pc3 = se3_apply(P.T, pw);                                  // :190
    const V3 ph = intr_mul(P.K, pc3);                                   // :193
    const int u = f2i(roundf(ph.x / ph.z));                             // :196-199
    const int w = f2i(roundf(ph.y / ph.z));                             // :202
    inb[j] = u >= 0 && u < P.W && w >= 0 && w < P.H;                    // :205
    kk[j] = inb[j] ? w * P.W + u : 0;
    phz[j] = ph.z;
  }
  WSTAMP(1);
  float4 ta[VPL];
  uint2 tb[VPL];
#pragma unroll
  for (int j = 0; j < VPL; ++j) {  // all gathers in flight together
    if (P.debug == 4) {            // diagnostic: no gathers
      ta[j] = make_float4(2.f, 1.f, -0.5f, -0.7f);
      tb[j] = make_uint2(0x00808080u, __float_as_uint(2.f));
      continue;
    }
    ta[j] = texA[kk[j]];           // depth, range, log ht, log lt
    tb[j] = texB[kk[j]];           // rgb, w_new
  }
#ifdef RATSDF_STAMPS
  if (wstamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif
  WSTAMP(2);
  uint32_t nupd = 0;
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    const float d = ta[j].x;
    const float sdf = ta[j].y * (d - phz[j]);                           // :216
    if (inb[j] && !(d == 0 || d > P.md) && sdf > -P.trunc) {            // :211,217
      const float ts = fminf(1, sdf / P.trunc);                         // :218
      const float wn = __uint_as_float(tb[j].y);                        // :226
      const uint32_t c = cv[j];
      const float wo = (float)(c >> 24);                                // :227
      const float wc = wo + wn;                                         // :228
      const float r_old = (float)(c & 0xFFu), g_old = (float)((c >> 8) & 0xFFu),
                  b_old = (float)((c >> 16) & 0xFFu);
      const uint32_t cn = tb[j].x;
      const float r_new = (float)(cn & 0xFFu), g_new = (float)((cn >> 8) & 0xFFu),
                  b_new = (float)((cn >> 16) & 0xFFu);
      // wc = weight (1..40) + w_new (0..4): six quotients share it (division, :234-247)
      const Recip rwc = make_recip(wc);
      const float rc = div_shared(r_old * wo + r_new * wn, rwc);        // :234-235
      const float gc = div_shared(g_old * wo + g_new * wn, rwc);
      const float bc = div_shared(b_old * wo + b_new * wn, rwc);
      const float t_old = __uint_as_float(tv[j]);
      const float t_num = t_old * wo + ts * wn;
      // a NaN / inf tsdf can only come from NaN / inf inputs; keep IEEE semantics for them
      tv[j] = __float_as_uint(fabsf(t_num) < 1e18f ? div_shared(t_num, rwc) : t_num / wc);  // :236
      const uint32_t wq = (uint32_t)f2i(fminf(roundf(wc), 40)) & 0xFFu; // :238
      cv[j] = ((uint32_t)f2i(roundf(rc)) & 0xFFu) | (((uint32_t)f2i(roundf(gc)) & 0xFFu) << 8) |
              (((uint32_t)f2i(roundf(bc)) & 0xFFu) << 16) | (wq << 24); // :239-240
      const float pr = __uint_as_float(sv[j]);
      if (P.debug != 6) {  // diagnostic 6: no transcendental part
      // hardware exp2/log2 based exp/log (v_exp_f32 / v_log_f32, ~1e-7 relative): the only
      // functions of the path whose last bits differ between any two libms anyway
      // log(0) = -inf is a legal input here (ht = 0, :242-247): guard the shared-divisor form
      const float lp = wo * __logf(pr) + wn * ta[j].z, ln = wo * __logf(1 - pr) + wn * ta[j].w;
      const float pos = __expf(fabsf(lp) < 1e18f ? div_shared(lp, rwc) : lp / wc);  // :242-244
      const float neg = __expf(fabsf(ln) < 1e18f ? div_shared(ln, rwc) : ln / wc);  // :245-247
      sv[j] = __float_as_uint(pos / (pos + neg));                       // :248
      }
      ++nupd;
    }
  }
  WSTAMP(3);
  if ((nupd || fresh) && P.debug != 5 && P.debug != 7) {
    VecIO<VPL>::store(reinterpret_cast<uint32_t*>(pool.tsdf + v), tv);
    VecIO<VPL>::store(reinterpret_cast<uint32_t*>(pool.segm + v), sv);
    VecIO<VPL>::store(pool.rgbw + v, cv);
  }
  // space_carving_kernel, :253-276: min |tsdf| over the block after the update
  float m = fabsf(__uint_as_float(tv[0]));
#pragma unroll
  for (int j = 1; j < VPL; ++j) m = fminf(m, fabsf(__uint_as_float(tv[j])));
  *out_nupd = nupd;
  *out_min = m;
}

// Per-block result word for the carve pass: bit 31 = carve candidate (min |tsdf| >= .9), low bits =
// voxels updated.  (A single device-wide atomic counter here costs more than the whole update:
// ~90 atomics/us per address.)  Combines the WPB waves of a block through LDS.
template <int WPB>
__device__ inline void publish_block(uint32_t* blk_info, size_t slot, bool active, float m,
                                     uint32_t nupd, uint32_t wv, uint32_t part, uint32_t lane,
                                     float* smin, uint32_t* supd) {
  m = wave_min(m);
  nupd = wave_sum(nupd);
  if (WPB == 1) {
    if (active && lane == 0) blk_info[slot] = nupd | ((m >= .9f) ? 0x80000000u : 0u);
  } else {
    __syncthreads();  // smin / supd free again
    if (lane == 0) {
      smin[wv] = m;
      supd[wv] = nupd;
    }
    __syncthreads();
    if (active && part == 0 && lane == 0) {
      float mm = smin[wv];
      uint32_t uu = supd[wv];
#pragma unroll
      for (int i = 1; i < WPB; ++i) {
        mm = fminf(mm, smin[wv + i]);
        uu += supd[wv + i];
      }
      blk_info[slot] = uu | ((mm >= .9f) ? 0x80000000u : 0u);
    }
  }
}

// ---- software-pipelined form of the block update ------------------------------------------------
// A Stage is one voxel block in flight in a lane's registers.  stage_issue() starts every memory
// operation of the block (voxel loads, then -- after projecting the lane's voxels -- the texel
// gathers) and returns without waiting; stage_finish() does the arithmetic and the stores.  A
// persistent wave keeps one block's loads in flight while it finishes the previous one, so memory
// latency overlaps arithmetic even with only two waves per SIMD and the grid is small enough to be
// fully resident (no per-block wave launches).
template <int VPL>
struct Stage {
  VisItem item;
  uint32_t tv[VPL], sv[VPL], cv[VPL];
  float4 ta[VPL];
  uint2 tb[VPL];
  float phz[VPL];
  bool inb[VPL];
  bool active;
};

template <int VPL>
__device__ inline void stage_issue(Stage<VPL>& s, const VisItem& item, bool active, const Pool& pool,
                                   const FrameParams& P, uint32_t vi0, const float4* texA,
                                   const uint2* texB) {
  s.item = item;
  s.active = active;
  if (!active) return;
  const int tx0 = vi0 & 7, ty = (vi0 >> 3) & 7, tz = vi0 >> 6;
  const size_t v = ((size_t)item.idx << 9) + vi0;
  VecIO<VPL>::load(pool.rgbw + v, s.cv);
  VecIO<VPL>::load(reinterpret_cast<const uint32_t*>(pool.tsdf + v), s.tv);
  VecIO<VPL>::load(reinterpret_cast<const uint32_t*>(pool.segm + v), s.sv);
  const int gy = (int16_t)((int16_t)(item.y << 3) + ty);
  const int gz = (int16_t)((int16_t)(item.z << 3) + tz);
  const float wy = (float)gy * P.vs, wz = (float)gz * P.vs;
  int kk[VPL];
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    const int gx = (int16_t)((int16_t)(item.x << 3) + tx0 + j);         // :183-184
    const V3 pw{(float)gx * P.vs, wy, wz};                              // :187
    const V3 pc3 = se3_apply(P.T, pw);                                  // :190
    const V3 ph = intr_mul(P.K, pc3);                                   // :193
    const int u = f2i(roundf(ph.x / ph.z));                             // :196-199
    const int w = f2i(roundf(ph.y / ph.z));                             // :202
    s.inb[j] = u >= 0 && u < P.W && w >= 0 && w < P.H;                  // :205
    kk[j] = s.inb[j] ? w * P.W + u : 0;
    s.phz[j] = ph.z;
  }
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    s.ta[j] = texA[kk[j]];  // depth, range, log ht, log lt
    s.tb[j] = texB[kk[j]];  // rgb, w_new
  }
}

template <int VPL>
__device__ inline void stage_finish(Stage<VPL>& s, const Pool& pool, const FrameParams& P,
                                    uint32_t vi0, uint32_t* out_nupd, float* out_min) {
  *out_nupd = 0;
  *out_min = 3.0e38f;
  if (!s.active) return;
  const size_t v = ((size_t)s.item.idx << 9) + vi0;
  uint32_t nupd = 0;
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    const float d = s.ta[j].x;
    const float sdf = s.ta[j].y * (d - s.phz[j]);                       // :216
    if (s.inb[j] && !(d == 0 || d > P.md) && sdf > -P.trunc) {          // :211,217
      const float ts = fminf(1, sdf / P.trunc);                         // :218
      const float wn = __uint_as_float(s.tb[j].y);                      // :226
      const uint32_t c = s.cv[j];
      const float wo = (float)(c >> 24);                                // :227
      const float wc = wo + wn;                                         // :228
      const float r_old = (float)(c & 0xFFu), g_old = (float)((c >> 8) & 0xFFu),
                  b_old = (float)((c >> 16) & 0xFFu);
      const uint32_t cn = s.tb[j].x;
      const float r_new = (float)(cn & 0xFFu), g_new = (float)((cn >> 8) & 0xFFu),
                  b_new = (float)((cn >> 16) & 0xFFu);
      const Recip rwc = make_recip(wc);
      const float rc = div_shared(r_old * wo + r_new * wn, rwc);        // :234-235
      const float gc = div_shared(g_old * wo + g_new * wn, rwc);
      const float bc = div_shared(b_old * wo + b_new * wn, rwc);
      const float t_old = __uint_as_float(s.tv[j]);
      const float t_num = t_old * wo + ts * wn;
      s.tv[j] = __float_as_uint(fabsf(t_num) < 1e18f ? div_shared(t_num, rwc) : t_num / wc);  // :236
      const uint32_t wq = (uint32_t)f2i(fminf(roundf(wc), 40)) & 0xFFu; // :238
      s.cv[j] = ((uint32_t)f2i(roundf(rc)) & 0xFFu) | (((uint32_t)f2i(roundf(gc)) & 0xFFu) << 8) |
                (((uint32_t)f2i(roundf(bc)) & 0xFFu) << 16) | (wq << 24);  // :239-240
      const float pr = __uint_as_float(s.sv[j]);
      const float lp = wo * __logf(pr) + wn * s.ta[j].z, ln = wo * __logf(1 - pr) + wn * s.ta[j].w;
      const float pos = __expf(fabsf(lp) < 1e18f ? div_shared(lp, rwc) : lp / wc);  // :242-244
      const float neg = __expf(fabsf(ln) < 1e18f ? div_shared(ln, rwc) : ln / wc);  // :245-247
      s.sv[j] = __float_as_uint(pos / (pos + neg));                     // :248
      ++nupd;
    }
  }
  if (nupd) {
    VecIO<VPL>::store(reinterpret_cast<uint32_t*>(pool.tsdf + v), s.tv);
    VecIO<VPL>::store(reinterpret_cast<uint32_t*>(pool.segm + v), s.sv);
    VecIO<VPL>::store(pool.rgbw + v, s.cv);
  }
  float m = fabsf(__uint_as_float(s.tv[0]));                            // :253-276
#pragma unroll
  for (int j = 1; j < VPL; ++j) m = fminf(m, fabsf(__uint_as_float(s.tv[j])));
  *out_nupd = nupd;
  *out_min = m;
}

// Persistent, software-pipelined k_integrate: grid = kNumLists * G workgroups (G per list, all
// resident); workgroup (list, g) walks blocks g, g + G, ... of its list with one block's loads always
// in flight behind the block being updated.  New blocks (allocation winners) are committed and
// integrated afterwards by all workgroups, as in k_integrate.
template <int VPL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80))) void k_integrate_pipe(
    Table tab, Pool pool, FrameParams P, VisItem* vis, uint32_t seg_cap, const Request* req,
    uint32_t req_cap, const uint32_t* req_k, const float4* texA, const uint2* texB,
    uint32_t* blk_info, Ctl* ctl) {
  constexpr int WPB = 8 / VPL;
  constexpr int BPW = 4 / WPB;
  __shared__ float smin[4];
  __shared__ uint32_t supd[4];
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wv = threadIdx.x >> 6;
  const uint32_t blk_in_wg = wv / WPB, part = wv % WPB;
  const uint32_t vi0 = (part * 64 + lane) * VPL;
  const uint32_t list = blockIdx.x & (kNumLists - 1);
  const uint32_t wg_in_list = blockIdx.x / kNumLists, wgs_per_list = gridDim.x / kNumLists;
  const VisItem* my_vis = vis + (size_t)list * seg_cap;
  uint32_t* my_info = blk_info + (size_t)list * seg_cap;
  // first two items are fetched together with the counters (slots exist even past the list's end)
  uint32_t it = wg_in_list;
  const uint32_t ja = it * BPW + blk_in_wg, jb = (it + wgs_per_list) * BPW + blk_in_wg;
  VisItem item_a = my_vis[ja < seg_cap ? ja : 0];
  VisItem item_b = my_vis[jb < seg_cap ? jb : 0];
  uint32_t n_mine = ctl->n_list[list];
  if (n_mine > seg_cap) n_mine = seg_cap;
  uint32_t n_req = ctl->n_req;
  if (n_req > req_cap) n_req = req_cap;
  const uint32_t n_win = ctl->n_win, alloc_base = ctl->alloc_base;

  Stage<VPL> A, B;
  if (it * BPW < n_mine) {  // uniform per workgroup
    stage_issue<VPL>(A, item_a, ja < n_mine, pool, P, vi0, texA, texB);
    while (true) {
      // ---- A in flight; start B = next block, then finish A
      const uint32_t itb = it + wgs_per_list;
      const bool has_b = itb * BPW < n_mine;
      const uint32_t jB = itb * BPW + blk_in_wg;
      if (has_b) {
        stage_issue<VPL>(B, item_b, jB < n_mine, pool, P, vi0, texA, texB);
        const uint32_t jn = (itb + wgs_per_list) * BPW + blk_in_wg;
        item_a = my_vis[jn < seg_cap ? jn : 0];  // prefetch for the block after B
      }
      {
        uint32_t nupd;
        float m;
        const uint32_t jA = it * BPW + blk_in_wg;
        stage_finish<VPL>(A, pool, P, vi0, &nupd, &m);
        publish_block<WPB>(my_info, jA, A.active, m, nupd, wv, part, lane, smin, supd);
      }
      if (!has_b) break;
      it = itb;
      // ---- B in flight; start A = next block, then finish B
      const uint32_t ita = it + wgs_per_list;
      const bool has_a = ita * BPW < n_mine;
      const uint32_t jA2 = ita * BPW + blk_in_wg;
      if (has_a) {
        stage_issue<VPL>(A, item_a, jA2 < n_mine, pool, P, vi0, texA, texB);
        const uint32_t jn = (ita + wgs_per_list) * BPW + blk_in_wg;
        item_b = my_vis[jn < seg_cap ? jn : 0];
      }
      {
        uint32_t nupd;
        float m;
        stage_finish<VPL>(B, pool, P, vi0, &nupd, &m);
        publish_block<WPB>(my_info, jB, B.active, m, nupd, wv, part, lane, smin, supd);
      }
      if (!has_a) break;
      it = ita;
    }
  }
  // this frame's allocation requests: commit (pool index, directory entry, occupancy) + first update
  for (uint32_t itq = blockIdx.x; itq * BPW < n_req; itq += gridDim.x) {
    const uint32_t t = itq * BPW + blk_in_wg;
    bool active = t < n_req;
    uint32_t k = 0;
    float m = 3.0e38f;
    uint32_t nupd = 0;
    if (active) {
      const Request r = req[t];
      uint32_t e = 0;
      int32_t idx = -1;
      const bool writer = part == 0 && lane == 0;
      k = (r.flags & kReqWinner) ? req_k[t] : 0u;
      active = commit_request(tab, pool, r, k, alloc_base, n_win, writer, &idx, &e) && k < seg_cap;
      if (active) {
        const VisItem item{r.x, r.y, r.z, 0, idx, e};
        if (writer) vis[(size_t)kNumLists * seg_cap + k] = item;
        integrate_block<VPL>(pool, P, item, true, vi0, texA, texB, &nupd, &m);
      }
    }
    publish_block<WPB>(blk_info, (size_t)kNumLists * seg_cap + k, active, m, nupd, wv, part, lane, smin,
                       supd);
  }
}

// Work lists: `vis` / `blk_info` are kNumLists + 1 segments of seg_cap items.  Segments 0..7 hold the
// visible blocks that existed before the frame, bucketed by image tile (block_list_of); workgroup b
// serves list b & 7, which keeps a tile's texels in one XCD's L2.  Segment 8 receives this frame's
// new blocks (slot = rank among the winners), committed and integrated here by every workgroup.
// SGPR cap: above 80 SGPRs the hardware admits only 6-7 instead of 8 workgroups of 256 threads per
// CU (MI355X_MICROARCH.md, residency formula), which pushed the last 20 % of the blocks into a
// second round of waves.
template <int VPL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80))) void k_integrate(Table tab, Pool pool, FrameParams P, VisItem* vis,
                                                   uint32_t seg_cap, const Request* req,
                                                   uint32_t req_cap, const uint32_t* req_k,
                                                   const float4* texA, const uint2* texB,
                                                   uint32_t* blk_info, Ctl* ctl) {
  constexpr int WPB = 8 / VPL;  // waves per voxel block
  constexpr int BPW = 4 / WPB;  // voxel blocks per 256-thread workgroup
  __shared__ float smin[4];
  __shared__ uint32_t supd[4];
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wv = threadIdx.x >> 6;
  const uint32_t blk_in_wg = wv / WPB, part = wv % WPB;
  const uint32_t vi0 = (part * 64 + lane) * VPL;  // first voxel of this lane, x + 8y + 64z
  const uint32_t list = blockIdx.x & (kNumLists - 1);
  const uint32_t wg_in_list = blockIdx.x / kNumLists, wgs_per_list = gridDim.x / kNumLists;
  const VisItem* my_vis = vis + (size_t)list * seg_cap;
  // the first item is fetched together with the counters (the slot exists even if the list is
  // shorter; it is only used when in range), which takes one memory round trip off every wave
  const uint32_t j0 = wg_in_list * BPW + blk_in_wg;
  const VisItem first = my_vis[j0 < seg_cap ? j0 : 0];
  uint32_t n_mine = ctl->n_list[list];
  if (n_mine > seg_cap) n_mine = seg_cap;
  uint32_t n_req = ctl->n_req;
  if (n_req > req_cap) n_req = req_cap;
  const uint32_t n_win = ctl->n_win, alloc_base = ctl->alloc_base;

  for (uint32_t it = wg_in_list; it * BPW < n_mine; it += wgs_per_list) {
    const uint32_t j = it * BPW + blk_in_wg;
    const bool active = j < n_mine;
    float m = 3.0e38f;
    uint32_t nupd = 0;
#ifdef RATSDF_STAMPS
    unsigned long long* ws = (ctl->debug_buf && P.debug != 8 && P.debug != 10) ? ctl->debug_buf + (size_t)((blockIdx.x * 4 + wv) & 16383) * 8 : nullptr;
    if (ws && lane == 0 && it == wg_in_list) { ws[0] = (unsigned long long)clock64(); ws[5] = wall_clock64(); }
#else
    unsigned long long* ws = nullptr;
#endif
    if (active) integrate_block<VPL>(pool, P, it == wg_in_list ? first : my_vis[j], false, vi0, texA,
                                     texB, &nupd, &m, it == wg_in_list ? ws : nullptr);
#ifdef RATSDF_STAMPS
    if (ws && lane == 0 && it == wg_in_list) { ws[4] = (unsigned long long)clock64(); ws[6] = wall_clock64(); }
#endif
    publish_block<WPB>(blk_info, (size_t)list * seg_cap + j, active, m, nupd, wv, part, lane, smin,
                       supd);
  }
  // this frame's allocation requests: commit (pool index, directory entry, occupancy) + first update
  for (uint32_t it = blockIdx.x; it * BPW < n_req; it += gridDim.x) {
    const uint32_t t = it * BPW + blk_in_wg;
    bool active = t < n_req;
    uint32_t k = 0;
    float m = 3.0e38f;
    uint32_t nupd = 0;
    if (active) {
      const Request r = req[t];
      uint32_t e = 0;
      int32_t idx = -1;
      const bool writer = part == 0 && lane == 0;
      k = (r.flags & kReqWinner) ? req_k[t] : 0u;
      active = commit_request(tab, pool, r, k, alloc_base, n_win, writer, &idx, &e) && k < seg_cap;
      if (active) {
        const VisItem item{r.x, r.y, r.z, 0, idx, e};
        if (writer) vis[(size_t)kNumLists * seg_cap + k] = item;  // the carve pass needs the entry
        integrate_block<VPL>(pool, P, item, true, vi0, texA, texB, &nupd, &m);
      }
    }
    publish_block<WPB>(blk_info, (size_t)kNumLists * seg_cap + k, active, m, nupd, wv, part, lane, smin,
                       supd);
  }
}

// explicit delete list (test hook): builds a pseudo visible list
__global__ void k_lookup_list(Table tab, const int16_t* pos, int n, VisItem* vis,
                              uint32_t* blk_info, Ctl* ctl) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) ctl->n_list[0] = (uint32_t)n;  // everything in list 0
  if (i >= n) return;
  EntryWords w;
  const int x = pos[3 * i], y = pos[3 * i + 1], z = pos[3 * i + 2];
  const uint32_t e = find_block(tab, x, y, z, &w);
  vis[i] = VisItem{(int16_t)x, (int16_t)y, (int16_t)z, (int16_t)(w.w1 >> 16), w.idx,
                   e == kInf ? 0u : e};
  blk_info[i] = e != kInf ? 0x80000000u : 0u;
}

__device__ inline void occ_clear(const Table& tab, uint32_t e) {
  atomicAnd(&tab.occ[e >> 6], ~(1ull << (e & 63)));
}
__device__ inline uint32_t ld_agent(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------
// k_carve: the whole carve pass in ONE workgroup.
//   (1) every flagged block: slot-0-of-home deletes happen directly (no lock, voxel_hash.cu:114-123),
//       the others claim their home bucket with atomicMin(entry index) and go to a small list
//   (2) head / chain deletes: one winner per home bucket = the first in entry order
//       (voxel_hash.cu:125-158); claims are released (ResetLocks)
//   (3) order of the ReleaseBlock calls = ascending hash entry of the deleted blocks:
//         few deletes (the steady state): entries in an LDS list, every delete counts the smaller
//         many deletes: entry-indexed bitmap + popcount prefix (self-cleaning)
//   (4) heap pushes, free-list bookkeeping, frame statistics; the control block is zeroed for the
//       next pass
// Data produced with atomics inside this kernel (bitmap words, claims, counters) is read either with
// agent-scope atomic loads or from lines this kernel has not touched before (L1 is cold).
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kSmallCarve = 2048;

__global__ __launch_bounds__(1024) void k_carve(Table tab, Pool pool, const VisItem* vis,
                                                uint32_t seg_cap, const uint32_t* blk_info,
                                                uint32_t* bitmap, uint32_t* summary,
                                                uint32_t* prefix, SlowDelete* slow,
                                                uint32_t slow_cap, Ctl* ctl,
                                                ratsdf_frame_stats* stats, CandJob next) {
  if (blockIdx.x != 0) {  // extra workgroups: a share of the NEXT frame's candidate pass
    cand_pixels_role(next, blockIdx.x - 1, ctl);
    return;
  }
  __shared__ uint32_t lds[32];
  __shared__ uint32_t n_list;                 // deletes recorded in the LDS list (may exceed cap)
  __shared__ uint32_t del_entry[kSmallCarve];
  __shared__ int32_t del_pool[kSmallCarve];   // pool index released by that delete
  const uint32_t tid = threadIdx.x, nt = blockDim.x;
  RATSDF_STAMP(ctl->stamps, 0);
#ifdef RATSDF_STAMPS
  if (threadIdx.x == 0) ctl->stamps[20] += wall_clock64();
#endif
  // lists 0..7: visible blocks that existed before the frame; list 8: blocks inserted this frame
  uint32_t n_seg[kNumLists + 1];
  uint32_t nv = 0;
#pragma unroll
  for (int l = 0; l < kNumLists; ++l) {
    n_seg[l] = ctl->n_list[l] < seg_cap ? ctl->n_list[l] : seg_cap;
    nv += n_seg[l];
  }
  n_seg[kNumLists] = ctl->n_win < seg_cap ? ctl->n_win : seg_cap;
  nv += n_seg[kNumLists];
#ifdef RATSDF_STAMPS
  if (threadIdx.x == 0) {
    uint32_t mx = 0;
    for (int l = 0; l < kNumLists; ++l) { ctl->stamps[22] += 0; mx = n_seg[l] > mx ? n_seg[l] : mx; }
    ctl->stamps[22] += mx;          // max list length
    ctl->stamps[23] += nv - n_seg[kNumLists];  // sum of list lengths
  }
#endif
  // flat index over all lists -> position in the segmented arrays
  auto locate = [&](uint32_t g) -> size_t {
    size_t at = 0;
#pragma unroll
    for (int l = 0; l <= kNumLists; ++l) {
      if (g < n_seg[l]) {
        at = (size_t)l * seg_cap + g;
        g = 0xFFFFFFFFu;
      } else if (g != 0xFFFFFFFFu) {
        g -= n_seg[l];
      }
    }
    return at;
  };
  const int32_t nf = ctl->num_free;
  if (tid == 0) n_list = 0;
  __syncthreads();
  uint32_t upd_part = 0;
  for (uint32_t g = tid; g < nv; g += nt) {
    const size_t at = locate(g);
    const uint32_t info = blk_info[at];
    const VisItem it = vis[at];
    upd_part += info & 0x7FFFFFFFu;
    if (!(info >> 31)) continue;
    const uint32_t bucket = block_hash(it.x, it.y, it.z, tab.bucket_mask);
    if (it.entry == (bucket << 1)) {
      uint32_t* pe = reinterpret_cast<uint32_t*>(tab.entries + it.entry);
      pe[1] = key1(it.z);  // offset = 0
      pe[2] = (uint32_t)-1;
      occ_clear(tab, it.entry);
      const uint32_t slot = atomicAdd(&n_list, 1u);
      if (slot < kSmallCarve) {
        del_entry[slot] = it.entry;
        del_pool[slot] = it.idx;
      }
    } else {
      atomicMin(&tab.claim[bucket], it.entry);
      const uint32_t slot = atomicAdd(&ctl->n_slow_del, 1u);
      if (slot < slow_cap) {
        slow[slot] = SlowDelete{it.x, it.y, it.z, 0, it.entry, -1};
      } else {
        set_error(ctl, RATSDF_ERR_CAPACITY);
      }
    }
  }
  __syncthreads();
  RATSDF_STAMP(ctl->stamps, 1);
  uint32_t ns = ld_agent(&ctl->n_slow_del);
  if (ns > slow_cap) ns = slow_cap;
  if (ns) {  // uniform
    // decide every winner before any claim is released (the state is parked in the item itself and
    // re-read by the same thread)
    for (uint32_t j = tid; j < ns; j += nt) {
      const SlowDelete s = slow[j];
      const uint32_t bucket = block_hash(s.x, s.y, s.z, tab.bucket_mask);
      slow[j].state = (ld_agent(&tab.claim[bucket]) == s.entry) ? 1 : 0;
    }
    __syncthreads();
    for (uint32_t j = tid; j < ns; j += nt) {
      const SlowDelete s = slow[j];
      const uint32_t bucket = block_hash(s.x, s.y, s.z, tab.bucket_mask);
      tab.claim[bucket] = kInf;  // ResetLocks
      if (!s.state) continue;
      const uint32_t k0 = key0(s.x, s.y), k1 = key1(s.z);
      uint32_t last = (bucket << 1) + 1;
      uint32_t* ph = reinterpret_cast<uint32_t*>(tab.entries + last);
      const EntryWords h = load_entry(tab.entries, last);
      int32_t freed = -1;
      if (entry_matches(h, k0, k1)) {                                     // voxel_hash.cu:125-140
        const uint32_t nxt = (last + (uint32_t)entry_offset(h)) & tab.entry_mask;
        uint32_t* pn = reinterpret_cast<uint32_t*>(tab.entries + nxt);
        const EntryWords nw = load_entry(tab.entries, nxt);
        freed = h.idx;
        const int noff = entry_offset(nw);
        const int16_t hoff = noff ? (int16_t)(entry_offset(h) + noff) : (int16_t)0;
        ph[0] = nw.w0;
        ph[1] = (nw.w1 & 0xFFFFu) | ((uint32_t)(uint16_t)hoff << 16);
        ph[2] = (uint32_t)nw.idx;
        pn[1] = pn[1] & 0xFFFFu;
        pn[2] = (uint32_t)-1;
        occ_clear(tab, nxt);  // the head keeps its bit unless it was its own successor
      } else {                                                            // voxel_hash.cu:142-158
        for (uint32_t g = 0; g < tab.num_entry; ++g) {
          const EntryWords lw = load_entry(tab.entries, last);
          const int loff = entry_offset(lw);
          if (!loff) break;
          const uint32_t cur = (last + (uint32_t)loff) & tab.entry_mask;
          const EntryWords cw = load_entry(tab.entries, cur);
          if (entry_matches(cw, k0, k1)) {
            const int coff = entry_offset(cw);
            const int16_t link = coff ? (int16_t)(loff + coff) : (int16_t)0;
            uint32_t* pl = reinterpret_cast<uint32_t*>(tab.entries + last);
            uint32_t* pcur = reinterpret_cast<uint32_t*>(tab.entries + cur);
            pl[1] = (pl[1] & 0xFFFFu) | ((uint32_t)(uint16_t)link << 16);
            freed = cw.idx;
            pcur[1] = pcur[1] & 0xFFFFu;
            pcur[2] = (uint32_t)-1;
            occ_clear(tab, cur);
            break;
          }
          last = cur;
        }
      }
      if (freed >= 0) {
        slow[j].state = 2;
        slow[j].freed = freed;
        const uint32_t slot = atomicAdd(&n_list, 1u);
        if (slot < kSmallCarve) {
          del_entry[slot] = s.entry;
          del_pool[slot] = freed;
        }
      }
    }
    __syncthreads();
  }
  RATSDF_STAMP(ctl->stamps, 2);
  uint32_t upd = 0;
  (void)block_exclusive_scan(upd_part, lds, &upd);
  const uint32_t n_del = n_list;  // uniform (read after the barriers above)
  RATSDF_STAMP(ctl->stamps, 3);
  if (n_del && n_del <= kSmallCarve) {
    // rank of a delete = number of deleted entries below its own (all in LDS)
    for (uint32_t w = tid; w < n_del; w += nt) {
      const uint32_t mine = del_entry[w];
      uint32_t k = 0;
#pragma unroll 4
      for (uint32_t j = 0; j < n_del; ++j) k += del_entry[j] < mine;
      pool.heap[(uint32_t)nf + k] = del_pool[w];                          // voxel_mem.cu:56-60
    }
  } else if (n_del) {
    for (uint32_t g = tid; g < nv; g += nt) {
      const size_t at = locate(g);
      const uint32_t info = blk_info[at];
      if (!(info >> 31)) continue;
      const VisItem it = vis[at];
      if (it.entry == (block_hash(it.x, it.y, it.z, tab.bucket_mask) << 1))
        bitmap_set(bitmap, summary, it.entry);
    }
    for (uint32_t j = tid; j < ns; j += nt)
      if (slow[j].state == 2) bitmap_set(bitmap, summary, slow[j].entry);
    __syncthreads();
    const uint32_t nwords = tab.num_entry >> 5;
    const uint32_t chunk = bitmap_chunk(nwords, nt);
    const uint32_t sum = chunk_popcount(bitmap, summary, nwords, chunk);
    uint32_t total = 0;
    const uint32_t excl = block_exclusive_scan(sum, lds, &total);
    bitmap_write_prefix(bitmap, summary, prefix, nwords, chunk, sum, excl);
    __syncthreads();
    for (uint32_t g = tid; g < nv; g += nt) {
      const size_t at = locate(g);
      const uint32_t info = blk_info[at];
      if (!(info >> 31)) continue;
      const VisItem it = vis[at];
      if (it.entry != (block_hash(it.x, it.y, it.z, tab.bucket_mask) << 1)) continue;
      pool.heap[(uint32_t)nf + bitmap_rank(bitmap, prefix, it.entry)] = it.idx;
    }
    for (uint32_t j = tid; j < ns; j += nt) {
      const SlowDelete s = slow[j];
      if (s.state == 2) pool.heap[(uint32_t)nf + bitmap_rank(bitmap, prefix, s.entry)] = s.freed;
    }
    __syncthreads();  // every reader of the delete bitmap is done: leave it clean for the next pass
    bitmap_clean(bitmap, summary, nwords);
  }
  RATSDF_STAMP(ctl->stamps, 4);
  if (tid == 0) {
    ctl->num_free = nf + (int32_t)n_del;
    if (stats) {
      stats->visible_blocks = (int32_t)nv;
      stats->updated_voxels = (int32_t)upd;
      stats->allocated_blocks = (int32_t)ctl->n_win;
      stats->deleted_blocks = (int32_t)n_del;
      stats->active_blocks = tab.num_block - (nf + (int32_t)n_del);
      stats->slow_requests = (int32_t)ctl->n_slow;
      ctl->totals[0] += 1;
      ctl->totals[1] += nv;
      ctl->totals[2] += upd;
      ctl->totals[3] += ctl->n_win;
      ctl->totals[4] += n_del;
    }
    // control block ready for the next pass (saves a memset node per frame)
    uint32_t* z = reinterpret_cast<uint32_t*>(ctl);
    for (int i = 0; i < kCtlFrameBytes / 4; ++i) z[i] = 0;
  }
  RATSDF_STAMP(ctl->stamps, 5);
#ifdef RATSDF_STAMPS
  if (threadIdx.x == 0) ctl->stamps[21] += wall_clock64();
#endif
}

}  // namespace ratsdf
