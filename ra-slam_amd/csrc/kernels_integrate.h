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
// The thread that ends up with a block's min |tsdf| also starts the block's deletion when the block
// qualifies for carving (carve_candidate, kernels_carve.h).
#pragma once
#include "kernels_carve.h"

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

template <>
struct VecIO<1> {
  static __device__ inline void load(const uint32_t* p, uint32_t (&v)[1]) { v[0] = p[0]; }
  static __device__ inline void store(uint32_t* p, const uint32_t (&v)[1]) { p[0] = v[0]; }
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
    // hnormalized(): two quotients with the same divisor (shared-divisor form, device_math.h; the
    // plain IEEE divisions for a depth outside its validity range or non-finite numerators)
    float qu, qv;
    if (recip_safe(ph.z) && fabsf(ph.x) < 1e18f && fabsf(ph.y) < 1e18f) {
      const Recip rz = make_recip(ph.z);
      qu = div_shared(ph.x, rz);
      qv = div_shared(ph.y, rz);
    } else {
      qu = ph.x / ph.z;
      qv = ph.y / ph.z;
    }
    const int u = f2i(roundf(qu));                                      // :196-199
    const int w = f2i(roundf(qv));                                      // :202
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
  const bool trunc_ok = recip_safe(P.trunc);  // uniform
  const Recip rtrunc = make_recip(P.trunc);
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    const float d = ta[j].x;
    const float sdf = ta[j].y * (d - phz[j]);                           // :216
    if (inb[j] && !(d == 0 || d > P.md) && sdf > -P.trunc) {            // :211,217
      const float ts = fminf(1, (trunc_ok && fabsf(sdf) < 1e18f) ? div_shared(sdf, rtrunc)
                                                                 : sdf / P.trunc);  // :218
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

// End of a block's update: combine the WPB waves of the block (min |tsdf|, voxels updated) through
// LDS; one thread adds the update count to its workgroup's counter (a single device-wide counter
// would cost more than the whole update: ~90 atomics/us per address) and files the block for carving
// when min |tsdf| >= 0.9 (space_carving_kernel, voxel_tsdf.cu:253-276).
template <int WPB>
__device__ inline void finish_block(const Table& tab, const CarveBufs& cb, Ctl* ctl, FrameCtl* F,
                                    const VisItem& item, bool active, float m, uint32_t nupd,
                                    uint32_t wv, uint32_t part, uint32_t lane, float* smin,
                                    uint32_t* supd) {
  m = wave_min(m);
  nupd = wave_sum(nupd);
  bool fin = active && lane == 0;
  if (WPB > 1) {
    __syncthreads();  // smin / supd free again
    if (lane == 0) {
      smin[wv] = m;
      supd[wv] = nupd;
    }
    __syncthreads();
    fin = fin && part == 0;
    if (fin) {
#pragma unroll
      for (int i = 1; i < WPB; ++i) {
        m = fminf(m, smin[wv + i]);
        nupd += supd[wv + i];
      }
    }
  }
  if (fin) {
    if (nupd) atomicAdd(&cb.upd_wg[blockIdx.x & (kUpdCounters - 1)], nupd);
    if (m >= .9f) carve_candidate(tab, cb, ctl, F, item);
  }
}

// Work lists: `vis` is kNumLists segments of seg_cap items holding the visible blocks that existed
// before the frame, bucketed by image tile (block_list_of); workgroup b serves list b & 7, which
// keeps a tile's texels in one XCD's L2.  This frame's new blocks come from the request list: every
// workgroup commits and integrates its share.
// SGPR cap: above 80 SGPRs the hardware admits only 6-7 instead of 8 workgroups of 256 threads per
// CU (MI355X_MICROARCH.md, residency formula), which pushed the last 20 % of the blocks into a
// second round of waves.
template <int VPL>
__global__ __launch_bounds__(VPL == 1 ? 512 : 256) __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(VPL <= 2 ? 8 : (VPL == 4 ? 5 : 3)))) void k_integrate(
    Table tab, Pool pool, FrameParams P, const VisItem* vis, uint32_t seg_cap, const Request* req,
    uint32_t req_cap, const uint32_t* req_k, const uint32_t* win_ranks, const float4* texA,
    const uint2* texB, CarveBufs cb, Ctl* ctl, uint32_t par) {
  constexpr int WPB = 8 / VPL;  // waves per voxel block
  constexpr int BPW = VPL == 1 ? 1 : 4 / WPB;  // voxel blocks per workgroup (256 threads; 512 for VPL 1)
  __shared__ float smin[8];
  __shared__ uint32_t supd[8];
  FrameCtl* F = &ctl->fr[par];
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
  uint32_t n_mine = F->n_list[list * kListStride];
  if (n_mine > seg_cap) n_mine = seg_cap;
  uint32_t n_req = F->n_req;
  if (n_req > req_cap) n_req = req_cap;
  const uint32_t n_win = F->n_win, alloc_base = F->alloc_base, n_winlist = F->n_winlist;

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
    VisItem item = first;
    if (active && it != wg_in_list) item = my_vis[j];
    if (active) integrate_block<VPL>(pool, P, item, false, vi0, texA, texB, &nupd, &m,
                                     it == wg_in_list ? ws : nullptr);
#ifdef RATSDF_STAMPS
    if (ws && lane == 0 && it == wg_in_list) { ws[4] = (unsigned long long)clock64(); ws[6] = wall_clock64(); }
#endif
    finish_block<WPB>(tab, cb, ctl, F, item, active, m, nupd, wv, part, lane, smin, supd);
  }
  // this frame's allocation requests: commit (pool index, directory entry, occupancy) + first update
  for (uint32_t it = blockIdx.x; it * BPW < n_req; it += gridDim.x) {
    const uint32_t t = it * BPW + blk_in_wg;
    bool active = t < n_req;
    float m = 3.0e38f;
    uint32_t nupd = 0;
    VisItem item{0, 0, 0, 0, -1, 0};
    if (active) {
      const Request r = req[t];
      uint32_t e = 0;
      int32_t idx = -1;
      const bool writer = part == 0 && lane == 0;
      uint32_t k = 0;
      if (r.flags & kReqWinner) {
        if (n_winlist) {  // few winners: position in raster order = winners with a smaller rank
          for (uint32_t j = lane; j < n_winlist; j += 64) k += win_ranks[j] < r.rank;
          k = wave_sum(k);
        } else {
          k = req_k[t];
        }
      }
      active = commit_request(tab, pool, r, k, alloc_base, n_win, writer, &idx, &e);
      if (active) {
        item = VisItem{r.x, r.y, r.z, 0, idx, e};
        integrate_block<VPL>(pool, P, item, true, vi0, texA, texB, &nupd, &m);
      }
    }
    finish_block<WPB>(tab, cb, ctl, F, item, active, m, nupd, wv, part, lane, smin, supd);
  }
}

}  // namespace ratsdf
