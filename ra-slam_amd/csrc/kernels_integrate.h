// kernels_integrate.h -- per-frame voxel update and space carving for gfx950.
//
// k_integrate replaces tsdf_integrate_kernel AND the read half of space_carving_kernel
// (utils/tsdf/voxel_tsdf.cu:170-276): one 64-lane wave owns one 8x8x8 voxel block, lane l owns the
// x-row (y = l & 7, z = l >> 3), i.e. 8 consecutive voxels = 32 contiguous bytes in each of the
// three SoA pools, moved as 16-byte vector loads/stores (the wave reads 2 KiB contiguous per pool).
// The block's min |tsdf| (space carving) is reduced across the wave with cross-lane shuffles from
// the values still in registers, so the reference's second pass over the block is gone.
//
// The carve pass (VoxelHashTable::Delete, voxel_hash.cu:110-159 + ReleaseBlock, voxel_mem.cu:56-61)
// is made deterministic the same way as allocation: deletions happen in visible-list (= ascending
// hash entry) order; deletes of a block sitting in slot 0 of its home bucket are lock-free and
// independent; head / chain deletes are serialised per home bucket by the bucket lock, i.e. the
// first one in list order wins (atomicMin claim), and the released pool indices are pushed on the
// free list in list order via a popcount prefix over a bitmap.
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

__global__ __launch_bounds__(256) void k_integrate(Pool pool, FrameParams P, const VisItem* vis,
                                                   const float4* texA, const uint2* texB,
                                                   uint8_t* carve_flag, Ctl* ctl) {
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
  const uint32_t nv = ctl->n_vis;
  const int ty = lane & 7, tz = lane >> 3;
  uint32_t updated_total = 0;
  for (uint32_t b = wave; b < nv; b += nwaves) {
    const VisItem it = vis[b];
    const size_t v = ((size_t)it.idx << 9) + lane * 8;
    float4* pt = reinterpret_cast<float4*>(pool.tsdf + v);
    float4* ps = reinterpret_cast<float4*>(pool.segm + v);
    uint4* pc = reinterpret_cast<uint4*>(pool.rgbw + v);
    float4 t0 = pt[0], t1 = pt[1];
    float4 s0 = ps[0], s1 = ps[1];
    uint4 c0 = pc[0], c1 = pc[1];
    float tv[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
    float sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    uint32_t cv[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};

    const int gy = (int16_t)((int16_t)(it.y << 3) + ty);
    const int gz = (int16_t)((int16_t)(it.z << 3) + tz);
    const float wy = (float)gy * P.vs, wz = (float)gz * P.vs;
    uint32_t nupd = 0;
#pragma unroll
    for (int tx = 0; tx < 8; ++tx) {
      const int gx = (int16_t)((int16_t)(it.x << 3) + tx);               // :183-184
      const V3 pw{(float)gx * P.vs, wy, wz};                              // :187
      const V3 pc3 = se3_apply(P.T, pw);                                  // :190
      const V3 ph = intr_mul(P.K, pc3);                                   // :193
      const int u = f2i(roundf(ph.x / ph.z));                             // :196-199
      const int w = f2i(roundf(ph.y / ph.z));                             // :202
      if (u >= 0 && u < P.W && w >= 0 && w < P.H) {                       // :205
        const int k = w * P.W + u;
        const float4 a = texA[k];  // depth, range, log ht, log lt
        const float d = a.x;
        if (!(d == 0 || d > P.md)) {                                      // :211
          const float sdf = a.y * (d - ph.z);                             // :216
          if (sdf > -P.trunc) {                                           // :217
            const uint2 bq = texB[k];  // rgb, w_new
            const float ts = fminf(1, sdf / P.trunc);                     // :218
            const float wn = __uint_as_float(bq.y);                       // :226
            const uint32_t c = cv[tx];
            const float wo = (float)(c >> 24);                            // :227
            const float wc = wo + wn;                                     // :228
            const float r_old = (float)(c & 0xFFu), g_old = (float)((c >> 8) & 0xFFu),
                        b_old = (float)((c >> 16) & 0xFFu);
            const float r_new = (float)(bq.x & 0xFFu), g_new = (float)((bq.x >> 8) & 0xFFu),
                        b_new = (float)((bq.x >> 16) & 0xFFu);
            const float rc = (r_old * wo + r_new * wn) / wc;              // :234-235
            const float gc = (g_old * wo + g_new * wn) / wc;
            const float bc = (b_old * wo + b_new * wn) / wc;
            tv[tx] = (tv[tx] * wo + ts * wn) / wc;                        // :236
            const uint32_t wq = (uint32_t)f2i(fminf(roundf(wc), 40)) & 0xFFu;   // :238
            cv[tx] = ((uint32_t)f2i(roundf(rc)) & 0xFFu) | (((uint32_t)f2i(roundf(gc)) & 0xFFu) << 8) |
                     (((uint32_t)f2i(roundf(bc)) & 0xFFu) << 16) | (wq << 24);  // :239-240
            const float pr = sv[tx];
            const float pos = expf((wo * logf(pr) + wn * a.z) / wc);      // :242-244
            const float neg = expf((wo * logf(1 - pr) + wn * a.w) / wc);  // :245-247
            sv[tx] = pos / (pos + neg);                                   // :248
            ++nupd;
          }
        }
      }
    }
    if (nupd) {
      pt[0] = make_float4(tv[0], tv[1], tv[2], tv[3]);
      pt[1] = make_float4(tv[4], tv[5], tv[6], tv[7]);
      ps[0] = make_float4(sv[0], sv[1], sv[2], sv[3]);
      ps[1] = make_float4(sv[4], sv[5], sv[6], sv[7]);
      pc[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
      pc[1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
    }
    // space_carving_kernel, :253-276: min |tsdf| over the block after the update
    float m = fabsf(tv[0]);
#pragma unroll
    for (int i = 1; i < 8; ++i) m = fminf(m, fabsf(tv[i]));
    m = wave_min(m);
    if (lane == 0) carve_flag[b] = (m >= .9f) ? 1 : 0;
    updated_total += nupd;
  }
  updated_total = wave_sum(updated_total);
  if (lane == 0 && updated_total) atomicAdd(&ctl->n_updated, updated_total);
}

// ---- carve pass ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_carve_mark(Table tab, const VisItem* vis,
                                                    const uint8_t* carve_flag, uint32_t* bitmap,
                                                    int32_t* del_idx, SlowDelete* slow,
                                                    uint32_t slow_cap, Ctl* ctl) {
  const uint32_t nv = ctl->n_vis;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += gridDim.x * blockDim.x) {
    if (!carve_flag[i]) continue;
    const VisItem it = vis[i];
    const uint32_t bucket = block_hash(it.x, it.y, it.z, tab.bucket_mask);
    if (it.entry == (bucket << 1)) {
      // slot 0 of the home bucket: no lock involved (voxel_hash.cu:114-123)
      uint32_t* pe = reinterpret_cast<uint32_t*>(tab.entries + it.entry);
      pe[1] = pe[1] & 0xFFFFu;  // offset = 0
      pe[2] = (uint32_t)-1;
      del_idx[i] = it.idx;
      atomicOr(&bitmap[i >> 5], 1u << (i & 31));
    } else {
      atomicMin(&tab.claim[bucket], i);
      const uint32_t slot = atomicAdd(&ctl->n_slow_del, 1u);
      if (slot < slow_cap) {
        slow[slot] = SlowDelete{it.x, it.y, it.z, 0, i};
      } else {
        set_error(ctl, RATSDF_ERR_CAPACITY);
      }
    }
  }
}

// explicit delete list (test hook): builds a pseudo visible list in list order
__global__ void k_lookup_list(Table tab, const int16_t* pos, int n, VisItem* vis,
                              uint8_t* carve_flag, Ctl* ctl) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) ctl->n_vis = (uint32_t)n;
  if (i >= n) return;
  EntryWords w;
  const int x = pos[3 * i], y = pos[3 * i + 1], z = pos[3 * i + 2];
  const uint32_t e = find_block(tab, x, y, z, &w);
  vis[i] = VisItem{(int16_t)x, (int16_t)y, (int16_t)z, (int16_t)(w.w1 >> 16), w.idx,
                   e == kInf ? 0u : e};
  carve_flag[i] = e != kInf;
}

// One workgroup: (a) head / chain deletes, one winner per home bucket (the first in list order),
// (b) popcount prefix of the delete bitmap, free-list bookkeeping and the frame's statistics.
__global__ __launch_bounds__(1024) void k_carve_scan(Table tab, SlowDelete* slow,
                                                     uint32_t slow_cap, uint32_t* bitmap,
                                                     uint32_t* prefix, int32_t* del_idx,
                                                     uint32_t* next_bitmap, uint32_t next_words,
                                                     Ctl* ctl, ratsdf_frame_stats* stats) {
  __shared__ uint32_t lds[1024];
  uint32_t ns = ctl->n_slow_del;
  if (ns > slow_cap) ns = slow_cap;
  // pass 1: decide every winner before any claim is released (items of one bucket may sit in
  // different strides of the loop); the flag is parked in the item itself (same thread re-reads it)
  for (uint32_t j = threadIdx.x; j < ns; j += blockDim.x) {
    const SlowDelete s = slow[j];
    const uint32_t bucket = block_hash(s.x, s.y, s.z, tab.bucket_mask);
    slow[j].pad = (tab.claim[bucket] == s.vis) ? 1 : 0;
  }
  __syncthreads();
  for (uint32_t j = threadIdx.x; j < ns; j += blockDim.x) {
    const SlowDelete s = slow[j];
    const uint32_t bucket = block_hash(s.x, s.y, s.z, tab.bucket_mask);
    const bool win = s.pad != 0;
    tab.claim[bucket] = kInf;  // ResetLocks
    if (win) {
      const uint32_t k0 = key0(s.x, s.y), k1 = key1(s.z);
      uint32_t last = (bucket << 1) + 1;
      uint32_t* ph = reinterpret_cast<uint32_t*>(tab.entries + last);
      EntryWords h = load_entry(tab.entries, last);
      if (entry_matches(h, k0, k1)) {                                     // voxel_hash.cu:125-140
        const uint32_t nxt = (last + (uint32_t)entry_offset(h)) & tab.entry_mask;
        uint32_t* pn = reinterpret_cast<uint32_t*>(tab.entries + nxt);
        const EntryWords nw = load_entry(tab.entries, nxt);
        del_idx[s.vis] = h.idx;
        const int noff = entry_offset(nw);
        const int16_t hoff = noff ? (int16_t)(entry_offset(h) + noff) : (int16_t)0;
        ph[0] = nw.w0;
        ph[1] = (nw.w1 & 0xFFFFu) | ((uint32_t)(uint16_t)hoff << 16);
        ph[2] = (uint32_t)nw.idx;
        pn[1] = pn[1] & 0xFFFFu;
        pn[2] = (uint32_t)-1;
        atomicOr(&bitmap[s.vis >> 5], 1u << (s.vis & 31));
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
            del_idx[s.vis] = cw.idx;
            pcur[1] = pcur[1] & 0xFFFFu;
            pcur[2] = (uint32_t)-1;
            atomicOr(&bitmap[s.vis >> 5], 1u << (s.vis & 31));
            break;
          }
          last = cur;
        }
      }
    }
  }
  __syncthreads();
  const uint32_t nv = ctl->n_vis;
  const uint32_t nwords = (nv + 31) >> 5;
  const uint32_t total = bitmap_prefix_scan<true>(bitmap, prefix, nwords, lds);
  for (uint32_t w = threadIdx.x; w < next_words; w += blockDim.x) next_bitmap[w] = 0;
  if (threadIdx.x == 0) {
    const int32_t nf = ctl->num_free;
    ctl->free_base = (uint32_t)nf;
    ctl->n_del = total;
    ctl->num_free = nf + (int32_t)total;
    if (stats) {
      stats->visible_blocks = (int32_t)nv;
      stats->updated_voxels = (int32_t)ctl->n_updated;
      stats->allocated_blocks = (int32_t)ctl->n_win;
      stats->deleted_blocks = (int32_t)total;
      stats->active_blocks = tab.num_block - (nf + (int32_t)total);
      stats->slow_requests = (int32_t)ctl->n_slow;
      ctl->totals[0] += 1;
      ctl->totals[1] += nv;
      ctl->totals[2] += ctl->n_updated;
      ctl->totals[3] += ctl->n_win;
      ctl->totals[4] += total;
    }
  }
}

// ReleaseBlock in list order: heap[free_base + k] = idx, voxel_mem.cu:56-61
__global__ __launch_bounds__(256) void k_carve_commit(Pool pool, const uint32_t* bitmap,
                                                      const uint32_t* prefix, const int32_t* del_idx,
                                                      Ctl* ctl) {
  const uint32_t nv = ctl->n_vis;
  const uint32_t base = ctl->free_base;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += gridDim.x * blockDim.x) {
    const uint32_t word = bitmap[i >> 5];
    if (!((word >> (i & 31)) & 1u)) continue;
    const uint32_t k = prefix[i >> 5] + __popc(word & ((1u << (i & 31)) - 1u));
    pool.heap[base + k] = del_idx[i];
  }
}

}  // namespace ratsdf
