// kernels_raycast.h -- virtual-view ray casting of the TSDF map for gfx950.
//
// Replaces ray_cast_kernel + VoxelHashTable::Retrieve / RetrieveTSDF
// (utils/tsdf/voxel_tsdf.cu:278-374, voxel_hash.cuh:104-143, voxel_hash.cu:161-188).  One lane per
// pixel, a wave covers a 16x4 pixel tile so neighbouring rays walk the same voxel blocks and their
// directory probes and voxel reads share cache lines; every lane keeps the reference's one-entry
// block cache (the last directory lookup) so consecutive samples inside a block cost no probe.
// Output goes to plain device buffers (uchar4 per pixel) instead of CUDA-GL interop textures.
#pragma once
#include "kernels_integrate.h"

namespace ratsdf {

struct BlockCache {  // VoxelBlock cache of RetrieveMutable, voxel_hash.cuh:124-143
  int bx, by, bz;
  int32_t idx;    // >= 0: pool block; -1 with miss == true: known absent
  bool valid;
};

// pool voxel index of integer voxel (px,py,pz) or -1 (block absent)
__device__ inline long voxel_index(const Table& tab, int px, int py, int pz, BlockCache& c) {
  const int bx = px >> 3, by = py >> 3, bz = pz >> 3;
  if (!(c.valid && c.bx == bx && c.by == by && c.bz == bz)) {
    EntryWords w;
    const uint32_t e = find_block(tab, bx, by, bz, &w);
    c.bx = bx;
    c.by = by;
    c.bz = bz;
    c.idx = e == kInf ? -1 : w.idx;
    c.valid = true;
  }
  if (c.idx < 0) return -1;
  return ((long)c.idx << 9) + ((px & 7) + (py & 7) * 8 + (pz & 7) * 64);
}

__device__ inline float tsdf_at(const Table& tab, const Pool& pool, int px, int py, int pz,
                                BlockCache& c) {
  const long vi = voxel_index(tab, px, py, pz, c);
  return vi >= 0 ? pool.tsdf[vi] : -10.f;  // VoxelTSDF(): -10, voxel_types.cu:8
}

// VoxelHashTable::RetrieveTSDF, voxel_hash.cu:161-188 (corner / weight pairing as written there)
__device__ inline float retrieve_tsdf(const Table& tab, const Pool& pool, const V3& pt,
                                      BlockCache& c) {
  const V3 pl{floorf(pt.x), floorf(pt.y), floorf(pt.z)};
  const V3 ph{pl.x + 1.f, pl.y + 1.f, pl.z + 1.f};
  const V3 al{ph.x - pt.x, ph.y - pt.y, ph.z - pt.z};
  float t[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int cx = (int16_t)f2i((i >> 2) & 1 ? pl.x : ph.x);
    const int cy = (int16_t)f2i((i >> 1) & 1 ? pl.y : ph.y);
    const int cz = (int16_t)f2i((i >> 0) & 1 ? pl.z : ph.z);
    t[i] = tsdf_at(tab, pool, cx, cy, cz, c);
  }
  const float t00 = t[0] * al.z + t[1] * (1 - al.z);
  const float t01 = t[2] * al.z + t[3] * (1 - al.z);
  const float t10 = t[4] * al.z + t[5] * (1 - al.z);
  const float t11 = t[6] * al.z + t[7] * (1 - al.z);
  const float t0 = t00 * al.y + t01 * (1 - al.y);
  const float t1 = t10 * al.y + t11 * (1 - al.y);
  return t0 * al.x + t1 * (1 - al.x);
}

__global__ __launch_bounds__(256) void k_raycast(Table tab, Pool pool, FrameParams P,
                                                 float step_size, int max_step, uint32_t* out_rgba,
                                                 uint32_t* out_normal, int row0, int row1) {
  // rows [row0, row1) of the P.H x P.W image (the whole image: 0, P.H); the output buffers hold those rows only
  const int x = blockIdx.x * 16 + (threadIdx.x & 15);
  const int y = row0 + blockIdx.y * 16 + (threadIdx.x >> 4);
  if (x >= P.W || y >= row1) return;
  const int idx = (y - row0) * P.W + x;
  uint32_t o_c = 0, o_n = 0;
  const V3 pc = intr_mul(P.Ki, V3{(float)x, (float)y, 1.f});                 // :289-290
  const float n2 = pc.x * pc.x + (pc.y * pc.y + pc.z * pc.z);
  V3 dc = pc;                                                                 // normalized(), :291
  if (n2 > 0.f) {
    const float nn = sqrtf(n2);
    dc = V3{pc.x / nn, pc.y / nn, pc.z / nn};
  }
  const V3 dw = quat_rotate(P.Ti.q, dc);                                      // :292-293
  const V3 full{dw.x * step_size / P.vs, dw.y * step_size / P.vs, dw.z * step_size / P.vs};  // :297
  V3 stepv = full;
  V3 p{P.Ti.t.x / P.vs, P.Ti.t.y / P.vs, P.Ti.t.z / P.vs};                    // :299
  BlockCache cache{0, 0, 0, -1, false};
  auto gi = [](float v) { return (int)(int16_t)f2i(roundf(v)); };
  float prev = tsdf_at(tab, pool, gi(p.x), gi(p.y), gi(p.z), cache);           // :302-303
  p = V3{p.x + stepv.x, p.y + stepv.y, p.z + stepv.z};
  for (int i = 1; i < max_step; ++i) {                                        // :305
    const int gx = gi(p.x), gy = gi(p.y), gz = gi(p.z);
    const long vi = voxel_index(tab, gx, gy, gz, cache);
    const float cur = vi >= 0 ? pool.tsdf[vi] : -10.f;
    const uint32_t wcur = vi >= 0 ? (pool.rgbw[vi] >> 24) : 0u;               // :308-309
    if (wcur < 10) {                                                          // :312-316
      p = V3{p.x + stepv.x, p.y + stepv.y, p.z + stepv.z};
      prev = cur;
      continue;
    }
    if (prev > 0 && cur <= 0 && prev - cur <= 2.0f) {                         // :318
      const V3 p1{p.x - stepv.x, p.y - stepv.y, p.z - stepv.z};
      const float ac = retrieve_tsdf(tab, pool, p, cache);                    // :323-324
      const float ap = retrieve_tsdf(tab, pool, p1, cache);
      const float f = ac / (ap - ac);                                         // :327-328
      const V3 pi{p.x + f * stepv.x, p.y + f * stepv.y, p.z + f * stepv.z};
      const int fx = gi(pi.x), fy = gi(pi.y), fz = gi(pi.z);                  // :329-330
      const long fi = voxel_index(tab, fx, fy, fz, cache);
      const uint32_t c = fi >= 0 ? pool.rgbw[fi] : 0u;                        // :333-334
      const float prob = fi >= 0 ? pool.segm[fi] : 0.f;
      auto at = [&](int dx, int dy, int dz) {
        return tsdf_at(tab, pool, (int16_t)(fx + dx), (int16_t)(fy + dy), (int16_t)(fz + dz), cache);
      };
      const V3 nr{at(1, 0, 0) - at(-1, 0, 0), at(0, 1, 0) - at(0, -1, 0),
                  at(0, 0, 1) - at(0, 0, -1)};                                // :337-348
      const float dotv = nr.x * (-dw.x) + (nr.y * (-dw.y) + nr.z * (-dw.z));
      const float nn = sqrtf(nr.x * nr.x + (nr.y * nr.y + nr.z * nr.z));
      const float diff = fmaxf(dotv / nn, 0);                                 // :349
      const float alpha = fmaxf(prob - 0.5f, 0) / .5f;                        // :350
      const float cr = (float)(c & 0xFFu), cg = (float)((c >> 8) & 0xFFu),
                  cb = (float)((c >> 16) & 0xFFu);
      o_c = ((uint32_t)f2i(alpha * 255 + (1 - alpha) * cr) & 0xFFu) |
            (((uint32_t)f2i((1 - alpha) * cg) & 0xFFu) << 8) |
            (((uint32_t)f2i((1 - alpha) * cb) & 0xFFu) << 16) | 0xFF000000u;  // :351-353
      const uint32_t n0 = (uint32_t)f2i(alpha * 255 + (1 - alpha) * diff * 255) & 0xFFu;
      const uint32_t n1 = (uint32_t)f2i((1 - alpha) * diff * 255) & 0xFFu;
      o_n = n0 | (n1 << 8) | (n1 << 16) | 0xFF000000u;                        // :354-356
      break;
    }
    prev = cur;
    if (cur < 0.5f) {                                                         // :361-367
      stepv = V3{full.x / 10, full.y / 10, full.z / 10};
    } else {
      stepv = full;
    }
    p = V3{p.x + stepv.x, p.y + stepv.y, p.z + stepv.z};
  }
  if (out_rgba) out_rgba[idx] = o_c;
  if (out_normal) out_normal[idx] = o_n;
}

}  // namespace ratsdf
