// kernels_raycast.h -- virtual-view ray casting of the TSDF map for gfx950.
//
// Replaces ray_cast_kernel + VoxelHashTable::Retrieve / RetrieveTSDF
// (utils/tsdf/voxel_tsdf.cu:278-374, voxel_hash.cuh:104-143, voxel_hash.cu:161-188).  One lane per
// pixel, a wave covers a 16x4 pixel tile so neighbouring rays walk the same voxel blocks and their
// directory probes and voxel reads share cache lines; every lane keeps the reference's one-entry
// block cache (the last directory lookup) so consecutive samples inside a block cost no probe.
// Output goes to plain device buffers (uchar4 per pixel) instead of CUDA-GL interop textures.
//
// The march is a chain of dependent memory round trips: a sample's tsdf decides the step to the next sample, whose
// address (directory probe, then the voxel) is only known then.  What round 5 measured (profiles/r05_raycast.txt;
// per-wave records of the diagnostic build, tools/raycast_probe.py): a 640x480 rendering is 130 us of vector issue
// but took 450 - 520 us; half of the waves are through after 90 us; the kernel lasts as long as its slowest waves,
// and those hold a ray that runs its FULL length (534 samples: it leaves through a gap, or grazes a surface), every
// sample of it a directory probe that finds nothing -- 64 % of such a lane's cycles were probes, one after the other
// (find_block branches on what it loaded, so four probes in a row are four round trips of ~0.7 us), 32 % the judging
// code, 5 % the voxel loads.  Hence:
//   * occupancy in LDS, two levels: k_occupancy_build hashes every live block into a 16 KiB Bloom filter (two bits per
//     block) and its super-cell of 4^3 blocks into a 4 KiB bitmap before the rendering (a scan of Table::active), every
//     workgroup copies both into LDS.  A clear cell bit proves the whole cell empty: its samples cost a compare of the
//     cell's coordinates and the three additions of the step.  Inside an occupied cell a clear block bit proves the block
//     absent: an LDS read instead of a directory probe (set bits may be collisions: probed);
//   * the zero crossing's interpolation, colour and normal (:318-357, two thirds of the loop's code) run once, after
//     the loop, not as a branch inside it;
//   * the three divisions of the full step by 10 (:361-367, at every sample that picks the fine step) are done once, and
//     (short)roundf() takes its four-instruction form.
// Tried and dropped, with their numbers in the same file: skipping AHEAD inside an empty cell (a conservative count of
// the samples that stay in it, their positions still produced by the same additions in a tight loop: half the
// iterations, slower every time -- the count costs more than the samples it saves), and fetching the next 2 / 3 /
// 4 / 8 samples speculatively under the assumption that the step does not change (depth 1, i.e. none, is fastest: the
// lookups that remain are LDS reads, and a group's extra samples only lengthen the slowest wave's instruction stream).
#pragma once
#include "kernels_integrate.h"

namespace ratsdf {

struct BlockCache {  // VoxelBlock cache of RetrieveMutable, voxel_hash.cuh:124-143
  int bx, by, bz;
  int32_t idx;    // >= 0: pool block; -1 with miss == true: known absent
  bool valid;
};

// hashed occupancy of the map's blocks, built before a rendering and held in LDS (see the header)
// (TWO bits per block, a Bloom filter: the 64 rays of a wave enter ~30 blocks per sample, so with one bit per block and
// 2 % of the bits set some lane runs into a collision every other sample and sends the whole wave to the directory)
constexpr uint32_t kOccWords = 4096;               // blocks: 16 KiB = 131 072 bits
constexpr uint32_t kCellWords = 1024;              // super-cells of 4^3 blocks (32^3 voxels), one bit each: 4 KiB
#ifndef RATSDF_CELL_BITS
#define RATSDF_CELL_BITS 5  // (same-box A/B of 16 / 32 / 64 / 128 voxels: 269 / 201 / 206 / 281 us per rendering)
#endif
constexpr int kCellVoxelBits = RATSDF_CELL_BITS;
// (any spreading of neighbouring blocks will do: 24-bit multiply-adds of the coordinates' low 16 bits -- the directory
// hash's three full 32-bit multiplications issue at a quarter of the rate, and a lookup is on the march's critical path)
__device__ inline uint32_t occ_bit(int bx, int by, int bz) {
  const uint32_t x = (uint32_t)bx & 0xFFFFu, y = (uint32_t)by & 0xFFFFu, z = (uint32_t)bz & 0xFFFFu;
  return (__umul24(x, 0x9E5u) + __umul24(y, 0x1F35Bu) + __umul24(z, 0x6A7C1u)) & (kOccWords * 32u - 1u);
}
__device__ inline uint32_t occ_bit2(int bx, int by, int bz) {
  const uint32_t x = (uint32_t)bx & 0xFFFFu, y = (uint32_t)by & 0xFFFFu, z = (uint32_t)bz & 0xFFFFu;
  return ((__umul24(x, 0x2C1B3u) + __umul24(y, 0x5D3u) + __umul24(z, 0x1B873u)) >> 3) & (kOccWords * 32u - 1u);
}
__device__ inline uint32_t cell_bit(int cx, int cy, int cz) {
  const uint32_t x = (uint32_t)cx & 0xFFFFu, y = (uint32_t)cy & 0xFFFFu, z = (uint32_t)cz & 0xFFFFu;
  return (__umul24(x, 0x9E5u) + __umul24(y, 0x1F35Bu) + __umul24(z, 0x6A7C1u)) & (kCellWords * 32u - 1u);
}
__device__ inline bool occ_maybe(const uint32_t* occ, int bx, int by, int bz) {
  const uint32_t h = occ_bit(bx, by, bz), g = occ_bit2(bx, by, bz);
  return (((occ[h >> 5] >> (h & 31u)) & (occ[g >> 5] >> (g & 31u))) & 1u) != 0u;
}

// pool voxel index of integer voxel (px,py,pz) or -1 (block absent).  `occ`: the occupancy bits (LDS) -- a clear bit
// proves the block absent and saves the directory probe.
__device__ inline long voxel_index(const Table& tab, int px, int py, int pz, BlockCache& c, const uint32_t* occ) {
  const int bx = px >> 3, by = py >> 3, bz = pz >> 3;
  if (!(c.valid && c.bx == bx && c.by == by && c.bz == bz)) {
    int32_t idx = -1;
    if (occ_maybe(occ, bx, by, bz)) {
      EntryWords w;
      const uint32_t e = find_block(tab, bx, by, bz, &w);
      idx = e == kInf ? -1 : w.idx;
    }
    c.bx = bx;
    c.by = by;
    c.bz = bz;
    c.idx = idx;
    c.valid = true;
  }
  if (c.idx < 0) return -1;
  return ((long)c.idx << 9) + ((px & 7) + (py & 7) * 8 + (pz & 7) * 64);
}

__device__ inline float tsdf_at(const Table& tab, const Pool& pool, int px, int py, int pz,
                                BlockCache& c, const uint32_t* occ) {
  const long vi = voxel_index(tab, px, py, pz, c, occ);
  return vi >= 0 ? pool.tsdf[vi] : -10.f;  // VoxelTSDF(): -10, voxel_types.cu:8
}

// VoxelHashTable::RetrieveTSDF, voxel_hash.cu:161-188 (corner / weight pairing as written there)
__device__ inline float retrieve_tsdf(const Table& tab, const Pool& pool, const V3& pt,
                                      BlockCache& c, const uint32_t* occ) {
  const V3 pl{floorf(pt.x), floorf(pt.y), floorf(pt.z)};
  const V3 ph{pl.x + 1.f, pl.y + 1.f, pl.z + 1.f};
  const V3 al{ph.x - pt.x, ph.y - pt.y, ph.z - pt.z};
  float t[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int cx = (int16_t)f2i((i >> 2) & 1 ? pl.x : ph.x);
    const int cy = (int16_t)f2i((i >> 1) & 1 ? pl.y : ph.y);
    const int cz = (int16_t)f2i((i >> 0) & 1 ? pl.z : ph.z);
    t[i] = tsdf_at(tab, pool, cx, cy, cz, c, occ);
  }
  const float t00 = t[0] * al.z + t[1] * (1 - al.z);
  const float t01 = t[2] * al.z + t[3] * (1 - al.z);
  const float t10 = t[4] * al.z + t[5] * (1 - al.z);
  const float t11 = t[6] * al.z + t[7] * (1 - al.z);
  const float t0 = t00 * al.y + t01 * (1 - al.y);
  const float t1 = t10 * al.y + t11 * (1 - al.y);
  return t0 * al.x + t1 * (1 - al.x);
}

// the occupancy bits (kOccWords + kCellWords words, zeroed by the caller) from the live blocks: one lane per slot of Table::active
// that has ever been in use
__global__ __launch_bounds__(256) void k_occupancy_build(Table tab, const Ctl* ctl, uint32_t* bits) {
  int32_t lo = ctl->free_low;
  if (lo < 0) lo = 0;
  const uint32_t n = (uint32_t)(tab.num_block - lo);
  for (uint32_t i = blockIdx.x * block_threads() + threadIdx.x; i < n; i += gridDim.x * block_threads()) {
    const uint4 a = reinterpret_cast<const uint4*>(tab.active)[(uint32_t)lo + i];  // {x | y << 16, z, idx, entry}
    if ((int32_t)a.z < 0) continue;  // the slot is empty
    const int bx = (int16_t)(a.x & 0xFFFFu), by = (int16_t)(a.x >> 16), bz = (int16_t)(a.y & 0xFFFFu);
    const uint32_t h = occ_bit(bx, by, bz), g = occ_bit2(bx, by, bz);
    atomicOr(&bits[h >> 5], 1u << (h & 31u));
    atomicOr(&bits[g >> 5], 1u << (g & 31u));
    const uint32_t c = cell_bit(bx >> (kCellVoxelBits - 3), by >> (kCellVoxelBits - 3), bz >> (kCellVoxelBits - 3));
    atomicOr(&bits[kOccWords + (c >> 5)], 1u << (c & 31u));
  }
}

__global__ __launch_bounds__(256) void k_raycast(Table tab, Pool pool, FrameParams P,
                                                 float step_size, int max_step, uint32_t* out_rgba,
                                                 uint32_t* out_normal, int row0, int row1, const uint32_t* occ_bits,
                                                 Ctl* ctl) {
  // the occupancy bits in LDS (every wave of the workgroup takes part, whatever its pixels)
  __shared__ __attribute__((aligned(16))) uint32_t occ[kOccWords + kCellWords];
  {
    const uint4* src = reinterpret_cast<const uint4*>(occ_bits);
    uint4* dst = reinterpret_cast<uint4*>(occ);
    for (uint32_t i = threadIdx.x; i < (kOccWords + kCellWords) / 4; i += 256) dst[i] = src[i];
  }
  __syncthreads();
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
  // (:361-367 divides the full step by 10 at every sample that chooses the fine step: the same three quotients every
  // time -- thirty instructions of IEEE division per sample, here once)
  const V3 fine_step{full.x / 10, full.y / 10, full.z / 10};
  V3 stepv = full;
  V3 p{P.Ti.t.x / P.vs, P.Ti.t.y / P.vs, P.Ti.t.z / P.vs};                    // :299
  BlockCache cache{0, 0, 0, -1, false};
  // (short)roundf(v): the four-instruction form (device_math.h: round_to_int) wherever the ray can get -- it equals the
  // long one for |v| < 2^31; one decision for the wave
  const float reach = (float)max_step;
  const bool small = __all(fabsf(p.x) + reach * fabsf(full.x) < 1e9f && fabsf(p.y) + reach * fabsf(full.y) < 1e9f &&
                           fabsf(p.z) + reach * fabsf(full.z) < 1e9f);  // uniform
  auto gi = [small](float v) { return small ? (int)(int16_t)round_to_int(v) : (int)(int16_t)f2i(roundf(v)); };
  float prev = tsdf_at(tab, pool, gi(p.x), gi(p.y), gi(p.z), cache, occ);           // :302-303
  p = V3{p.x + stepv.x, p.y + stepv.y, p.z + stepv.z};
  bool done = false;
#ifdef RATSDF_STAMPS
  // per-wave record (tools/raycast_probe.py): [0] start, [2] end (10 ns ticks); [3] / [4] samples of the slowest lane,
  // [5] lanes that hit; a full-length lane's cycles: [1] stepping + judging, [6] lookups, [7] voxel loads
  unsigned long long* ws = ctl && ctl->debug_buf
      ? ctl->debug_buf + (size_t)((((blockIdx.y * gridDim.x + blockIdx.x) * 4u + (threadIdx.x >> 6)) & 16383u)) * 8 : nullptr;
  if (ws) atomicMin(&ws[0], wall_clock64());
  unsigned long long ph[4] = {0, 0, 0, 0};
#define RC_STAMP(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = clock64(); ph[k] += t_ - t_last; t_last = t_; } while (0)
  unsigned long long t_last = clock64();
#else
#define RC_STAMP(k) do { } while (0)
#endif
  int i = 1;
  int cell_x = 0, cell_y = 0, cell_z = 0;
  bool cell_known = false, cell_empty = false;
  for (; i < max_step; ++i) {                                                 // :305
    RC_STAMP(0);
    const int gx = gi(p.x), gy = gi(p.y), gz = gi(p.z);
    {  // a sample inside a super-cell that provably holds no block reads "no voxel" whatever its block (:312-316: it is
       // skipped, the step stays as it is): no hash, no LDS read while the ray stays in the cell
      const int cx = gx >> kCellVoxelBits, cy = gy >> kCellVoxelBits, cz = gz >> kCellVoxelBits;
      if (!(cell_known && cx == cell_x && cy == cell_y && cz == cell_z)) {
        const uint32_t h = cell_bit(cx, cy, cz);
        cell_empty = ((occ[kOccWords + (h >> 5)] >> (h & 31u)) & 1u) == 0u;
        cell_x = cx;
        cell_y = cy;
        cell_z = cz;
        cell_known = true;
      }
      if (cell_empty) {
        p = V3{p.x + stepv.x, p.y + stepv.y, p.z + stepv.z};
        prev = -10.f;
        continue;
      }
    }
    const long vi = voxel_index(tab, gx, gy, gz, cache, occ);
    RC_STAMP(1);
    const float cur = vi >= 0 ? pool.tsdf[vi] : -10.f;
    const uint32_t wcur = vi >= 0 ? (pool.rgbw[vi] >> 24) : 0u;               // :308-309
    RC_STAMP(2);
    if (wcur < 10) {                                                          // :312-316
      p = V3{p.x + stepv.x, p.y + stepv.y, p.z + stepv.z};
      prev = cur;
      continue;
    }
    if (prev > 0 && cur <= 0 && prev - cur <= 2.0f) {                         // :318: the zero crossing (below)
      done = true;
      break;
    }
    prev = cur;
    stepv = cur < 0.5f ? fine_step : full;                                    // :361-367
    p = V3{p.x + stepv.x, p.y + stepv.y, p.z + stepv.z};
  }
  RC_STAMP(3);
  if (done) {  // :318-357 at the crossing sample: p and the step are that sample's
    const V3 p1{p.x - stepv.x, p.y - stepv.y, p.z - stepv.z};
    const float ac = retrieve_tsdf(tab, pool, p, cache, occ);                 // :323-324
    const float ap = retrieve_tsdf(tab, pool, p1, cache, occ);
    const float f = ac / (ap - ac);                                           // :327-328
    const V3 pi{p.x + f * stepv.x, p.y + f * stepv.y, p.z + f * stepv.z};
    const int fx = gi(pi.x), fy = gi(pi.y), fz = gi(pi.z);                    // :329-330
    const long fi = voxel_index(tab, fx, fy, fz, cache, occ);
    const uint32_t c = fi >= 0 ? pool.rgbw[fi] : 0u;                          // :333-334
    const float prob = fi >= 0 ? pool.segm[fi] : 0.f;
    auto at = [&](int dx, int dy, int dz) {
      return tsdf_at(tab, pool, (int16_t)(fx + dx), (int16_t)(fy + dy), (int16_t)(fz + dz), cache, occ);
    };
    const V3 nr{at(1, 0, 0) - at(-1, 0, 0), at(0, 1, 0) - at(0, -1, 0),
                at(0, 0, 1) - at(0, 0, -1)};                                  // :337-348
    const float dotv = nr.x * (-dw.x) + (nr.y * (-dw.y) + nr.z * (-dw.z));
    const float nn = sqrtf(nr.x * nr.x + (nr.y * nr.y + nr.z * nr.z));
    const float diff = fmaxf(dotv / nn, 0);                                   // :349
    const float alpha = fmaxf(prob - 0.5f, 0) / .5f;                          // :350
    const float cr = (float)(c & 0xFFu), cg = (float)((c >> 8) & 0xFFu),
                cb = (float)((c >> 16) & 0xFFu);
    o_c = ((uint32_t)f2i(alpha * 255 + (1 - alpha) * cr) & 0xFFu) |
          (((uint32_t)f2i((1 - alpha) * cg) & 0xFFu) << 8) |
          (((uint32_t)f2i((1 - alpha) * cb) & 0xFFu) << 16) | 0xFF000000u;    // :351-353
    const uint32_t n0 = (uint32_t)f2i(alpha * 255 + (1 - alpha) * diff * 255) & 0xFFu;
    const uint32_t n1 = (uint32_t)f2i((1 - alpha) * diff * 255) & 0xFFu;
    o_n = n0 | (n1 << 8) | (n1 << 16) | 0xFF000000u;                          // :354-356
  }
#ifdef RATSDF_STAMPS
  if (ws) {
    atomicMax(&ws[2], wall_clock64());
    atomicMax(&ws[3], (unsigned long long)i);
    atomicMax(&ws[4], (unsigned long long)i);
    if (done) atomicAdd(&ws[5], 1ull);
    if (i >= max_step) {  // a ray that ran its full length: cycles in positions | probes | voxel loads | judging
      ws[6] = ph[1];
      ws[7] = ph[2];
      ws[1] = ph[0] + ph[3];
    }
  }
#endif
  if (out_rgba) out_rgba[idx] = o_c;
  if (out_normal) out_normal[idx] = o_n;
}

}  // namespace ratsdf
