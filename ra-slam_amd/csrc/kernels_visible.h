// kernels_visible.h -- ordered compaction of the hash directory for gfx950.
//
// Replaces check_visibility_kernel / check_valid_kernel / check_bound_kernel + prefix_sum +
// gather_visible_blocks_kernel + the blocking read-back of the count
// (utils/tsdf/voxel_tsdf.cu:15-33,98-118,465-474,847-867; utils/cuda/arithmetic.cuh:52-172).
// The output list is ordered by ascending hash-entry index exactly like the reference's
// scan + gather, but is built from per-wave ballots: flags kernel (ballot mask per wave + count per
// workgroup), one-workgroup scan of the 2^22/1024 workgroup counts, scatter kernel.  The count
// stays on the device (Ctl::n_vis / n_sel); consumers are persistent grids that read it there.
#pragma once
#include "kernels_alloc.h"

namespace ratsdf {

enum SelectMode { kSelVisible = 0, kSelValid = 1, kSelBounds = 2 };

struct GridBounds {  // BoundingCube<short>, voxel_tsdf.cuh:19-34
  int16_t xmin, xmax, ymin, ymax, zmin, zmax;
};

constexpr int kSelWG = 1024;  // entries per workgroup

template <int Mode>
__global__ __launch_bounds__(kSelWG) void k_select_flags(Table tab, FrameParams P, GridBounds gb,
                                                         unsigned long long* masks,
                                                         uint32_t* wg_count) {
  __shared__ uint32_t wave_cnt[kSelWG / 64];
  const uint32_t e = blockIdx.x * kSelWG + threadIdx.x;
  const EntryWords w = load_entry(tab.entries, e);
  bool sel = false;
  if (w.idx >= 0) {
    const int bx = (int16_t)(w.w0 & 0xFFFFu), by = (int16_t)(w.w0 >> 16),
              bz = (int16_t)(w.w1 & 0xFFFFu);
    if (Mode == kSelVisible) {
      sel = block_visible<false>(bx, by, bz, P);                          // voxel_tsdf.cu:98-109
    } else if (Mode == kSelValid) {
      sel = true;                                                         // voxel_tsdf.cu:28-33
    } else {
      const int gx = (int16_t)(bx << 3), gy = (int16_t)(by << 3), gz = (int16_t)(bz << 3);
      sel = gx >= gb.xmin && gy >= gb.ymin && gz >= gb.zmin && gx + 8 - 1 <= gb.xmax &&
            gy + 8 - 1 <= gb.ymax && gz + 8 - 1 <= gb.zmax;               // voxel_tsdf.cu:15-26
    }
  }
  const unsigned long long m = __ballot(sel);
  const uint32_t wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    masks[blockIdx.x * (kSelWG / 64) + wv] = m;
    wave_cnt[wv] = __popcll(m);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < kSelWG / 64; ++i) s += wave_cnt[i];
    wg_count[blockIdx.x] = s;
  }
}

// one workgroup: exclusive scan of the workgroup counts; total -> *total_out
__global__ __launch_bounds__(1024) void k_select_scan(const uint32_t* wg_count, uint32_t* wg_offset,
                                                      uint32_t nwg, uint32_t* total_out) {
  __shared__ uint32_t lds[1024];
  const uint32_t tid = threadIdx.x;
  const uint32_t chunk = (nwg + 1023) / 1024;
  const uint32_t lo = tid * chunk;
  const uint32_t hi = lo + chunk < nwg ? lo + chunk : nwg;
  uint32_t sum = 0;
  for (uint32_t i = lo; i < hi; ++i) sum += wg_count[i];
  lds[tid] = sum;
  __syncthreads();
  for (uint32_t d = 1; d < 1024; d <<= 1) {
    const uint32_t v = tid >= d ? lds[tid - d] : 0;
    __syncthreads();
    lds[tid] += v;
    __syncthreads();
  }
  uint32_t run = lds[tid] - sum;
  for (uint32_t i = lo; i < hi; ++i) {
    wg_offset[i] = run;
    run += wg_count[i];
  }
  if (tid == 1023) *total_out = lds[1023];
}

// scatter selected entries, in entry order, as 16-byte items {entry copy, entry index}
__global__ __launch_bounds__(kSelWG) void k_select_scatter(Table tab,
                                                           const unsigned long long* masks,
                                                           const uint32_t* wg_count,
                                                           const uint32_t* wg_offset, VisItem* out,
                                                           uint32_t out_cap) {
  if (wg_count[blockIdx.x] == 0) return;
  const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned long long* wm = masks + blockIdx.x * (kSelWG / 64);
  const unsigned long long m = wm[wv];
  if (!((m >> lane) & 1ull)) return;
  uint32_t pos = wg_offset[blockIdx.x];
  for (uint32_t i = 0; i < wv; ++i) pos += __popcll(wm[i]);
  pos += __popcll(m & ((1ull << lane) - 1ull));
  if (pos >= out_cap) return;
  const uint32_t e = blockIdx.x * kSelWG + threadIdx.x;
  const EntryWords w = load_entry(tab.entries, e);
  uint4 v;
  v.x = w.w0;
  v.y = w.w1;
  v.z = (uint32_t)w.idx;
  v.w = e;
  reinterpret_cast<uint4*>(out)[pos] = v;
}

// download_tsdf_kernel / download_semantic_kernel, voxel_tsdf.cu:35-62: one wave per selected
// block, lane l writes voxels 8l..8l+7 (x + 8y + 64z order), 16 B or 20 B records.
template <bool Semantic>
__global__ __launch_bounds__(256) void k_download(Pool pool, const VisItem* sel, const uint32_t* n_sel,
                                                  float vs, float* out) {
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
  const uint32_t n = *n_sel;
  constexpr int R = Semantic ? 5 : 4;
  for (uint32_t b = wave; b < n; b += nwaves) {
    const VisItem it = sel[b];
    const int ty = lane & 7, tz = lane >> 3;
    const size_t v = ((size_t)it.idx << 9) + lane * 8;
    const int gy = (int16_t)((int16_t)(it.y << 3) + ty);
    const int gz = (int16_t)((int16_t)(it.z << 3) + tz);
    float* o = out + ((size_t)b * 512 + lane * 8) * R;
#pragma unroll
    for (int tx = 0; tx < 8; ++tx) {
      const int gx = (int16_t)((int16_t)(it.x << 3) + tx);
      o[tx * R + 0] = (float)gx * vs;
      o[tx * R + 1] = (float)gy * vs;
      o[tx * R + 2] = (float)gz * vs;
      o[tx * R + 3] = pool.tsdf[v + tx];
      if (Semantic) o[tx * R + 4] = pool.segm[v + tx];
    }
  }
}

// compact copy of the selected directory entries (12 B each) for export / dumps
__global__ void k_export_entries(const VisItem* sel, const uint32_t* n_sel, Entry* out_blocks,
                                 int32_t* out_entry_index, uint32_t cap, int32_t* out_count) {
  uint32_t n = *n_sel;
  if (n > cap) n = cap;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const VisItem it = sel[i];
    if (out_blocks) out_blocks[i] = Entry{it.x, it.y, it.z, it.offset, it.idx};
    if (out_entry_index) out_entry_index[i] = (int32_t)it.entry;
  }
  if (out_count && blockIdx.x == 0 && threadIdx.x == 0) *out_count = (int32_t)n;
}

// raw voxel storage of listed pool blocks (test hook)
__global__ void k_gather_voxels(Pool pool, const int32_t* pool_idx, int n, float* tsdf,
                                uint32_t* rgbw, float* prob) {
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
  for (uint32_t b = wave; b < (uint32_t)n; b += nwaves) {
    const size_t src = ((size_t)pool_idx[b] << 9) + lane * 8;
    const size_t dst = ((size_t)b << 9) + lane * 8;
    for (int i = 0; i < 8; ++i) {
      tsdf[dst + i] = pool.tsdf[src + i];
      rgbw[dst + i] = pool.rgbw[src + i];
      prob[dst + i] = pool.segm[src + i];
    }
  }
}

// Retrieve<Voxel>(point, cache) with a fresh cache, voxel_hash.cuh:104-143 (test hook)
__global__ void k_retrieve(Table tab, Pool pool, const int16_t* pts, int n, uint32_t* rgbw,
                           float* tsdf, float* prob, Entry* blocks) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
  const int bx = px >> 3, by = py >> 3, bz = pz >> 3;
  EntryWords w;
  const uint32_t e = find_block(tab, bx, by, bz, &w);
  const int vi = (px & 7) + (py & 7) * 8 + (pz & 7) * 64;
  if (e != kInf) {
    const size_t v = ((size_t)w.idx << 9) + vi;
    rgbw[i] = pool.rgbw[v];
    tsdf[i] = pool.tsdf[v];
    prob[i] = pool.segm[v];
  } else {
    rgbw[i] = 0;        // VoxelRGBW(): rgb 0, weight 0   (voxel_types.cu:3)
    tsdf[i] = -10.f;    // VoxelTSDF(): -10               (voxel_types.cu:8)
    prob[i] = 0.f;      // VoxelSEGM(): 0                 (voxel_types.cu:11)
  }
  blocks[i] = Entry{(int16_t)(w.w0 & 0xFFFFu), (int16_t)(w.w0 >> 16), (int16_t)(w.w1 & 0xFFFFu),
                    (int16_t)(w.w1 >> 16), w.idx};
}

// *RetrieveMutable<VoxelRGBW>(point) = value, voxel_hash_test.cu:47-54 (test hook)
__global__ void k_assign_rgbw(Table tab, Pool pool, const int16_t* pts, const uint32_t* vals,
                              int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
  EntryWords w;
  if (find_block(tab, px >> 3, py >> 3, pz >> 3, &w) == kInf) return;
  const int vi = (px & 7) + (py & 7) * 8 + (pz & 7) * 64;
  pool.rgbw[((size_t)w.idx << 9) + vi] = vals[i];
}

}  // namespace ratsdf
