// kernels_mesh.h -- marching-cubes mesh export for gfx950.
//
// Replaces marching_cube_kernel, compactify_kernel, transform_triangle_id_kernel and the prefix sums
// of TSDFGrid::GatherValidMesh (utils/tsdf/voxel_tsdf.cu:561-845).  One 512-thread workgroup per
// allocated block stages the 16^3 neighbourhood (this block and its +x/+y/+z neighbours, voxels with
// weight <= 10 read as "unobserved" = -10) of tsdf and probability in 32 KiB of LDS, emits the three
// candidate vertices of every point of the block's 9^3 vertex lattice and up to five triangles per
// voxel cube.  Compaction of vertices and triangles keeps index order (as the reference's inclusive
// scans do): a generic three-launch exclusive scan over the 0/1 masks.
#pragma once
#include "kernels_raycast.h"

namespace ratsdf {

// corner i of a cube, edge e = (corner, corner): conventions of the published table
__constant__ int8_t kCorner[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 0, 1}, {0, 0, 1},
                                     {0, 1, 0}, {1, 1, 0}, {1, 1, 1}, {0, 1, 1}};
__constant__ int8_t kEdge[12][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6},
                                    {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};

struct McTables {
  int8_t tri[256][16];      // edge list per sign pattern, -1 terminated
  int8_t edge_lower[12];    // corner of the edge with the smaller coordinate along the edge's axis
  int8_t edge_dim[12];      // axis of the edge (0 = x, 1 = y, 2 = z)
};

inline McTables make_mc_tables() {
  static const char* const cases[256] = {
#include "mc_cases.inc"
  };
  static const int corner[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 0, 1}, {0, 0, 1},
                                   {0, 1, 0}, {1, 1, 0}, {1, 1, 1}, {0, 1, 1}};
  static const int edge[12][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6},
                                  {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};
  McTables t;
  for (int c = 0; c < 256; ++c) {
    int n = 0;
    for (const char* p = cases[c]; *p; ++p) t.tri[c][n++] = (int8_t)(*p <= '9' ? *p - '0' : *p - 'a' + 10);
    for (; n < 16; ++n) t.tri[c][n] = -1;
  }
  for (int e = 0; e < 12; ++e) {
    const int* a = corner[edge[e][0]];
    const int* b = corner[edge[e][1]];
    int dim = 0;
    for (int d = 0; d < 3; ++d)
      if (a[d] != b[d]) dim = d;
    t.edge_dim[e] = (int8_t)dim;
    t.edge_lower[e] = (int8_t)(a[dim] < b[dim] ? edge[e][0] : edge[e][1]);
  }
  return t;
}

constexpr int kVertVolume = 729;  // 9^3 lattice points per block

__global__ __launch_bounds__(512) void k_marching_cubes(Table tab, Pool pool, const VisItem* blocks,
                                                        const McTables* mc, float vs, float* verts,
                                                        float* vprob, uint32_t* vmask,
                                                        int32_t* tids, uint32_t* tmask) {
  __shared__ float ct[16][16][16];
  __shared__ float cp[16][16][16];
  __shared__ int32_t nb_idx[8];
  const uint32_t bi = blockIdx.x;
  const VisItem base = blocks[bi];
  const int tid = threadIdx.x;
  const int tx = tid & 7, ty = (tid >> 3) & 7, tz = tid >> 6;
  if (tid < 8) {  // GetBlock of the 2x2x2 neighbourhood, voxel_tsdf.cu:582-586
    EntryWords w;
    const uint32_t e = find_block(tab, (int16_t)(base.x + (tid & 1)), (int16_t)(base.y + ((tid >> 1) & 1)),
                                  (int16_t)(base.z + (tid >> 2)), &w);
    nb_idx[tid] = e == kInf ? -1 : w.idx;
  }
  __syncthreads();
#pragma unroll
  for (int n = 0; n < 8; ++n) {  // n = x + 2y + 4z of the neighbour
    const int32_t idx = nb_idx[n];
    float t = -10.f, pr = 0.f;                                             // :612-620
    if (idx >= 0) {
      const size_t vi = ((size_t)idx << 9) + tid;
      if ((int)(pool.rgbw[vi] >> 24) > 10) {                               // :600
        t = pool.tsdf[vi];
        pr = pool.segm[vi];
      }
    }
    ct[(n >> 2) * 8 + tz][((n >> 1) & 1) * 8 + ty][(n & 1) * 8 + tx] = t;
    cp[(n >> 2) * 8 + tz][((n >> 1) & 1) * 8 + ty][(n & 1) * 8 + tx] = pr;
  }
  __syncthreads();
  float lt[8];
  int cubeindex = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {                                            // :627-636
    lt[i] = ct[tz + kCorner[i][2]][ty + kCorner[i][1]][tx + kCorner[i][0]];
    cubeindex |= (lt[i] < 0) << i;
  }
  // three vertices per lattice point, :639-671
  for (int c = tid; c < kVertVolume; c += 512) {
    const int vx = c % 9, vy = c / 9 % 9, vz = c / 81;
    const float v1[3] = {(float)(int16_t)((int16_t)(base.x << 3) + vx),
                         (float)(int16_t)((int16_t)(base.y << 3) + vy),
                         (float)(int16_t)((int16_t)(base.z << 3) + vz)};
    const float t1 = ct[vz][vy][vx], p1 = cp[vz][vy][vx];
    const size_t cube_idx = (size_t)bi * kVertVolume + c;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int ox = j == 0, oy = j == 1, oz = j == 2;
      const float t2 = ct[vz + oz][vy + oy][vx + ox];
      const float p2 = cp[vz + oz][vy + oy][vx + ox];
      const float sfac = (-t1) / (t2 - t1);
      float* o = verts + (cube_idx * 3 + j) * 3;
      o[0] = (v1[0] + sfac * (float)ox) * vs;
      o[1] = (v1[1] + sfac * (float)oy) * vs;
      o[2] = (v1[2] + sfac * (float)oz) * vs;
      vprob[cube_idx * 3 + j] = (p1 + p2) / 2;
      vmask[cube_idx * 3 + j] = 0;
    }
  }
  __syncthreads();
  const size_t thread_idx = (size_t)bi * 512 + tid;
  const int8_t* cs = mc->tri[cubeindex];
#pragma unroll 1
  for (int i = 0; i < 5; ++i) {                                            // :678-714
    const size_t tri = thread_idx * 5 + i;
    uint32_t m = 0;
    if (cs[i * 3] != -1) {
      m = 1;
      for (int j = 0; j < 3; ++j) {
        const int e = cs[i * 3 + j];
        const float diff = fabsf(lt[kEdge[e][1]] - lt[kEdge[e][0]]);
        if ((double)diff < 1e-3 || diff >= 2) {                            // :692-696
          m = 0;
          break;
        }
        const int lo = mc->edge_lower[e], dim = mc->edge_dim[e];
        const int coi = (kCorner[lo][2] + tz) * 81 + (kCorner[lo][1] + ty) * 9 + (kCorner[lo][0] + tx);
        const size_t cube_idx = (size_t)bi * kVertVolume + coi;
        tids[tri * 3 + j] = (int32_t)(cube_idx * 3 + dim);
        vmask[cube_idx * 3 + dim] = 1;
      }
    }
    tmask[tri] = m;
  }
}

// ---- exclusive scan of a 0/1 mask of n words: per-tile sums -> scan of tile sums -> positions ----
constexpr int kScanTile = 4096;  // items per 1024-thread workgroup (4 per thread)

__global__ __launch_bounds__(1024) void k_mask_tile_sums(const uint32_t* mask, size_t n,
                                                         uint32_t* tile_sum) {
  __shared__ uint32_t lds[32];
  const size_t base = (size_t)blockIdx.x * kScanTile + threadIdx.x * 4;
  uint32_t s = 0;
  for (int k = 0; k < 4; ++k)
    if (base + k < n) s += mask[base + k];
  uint32_t total = 0;
  (void)block_exclusive_scan(s, lds, &total);
  if (threadIdx.x == 0) tile_sum[blockIdx.x] = total;
}

// single workgroup: exclusive scan of the tile sums in place; grand total -> *total_out
__global__ __launch_bounds__(1024) void k_scan_tile_sums(uint32_t* tile_sum, uint32_t ntiles,
                                                         uint32_t* total_out) {
  __shared__ uint32_t lds[32];
  uint32_t carry = 0;
  for (uint32_t base = 0; base < ntiles; base += 1024) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < ntiles ? tile_sum[i] : 0;
    uint32_t total = 0;
    const uint32_t ex = block_exclusive_scan(v, lds, &total);
    if (i < ntiles) tile_sum[i] = carry + ex;
    carry += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total_out = carry;
}

// position of every item among the set ones (exclusive), written to `pos` (only meaningful where set)
__global__ __launch_bounds__(1024) void k_mask_positions(const uint32_t* mask, size_t n,
                                                         const uint32_t* tile_off, uint32_t* pos) {
  __shared__ uint32_t lds[32];
  const size_t base = (size_t)blockIdx.x * kScanTile + threadIdx.x * 4;
  uint32_t m[4], s = 0;
  for (int k = 0; k < 4; ++k) {
    m[k] = base + k < n ? mask[base + k] : 0;
    s += m[k];
  }
  uint32_t total = 0;
  uint32_t run = tile_off[blockIdx.x] + block_exclusive_scan(s, lds, &total);
  for (int k = 0; k < 4; ++k) {
    if (base + k < n) pos[base + k] = run;
    run += m[k];
  }
}

__global__ void k_compact_vertices(const float* verts, const float* vprob, const uint32_t* vmask,
                                   const uint32_t* vpos, size_t n, float* out_v, float* out_p) {
  for (size_t i = (size_t)blockIdx.x * block_threads() + threadIdx.x; i < n;
       i += (size_t)gridDim.x * block_threads()) {
    if (!vmask[i]) continue;
    const uint32_t o = vpos[i];
    out_v[(size_t)o * 3 + 0] = verts[i * 3 + 0];
    out_v[(size_t)o * 3 + 1] = verts[i * 3 + 1];
    out_v[(size_t)o * 3 + 2] = verts[i * 3 + 2];
    out_p[o] = vprob[i];
  }
}

__global__ void k_compact_triangles(const int32_t* tids, const uint32_t* tmask, const uint32_t* tpos,
                                    const uint32_t* vpos, size_t n, int32_t* out_i) {
  for (size_t i = (size_t)blockIdx.x * block_threads() + threadIdx.x; i < n;
       i += (size_t)gridDim.x * block_threads()) {
    if (!tmask[i]) continue;
    const uint32_t o = tpos[i];
    for (int j = 0; j < 3; ++j) out_i[(size_t)o * 3 + j] = (int32_t)vpos[(size_t)tids[i * 3 + j]];
  }
}

}  // namespace ratsdf
