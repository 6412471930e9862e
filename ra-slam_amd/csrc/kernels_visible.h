// kernels_visible.h -- ordered compaction of the hash directory for gfx950.
//
// Replaces check_visibility_kernel / check_valid_kernel / check_bound_kernel + prefix_sum +
// gather_visible_blocks_kernel + the blocking read-back of the count
// (utils/tsdf/voxel_tsdf.cu:15-33,98-118,465-474,847-867; utils/cuda/arithmetic.cuh:52-172).
// The reference scans all 2^22 directory entries (48 MiB) three times per frame.  Here the engine
// keeps a 512 KiB occupancy bitmap of the directory (one bit per entry, maintained by the commit and
// carve code); a pass reads the bitmap and only the entries whose bit is set.
//   * per frame: visible_append_role builds the (unordered) visible work list;
//   * queries / exports: k_select_flags + k_select_scatter build a list ordered by ascending
//     hash-entry index exactly like the reference's scan + gather (one lane per 64-entry word
//     produces a selection mask, the scatter kernel turns per-workgroup counts into offsets).
// Counts stay on the device (FrameCtl::n_list / Ctl::n_sel); consumers are persistent grids that read them.
#pragma once
#include "kernels_cand.h"

namespace ratsdf {

enum SelectMode { kSelVisible = 0, kSelValid = 1, kSelBounds = 2, kSelOwned = 3 };

struct GridBounds {  // BoundingCube<short>, voxel_tsdf.cuh:19-34
  int16_t xmin, xmax, ymin, ymax, zmin, zmax;
};

constexpr int kVisWG = 256;  // occupancy words (64 entries each) per workgroup

// One lane per 64-entry occupancy word: evaluates the selection predicate for every allocated entry
// of the word and writes the resulting 64-bit mask; the workgroup's total goes to wg_count.
// Must be called by all threads of a kVisWG-thread workgroup (it synchronises).
template <int Mode>
__device__ inline void select_flags_role(const Table& tab, const FrameParams& P,
                                         const GridBounds& gb, uint32_t wg,
                                         unsigned long long* masks, uint32_t* wg_count) {
  __shared__ uint32_t wave_cnt[kVisWG / 64];
  const uint32_t nwords = tab.num_entry >> 6;
  const uint32_t w = wg * kVisWG + threadIdx.x;
  unsigned long long occ = w < nwords ? tab.occ[w] : 0ull;
  unsigned long long sel = 0;
  if (Mode == kSelValid) {
    sel = occ;                                                              // voxel_tsdf.cu:28-33
  } else {
    while (occ) {
      const int b = __ffsll((long long)occ) - 1;
      occ &= occ - 1;
      const uint32_t* p = reinterpret_cast<const uint32_t*>(tab.entries + ((size_t)w * 64 + b));
      const uint32_t w0 = p[0], w1 = p[1];
      const int bx = (int16_t)(w0 & 0xFFFFu), by = (int16_t)(w0 >> 16), bz = (int16_t)(w1 & 0xFFFFu);
      bool s;
      if (Mode == kSelVisible) {
        s = block_visible<false>(bx, by, bz, P);                            // voxel_tsdf.cu:98-109
      } else if (Mode == kSelOwned) {
        s = shard_owned(bx, P);  // (blocks imported from a neighbour's subvolume are read, not meshed)
      } else {
        const int gx = (int16_t)(bx << 3), gy = (int16_t)(by << 3), gz = (int16_t)(bz << 3);
        s = gx >= gb.xmin && gy >= gb.ymin && gz >= gb.zmin && gx + 8 - 1 <= gb.xmax &&
            gy + 8 - 1 <= gb.ymax && gz + 8 - 1 <= gb.zmax;                 // voxel_tsdf.cu:15-26
      }
      if (s) sel |= 1ull << b;
    }
  }
  if (w < nwords) masks[w] = sel;
  uint32_t cnt = __popcll(sel);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
  if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0;
#pragma unroll
    for (int i = 0; i < kVisWG / 64; ++i) t += wave_cnt[i];
    wg_count[wg] = t;
  }
}

// Visibility of the blocks that exist before this frame (check_visibility_kernel +
// gather_visible_blocks_kernel, voxel_tsdf.cu:98-118): every live block is tested (any of the 8 corners in view)
// and the workgroup appends its visible blocks to the frame's 8 work lists (one per XCD, see block_list_of) with
// one atomicAdd per non-empty list.  The lists are unordered; nothing in a frame depends on their order (blocks are
// independent; the carve pass orders its pool releases by hash entry).
//
// Where the live blocks come from (round 5): Table::active, one 16-byte item per POOL slot, kept by the code that
// writes an entry's pool index.  Pool indices are handed out from the top (voxel_mem.cu:24,38-41), so the slots
// that have ever been in use are [Ctl::free_low, num_block): lane m of the role takes slot num_block - 1 - m (then
// every `lanes`-th slot further down).  Its first load needs nothing but the lane's own index -- it is issued
// before anything else, beside the low-water mark it will be checked against -- and is a coalesced 16-byte read:
// ONE memory round trip before the visibility test.  Until round 5 the role read the 512 KiB occupancy bitmap
// (256 workgroups), compacted the set bits through LDS and gathered the 12-byte entries out of the 48 MiB table:
// two dependent round trips, two more barriers, and 256 x 8 returning atomics on the eight list counters where
// ~20 x 8 are made now (a 640x480 / 5 mm map has ~5 k live blocks: the workgroups below them find empty slots
// and leave).
// `gate()` makes sure the previous frame's queued deletes have happened (carve_resolve_gate: they edit `active`
// with write-through stores); the first load is issued before it so that the two round trips overlap, and is
// repeated past the caches when the gate had to wait.
constexpr int kVisWordsPerLane = 1;
constexpr uint32_t kVisListCap = 2048;  // (LDS words the role may use: the front kernels size their buffer by it)
constexpr int kVisSlotsPerLane = 2;     // pool slots a lane tests per round

__device__ inline uint4 ld_agent_item(const VisItem* p) {
  const uint32_t* w = reinterpret_cast<const uint32_t*>(p);
  return make_uint4(__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                    __hip_atomic_load(w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                    __hip_atomic_load(w + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                    __hip_atomic_load(w + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

template <typename Gate>
__device__ inline void visible_append_role(const Table& tab, const FrameParams& P, uint32_t wg, uint32_t n_wg,
                                           VisItem* vis, uint32_t seg_cap, Ctl* ctl, FrameCtl* F,
                                           Gate gate, uint32_t* lds /* role LDS: unused words */) {
  __shared__ uint32_t cnt[kNumLists], base[kNumLists], cnt2[kNumLists];
  (void)lds;
  const uint32_t tid = threadIdx.x, nt = block_threads();
  const uint32_t lanes = n_wg * nt;
  const uint32_t me = wg * nt + tid;
  const uint32_t nb = (uint32_t)tab.num_block;
  const uint4* act = reinterpret_cast<const uint4*>(tab.active);
  // (an item is "empty" when its pool index is negative; loads are clamped into the array instead of selected
  // against a constant item -- a select between two 16-byte objects made the compiler park one in scratch memory)
  uint4 it[kVisSlotsPerLane];
  it[0] = act[me < nb ? nb - 1u - me : 0u];      // (speculative: whatever the low-water mark says)
  if (me >= nb) it[0].z = ~0u;
  int32_t low = ctl->free_low;                    // written by earlier launches only
  const uint32_t g = gate();
  if (g == kGateExpired) return;  // uniform: the directory may be half-edited (sticky error set)
  const bool stale = g != kGateOpen;  // rare: `active` was edited after the load above was issued
  if (low < 0) low = 0;
  const uint32_t used = nb - (uint32_t)low;       // slots [low, nb) have been in use at some time
  // a workgroup whose first slot lies below the mark has nothing in any round (slots only go down from there)
  if (wg * nt >= used) return;  // uniform
  for (uint32_t first = 0; first < used; first += lanes * kVisSlotsPerLane) {  // uniform trip count
    if (tid < kNumLists) {
      cnt[tid] = 0;
      cnt2[tid] = 0;
    }
    __syncthreads();
    uint32_t tag[kVisSlotsPerLane];
#pragma unroll
    for (int j = 0; j < kVisSlotsPerLane; ++j) {
      const uint32_t o = first + (uint32_t)j * lanes + me;   // distance from the top of the pool
      if (first != 0 || j != 0) {
        it[j] = act[o < used ? nb - 1u - o : 0u];
        if (o >= used) it[j].z = ~0u;
      }
    }
#pragma unroll
    for (int j = 0; j < kVisSlotsPerLane; ++j) {
      const uint32_t o = first + (uint32_t)j * lanes + me;
      if (stale && o < used) it[j] = ld_agent_item(tab.active + (nb - 1u - o));
      tag[j] = ~0u;
      if (o < used && (int32_t)it[j].z >= 0) {  // a live block
        const int bx = (int16_t)(it[j].x & 0xFFFFu), by = (int16_t)(it[j].x >> 16), bz = (int16_t)(it[j].y & 0xFFFFu);
        if (block_visible<false>(bx, by, bz, P)) {
          const int l = block_list_of(bx, by, bz, P);
          atomicAdd(&cnt[l], 1u);
          tag[j] = (uint32_t)l;
        }
      }
    }
    __syncthreads();
    if (tid < kNumLists) {
      const uint32_t c = cnt[tid];
      base[tid] = c ? atomicAdd(&F->n_list[tid * kListStride], c) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kVisSlotsPerLane; ++j) {
      if (tag[j] == ~0u) continue;
      const uint32_t l = tag[j];
      const uint32_t pos = base[l] + atomicAdd(&cnt2[l], 1u);
      if (pos < seg_cap - kFreshCap) {  // (the next segment's new-block items start there)
        uint4 v = it[j];
        v.y &= 0xFFFFu;  // {x | y << 16, z, pool index, hash entry}
        reinterpret_cast<uint4*>(vis)[(size_t)l * seg_cap + pos] = v;
      }
    }
    __syncthreads();  // (the counters are reset at the top of the next round)
  }
}

template <int Mode>
__global__ __launch_bounds__(kVisWG) void k_select_flags(Table tab, FrameParams P, GridBounds gb,
                                                         unsigned long long* masks,
                                                         uint32_t* wg_count) {
  select_flags_role<Mode>(tab, P, gb, blockIdx.x, masks, wg_count);
}

// Scatter the selected entries, in ascending entry order, as 16-byte items {entry copy, entry
// index}.  Each workgroup derives its output offset from the (<= a few hundred) workgroup counts,
// so no separate scan launch is needed; the last workgroup publishes the total.
__global__ __launch_bounds__(kVisWG) void k_select_scatter(Table tab,
                                                           const unsigned long long* masks,
                                                           const uint32_t* wg_count, VisItem* out,
                                                           uint32_t out_cap, uint32_t* total_out) {
  __shared__ uint32_t lds[32];
  const uint32_t wg = blockIdx.x, tid = threadIdx.x;
  uint32_t before = 0;
  for (uint32_t j = tid; j < wg; j += kVisWG) before += wg_count[j];
  uint32_t wg_base = 0;
  (void)block_exclusive_scan(before, lds, &wg_base);
  const uint32_t own = wg_count[wg];
  if (wg == gridDim.x - 1 && tid == 0) *total_out = wg_base + own;
  if (own == 0) return;  // uniform
  const uint32_t nwords = tab.num_entry >> 6;
  const uint32_t w = wg * kVisWG + tid;
  unsigned long long m = w < nwords ? masks[w] : 0ull;
  uint32_t dummy = 0;
  uint32_t pos = wg_base + block_exclusive_scan((uint32_t)__popcll(m), lds, &dummy);
  while (m) {
    const int b = __ffsll((long long)m) - 1;
    m &= m - 1;
    const uint32_t e = w * 64 + b;
    if (pos < out_cap) {
      const EntryWords ew = load_entry(tab.entries, e);
      uint4 v;
      v.x = ew.w0;
      v.y = ew.w1;
      v.z = (uint32_t)ew.idx;
      v.w = e;
      reinterpret_cast<uint4*>(out)[pos] = v;
    }
    ++pos;
  }
}

// download_tsdf_kernel / download_semantic_kernel, voxel_tsdf.cu:35-62: one wave per selected
// block, lane l writes voxels 8l..8l+7 (x + 8y + 64z order), 16 B or 20 B records.
template <bool Semantic>
__global__ __launch_bounds__(256) void k_download(Pool pool, const VisItem* sel, const uint32_t* n_sel,
                                                  float vs, float* out) {
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * block_threads() + threadIdx.x) >> 6;
  const uint32_t nwaves = (gridDim.x * block_threads()) >> 6;
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

// compact copy of the selected directory entries (12 B each) for export / dumps.  *out_count gets
// the number of selected entries even when it exceeds `cap` (then only `cap` are written and the
// sticky error says so: a truncated directory must not pass for a complete one).
__global__ void k_export_entries(const VisItem* sel, const uint32_t* n_sel, Entry* out_blocks,
                                 int32_t* out_entry_index, uint32_t cap, int32_t* out_count,
                                 Ctl* ctl) {
  const uint32_t total = *n_sel;
  const uint32_t n = total > cap ? cap : total;
  for (uint32_t i = blockIdx.x * block_threads() + threadIdx.x; i < n; i += gridDim.x * block_threads()) {
    const VisItem it = sel[i];
    if (out_blocks) out_blocks[i] = Entry{it.x, it.y, it.z, it.offset, it.idx};
    if (out_entry_index) out_entry_index[i] = (int32_t)it.entry;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (out_count) *out_count = (int32_t)total;
    if (total > cap && ctl) set_error(ctl, RATSDF_ERR_CAPACITY);
  }
}

// ratsdf_import_blocks[_device]: the voxels of block i of the list into the pool block its directory entry names; one
// wave per block; blocks the directory does not hold are counted in *missing.  `stride`: 32-bit words between
// consecutive blocks of one array (512: three arrays of n x 512 words, the host entry point's layout; 1536: one
// record {tsdf | rgbw | prob} per block, the device entry points' layout -- tsdf / rgbw / prob then point 0 / 512 /
// 1024 words into the first record).
__global__ __launch_bounds__(256) void k_import_voxels(Table tab, Pool pool, const int16_t* pos, int n,
                                                       const float* tsdf, const uint32_t* rgbw, const float* prob,
                                                       uint32_t stride, uint32_t* missing) {
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * block_threads() + threadIdx.x) >> 6;
  const uint32_t nwaves = (gridDim.x * block_threads()) >> 6;
  for (uint32_t b = wave; b < (uint32_t)n; b += nwaves) {
    EntryWords w;
    const uint32_t e = find_block(tab, pos[3 * b], pos[3 * b + 1], pos[3 * b + 2], &w);
    if (e == kInf || w.idx < 0 || w.idx == kPlaceholderIdx) {
      if (lane == 0) atomicAdd(missing, 1u);
      continue;
    }
    const size_t dst = ((size_t)w.idx << 9) + lane * 8, src = (size_t)b * stride + lane * 8;
    for (int i = 0; i < 8; ++i) {
      pool.tsdf[dst + i] = tsdf[src + i];
      pool.rgbw[dst + i] = rgbw[src + i];
      pool.segm[dst + i] = prob[src + i];
    }
  }
}

// ratsdf_export_blocks_device: the reverse -- record b of `out` ({tsdf | rgbw | prob}, 1536 words) takes the voxels of
// the block at position b of the list; a block the directory does not hold leaves a record of zeros and is counted.
__global__ __launch_bounds__(256) void k_export_blocks(Table tab, Pool pool, const int16_t* pos, int n, uint32_t* out,
                                                       uint32_t* missing) {
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * block_threads() + threadIdx.x) >> 6;
  const uint32_t nwaves = (gridDim.x * block_threads()) >> 6;
  for (uint32_t b = wave; b < (uint32_t)n; b += nwaves) {
    EntryWords w;
    const uint32_t e = find_block(tab, pos[3 * b], pos[3 * b + 1], pos[3 * b + 2], &w);
    const bool have = !(e == kInf || w.idx < 0 || w.idx == kPlaceholderIdx);
    if (!have && lane == 0) atomicAdd(missing, 1u);
    const size_t src = ((size_t)(have ? w.idx : 0) << 9) + lane * 8, dst = (size_t)b * 1536u + lane * 8;
    for (int i = 0; i < 8; ++i) {
      out[dst + i] = have ? __float_as_uint(pool.tsdf[src + i]) : 0u;
      out[dst + 512 + i] = have ? pool.rgbw[src + i] : 0u;
      out[dst + 1024 + i] = have ? __float_as_uint(pool.segm[src + i]) : 0u;
    }
  }
}

// ---- directory delta ("RCCL all-gather of the block directory delta") ---------------------------
// Entries written since the last export, as they are now: one lane per 64-entry word of the dirty bitmap (the
// words behind the occupancy bitmap); the workgroup's live entries go to out[0 ..) behind one returning atomic
// per workgroup; the word is cleared.  counts[0] = entries listed (the TRUE number even past `cap`: the caller
// sees that the payload was too small).
__global__ __launch_bounds__(kVisWG) void k_delta_added(Table tab, Entry* out, uint32_t cap, uint32_t* counts) {
  __shared__ uint32_t wg_n, wg_base;
  const uint32_t nwords = tab.num_entry >> 6;
  const uint32_t w = blockIdx.x * kVisWG + threadIdx.x;
  if (threadIdx.x == 0) wg_n = 0;
  __syncthreads();
  unsigned long long* dirty = tab.occ + nwords;
  const unsigned long long m = w < nwords ? dirty[w] : 0ull;
  if (m) dirty[w] = 0ull;
  unsigned long long live = 0;  // dirty AND still holding a block
  for (unsigned long long t = m; t; t &= t - 1) {
    const int b = __ffsll((long long)t) - 1;
    if (load_entry(tab.entries, w * 64 + (uint32_t)b).idx >= 0) live |= 1ull << b;
  }
  uint32_t pos = live ? atomicAdd(&wg_n, (uint32_t)__popcll(live)) : 0u;
  __syncthreads();
  if (threadIdx.x == 0) wg_base = wg_n ? atomicAdd(&counts[0], wg_n) : 0u;
  __syncthreads();
  pos += wg_base;
  for (; live; live &= live - 1, ++pos) {
    const EntryWords ew = load_entry(tab.entries, w * 64 + (uint32_t)(__ffsll((long long)live) - 1));
    if (pos < cap) {
      uint32_t* o = reinterpret_cast<uint32_t*>(out + pos);
      o[0] = ew.w0;
      o[1] = ew.w1;
      o[2] = (uint32_t)ew.idx;
    }
  }
}
// Positions deleted since the last export, behind the added entries: out[counts[0] ..); counts[1] = their number
// (0x7FFFFFFF when the log overflowed: the delta is unusable, the caller takes a whole directory).  Resets the log.
__global__ __launch_bounds__(256) void k_delta_deleted(Table tab, Entry* out, uint32_t cap, uint32_t* counts) {
  const uint32_t logged = *tab.del_count;
  const uint32_t n = logged < tab.del_cap ? logged : tab.del_cap;
  const uint32_t base = counts[0];
  for (uint32_t i = blockIdx.x * block_threads() + threadIdx.x; i < n; i += gridDim.x * block_threads()) {
    const uint2 d = tab.del_log[i];
    if (base + i < cap) {
      uint32_t* o = reinterpret_cast<uint32_t*>(out + base + i);
      o[0] = d.x;
      o[1] = d.y;  // offset 0
      o[2] = (uint32_t)-1;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) counts[1] = logged > tab.del_cap ? 0x7FFFFFFFu : n;
}
__global__ void k_delta_reset(Table tab) { *tab.del_count = 0; }

// raw voxel storage of listed pool blocks (test hook)
__global__ void k_gather_voxels(Pool pool, const int32_t* pool_idx, int n, float* tsdf,
                                uint32_t* rgbw, float* prob) {
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * block_threads() + threadIdx.x) >> 6;
  const uint32_t nwaves = (gridDim.x * block_threads()) >> 6;
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
  const int i = blockIdx.x * block_threads() + threadIdx.x;
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
  const int i = blockIdx.x * block_threads() + threadIdx.x;
  if (i >= n) return;
  const int px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
  EntryWords w;
  if (find_block(tab, px >> 3, py >> 3, pz >> 3, &w) == kInf) return;
  const int vi = (px & 7) + (py & 7) * 8 + (pz & 7) * 64;
  pool.rgbw[((size_t)w.idx << 9) + vi] = vals[i];
}

}  // namespace ratsdf
