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
// gather_visible_blocks_kernel, voxel_tsdf.cu:98-118): one lane per 64-entry occupancy word tests
// the allocated entries of its word (any of the 8 corners in view) and the workgroup appends its
// visible blocks to the frame's 8 work lists (one per XCD, see block_list_of) with one atomicAdd per
// non-empty list.  The lists are unordered; nothing in a frame depends on their order (blocks are
// independent; the carve pass orders its pool releases by hash entry).
// `gate()` makes sure the previous frame's queued deletes have happened (carve_resolve_gate); the
// first load is issued before it so that the two round trips overlap.
// The set bits of the workgroup's occupancy words are first compacted into an LDS list and then
// handed out one per lane: a lane that walks the bits of its own word serially pays one dependent
// memory round trip per block, and the launch waits for the unluckiest lane of 65 536.
constexpr int kVisWordsPerLane = 1;
constexpr uint32_t kVisListCap = 2048;  // entries per round (more occupied entries: more rounds)

template <typename Gate>
__device__ inline void visible_append_role(const Table& tab, const FrameParams& P, uint32_t wg,
                                           VisItem* vis, uint32_t seg_cap, Ctl* ctl, FrameCtl* F,
                                           Gate gate, uint32_t* list /* LDS, kVisListCap words */) {
  __shared__ uint32_t n_items, more, cnt[kNumLists], base[kNumLists], cnt2[kNumLists];
  const uint32_t nwords = tab.num_entry >> 6;
  const uint32_t w = wg * kVisWG + threadIdx.x;
  const uint32_t tid = threadIdx.x, nt = block_threads();
  unsigned long long occ = w < nwords ? tab.occ[w] : 0ull;
  const uint32_t g = gate();
  if (g == kGateExpired) return;  // uniform: the directory may be half-edited (sticky error set)
  if (g != kGateOpen)  // rare: the directory changed after the load above was issued
    occ = w < nwords ? __hip_atomic_load(&tab.occ[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
  for (;;) {  // one round unless the workgroup's 16 384 entries hold more than kVisListCap blocks
    if (tid < kNumLists) {
      cnt[tid] = 0;
      cnt2[tid] = 0;
    }
    if (tid == 0) {
      n_items = 0;
      more = 0;
    }
    __syncthreads();
    if (occ) {
      uint32_t at = atomicAdd(&n_items, (uint32_t)__popcll(occ));
      while (occ && at < kVisListCap) {
        const int b = __ffsll((long long)occ) - 1;
        occ &= occ - 1;
        list[at++] = w * 64 + (uint32_t)b;
      }
      if (occ) more = 1;
    }
    __syncthreads();
    const uint32_t n = n_items < kVisListCap ? n_items : kVisListCap;
    const bool again = more != 0;
    // one listed entry per lane: load, test (any corner in view, voxel_tsdf.cu:98-109), pick a list
    EntryWords first{0, 0, -1};
    for (uint32_t i = tid; i < n; i += nt) {
      const uint32_t e = list[i];
      const EntryWords ew = load_entry(tab.entries, e);
      if (i == tid) first = ew;
      const int bx = (int16_t)(ew.w0 & 0xFFFFu), by = (int16_t)(ew.w0 >> 16),
                bz = (int16_t)(ew.w1 & 0xFFFFu);
      uint32_t tag = 0;
      if (block_visible<false>(bx, by, bz, P)) {
        const int l = block_list_of(bx, by, bz, P);
        atomicAdd(&cnt[l], 1u);
        tag = 0x10000000u | ((uint32_t)l << 29);
      }
      list[i] = e | tag;  // entry indices use 27 bits at most (bucket_bits <= 26)
    }
    __syncthreads();
    if (tid < kNumLists) {
      const uint32_t c = cnt[tid];
      base[tid] = c ? atomicAdd(&F->n_list[tid * kListStride], c) : 0u;
    }
    __syncthreads();
    for (uint32_t i = tid; i < n; i += nt) {
      const uint32_t t = list[i];
      if (!(t & 0x10000000u)) continue;
      const uint32_t e = t & 0x0FFFFFFFu, l = t >> 29;
      const uint32_t pos = base[l] + atomicAdd(&cnt2[l], 1u);
      if (pos < seg_cap - kFreshCap) {  // (the next segment's new-block items start there)
        const EntryWords ew = i == tid ? first : load_entry(tab.entries, e);
        uint4 v;
        v.x = ew.w0;
        v.y = ew.w1;
        v.z = (uint32_t)ew.idx;
        v.w = e;
        reinterpret_cast<uint4*>(vis)[(size_t)l * seg_cap + pos] = v;
      }
    }
    if (!again) break;  // uniform
    __syncthreads();
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

// ratsdf_import_blocks: the voxels of block i of the list into the pool block its directory entry names; one wave
// per block; blocks the directory does not hold are counted in *missing.
__global__ __launch_bounds__(256) void k_import_voxels(Table tab, Pool pool, const int16_t* pos, int n,
                                                       const float* tsdf, const uint32_t* rgbw, const float* prob,
                                                       uint32_t* missing) {
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
    const size_t dst = ((size_t)w.idx << 9) + lane * 8, src = ((size_t)b << 9) + lane * 8;
    for (int i = 0; i < 8; ++i) {
      pool.tsdf[dst + i] = tsdf[src + i];
      pool.rgbw[dst + i] = rgbw[src + i];
      pool.segm[dst + i] = prob[src + i];
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
