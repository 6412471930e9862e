// ratsdf_oracle.cpp -- CPU restatement of RA-SLAM's voxel-hashed TSDF + semantic integration path.
//
// *** TEST INFRASTRUCTURE ONLY. ***  Nothing under oracle/ may be imported, linked or executed by
// the product path (ra-slam_amd/, the C-ABI library libratsdf.so).  Only tests/, the smoke check
// in __graft_entry__.py and bench.py's cpu_baseline leg use it, and only as the checker / the
// reported CPU baseline.
//
// What it restates (paths relative to the reference tree; the reference itself is CUDA + Eigen +
// OpenCV and cannot be built in this image):
//   utils/tsdf/voxel_mem.cuh:11-95, voxel_mem.cu:13-61     pool, free list, block init
//   utils/tsdf/voxel_hash.cuh:13-25,104-143                 table geometry, Retrieve / cache
//   utils/tsdf/voxel_hash.cu:19-23,46-159,190-218           Hash, Allocate, Delete, GetBlock
//   utils/tsdf/voxel_tsdf.cu:15-276                         all integration-side kernels
//   utils/tsdf/voxel_tsdf.cu:416-474,532-559,847-883        host sequencing of a frame / queries
//   utils/cuda/camera.cuh:35-67, utils/cuda/lie_group.cuh:15-40
//   utils/tsdf/voxel_types.cu:3-12                          default voxels
//   modules/tsdf_module.cc:22-37                            ones-fill for missing ht/lt
// Third-party arithmetic that is NOT in the reference tree and is restated from its published
// algorithm: Eigen 3.3.7 (reference CMakeLists.txt:22) -- quaternion*vector
// (QuaternionBase::_transformVector), quaternion inverse (conjugate / squaredNorm), 3-vector norm
// and 4-vector squaredNorm in the order of Eigen's non-vectorised redux unroller (halving split),
// hnormalized as per-component division, vector/scalar as per-component division.
//
// Parity pinning: the hash function, bucket/lock semantics and pool behaviour are pinned by the
// reference's own gtest known-answers (utils/tests/voxel_hash_test.cu, voxel_mem_test.cu; see
// tests/test_oracle_kat.py).  The reference has NO test, fixture or golden value for
// TSDFGrid::Integrate, so voxel-value parity (pose math, rounding of transcendental functions) is
// "parity unpinned" beyond line-by-line fidelity to the formulas cited above.
//
// Canonical linearisation (the CUDA reference is run-to-run non-deterministic in which thread wins
// a bucket lock and in atomicSub order): the allocation pass runs pixels in raster order (y outer,
// x inner), ray samples ascending, sequentially; the carve pass runs the visible list in ascending
// hash-entry order, sequentially.  Bucket locks are only released by ResetLocks after each pass.
// Defined values for reference UB: hash entries start {pos 0, offset 0, idx -1}; voxel memory
// starts zeroed; float->int of NaN is 0 and saturates otherwise (CUDA cvt semantics).
//
// Build: see oracle/Makefile (g++ -O2 -ffp-contract=off, no fast-math).

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../include/ratsdf.h"

namespace {

struct V3 {
  float x, y, z;
};
struct Quat {
  float x, y, z, w;
};
struct S3 {
  int16_t x, y, z;
};

inline bool operator==(const S3& a, const S3& b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

// Eigen cross product of 3-vectors.
inline V3 cross(const V3& a, const V3& b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// Eigen QuaternionBase::_transformVector: uv = q.vec x v; uv += uv; v + w*uv + q.vec x uv.
inline V3 qrot(const Quat& q, const V3& v) {
  const V3 qv{q.x, q.y, q.z};
  V3 uv = cross(qv, v);
  uv.x += uv.x;
  uv.y += uv.y;
  uv.z += uv.z;
  const V3 c = cross(qv, uv);
  return {(v.x + q.w * uv.x) + c.x, (v.y + q.w * uv.y) + c.y, (v.z + q.w * uv.z) + c.z};
}

struct Se3 {
  Quat q;
  V3 t;
  // SE3::Apply, lie_group.cuh:33-36
  inline V3 apply(const V3& v) const {
    const V3 r = qrot(q, v);
    return {r.x + t.x, r.y + t.y, r.z + t.z};
  }
  // SE3::Inverse, lie_group.cuh:25-27: (R^-1, R^-1 * (-t)); Eigen inverse = conjugate / squaredNorm
  inline Se3 inverse() const {
    const float n2 = (q.x * q.x + q.y * q.y) + (q.z * q.z + q.w * q.w);
    Quat qi;
    if (n2 > 0.f) {
      qi = {(-q.x) / n2, (-q.y) / n2, (-q.z) / n2, q.w / n2};
    } else {
      qi = {0.f, 0.f, 0.f, 0.f};  // Eigen returns a zero quaternion for a zero input
    }
    const V3 nt{-t.x, -t.y, -t.z};
    return {qi, qrot(qi, nt)};
  }
};

struct Intr {
  float fx, fy, cx, cy;
  // CameraIntrinsics::operator*, camera.cuh:48-51
  inline V3 mul(const V3& v) const { return {fx * v.x + cx * v.z, fy * v.y + cy * v.z, v.z}; }
  // CameraIntrinsics::Inverse, camera.cuh:35-39
  inline Intr inverse() const {
    const float fxi = 1 / fx;
    const float fyi = 1 / fy;
    return {fxi, fyi, -cx * fxi, -cy * fyi};
  }
};

// float -> int with CUDA cvt.rzi semantics (NaN -> 0, saturating).
inline int f2i(float f) {
  if (f != f) return 0;
  if (f >= 2147483648.f) return 2147483647;
  if (f <= -2147483648.f) return (-2147483647 - 1);
  return (int)f;
}
inline int16_t f2s(float f) { return (int16_t)f2i(f); }

struct Entry {  // VoxelBlock, voxel_mem.cuh:75-95
  S3 pos;
  int16_t offset;
  int32_t idx;
};
static_assert(sizeof(Entry) == 12, "VoxelBlock is 12 bytes");

struct VisBlock {
  Entry blk;
  uint32_t entry;
};

}  // namespace

struct ratsdf_engine {
  float vs = 0, trunc = 0;
  int block_bits = 0, bucket_bits = 0;
  uint32_t num_block = 0, num_bucket = 0, num_entry = 0, bucket_mask = 0, entry_mask = 0;
  int shard_rank = 0, shard_count = 1, shard_slab_bits = 2;
  int threads = 1;

  Entry* table = nullptr;
  int* locks = nullptr;
  std::vector<uint32_t> locked;  // buckets locked in the current pass (fast ResetLocks)
  ratsdf_rgbw* rgbw = nullptr;
  float* tsdf = nullptr;
  float* segm = nullptr;
  int* heap = nullptr;
  int num_free = 0;

  std::vector<float> range;
  std::vector<VisBlock> visible;
  ratsdf_frame_stats stats{};
  int64_t totals[5] = {0, 0, 0, 0, 0};
  int sticky = RATSDF_OK;

  // ---- geometry helpers (voxel_mem.cuh:31-70, voxel_hash.cu:19-23) -------------------------
  inline uint32_t hash(const S3& p) const {
    return (((uint32_t)(int32_t)p.x * 73856093u) ^ ((uint32_t)(int32_t)p.y * 19349669u) ^
            ((uint32_t)(int32_t)p.z * 83492791u)) &
           bucket_mask;
  }
  inline bool owned(const S3& p) const {
    if (shard_count <= 1) return true;
    int s = ((int)p.x) >> shard_slab_bits;
    int m = s % shard_count;
    if (m < 0) m += shard_count;
    return m == shard_rank;
  }

  // ---- locks (voxel_hash.cu:7-12,35-38; atomicExch try-lock :71,93,94,128,147) ---------------
  inline bool try_lock(uint32_t b) {
    const int old = locks[b];
    locks[b] = 1;
    if (old == 0) locked.push_back(b);
    return old == 0;
  }
  void reset_locks() {
    for (uint32_t b : locked) locks[b] = 0;
    locked.clear();
  }

  // ---- pool (voxel_mem.cu:37-61) -------------------------------------------------------------
  int acquire_block() {
    const int idx = num_free;
    num_free -= 1;
    if (idx < 1) {  // reference: device assert; here: sticky error, no block handed out
      num_free += 1;
      sticky = RATSDF_ERR_POOL_EXHAUSTED;
      return -1;
    }
    const int b = heap[idx - 1];
    const size_t base = (size_t)b << 9;
    for (int i = 0; i < RATSDF_BLOCK_VOLUME; ++i) {
      rgbw[base + i].weight = 1;  // rgb deliberately left as is (voxel_mem.cu:43-51)
      tsdf[base + i] = -1;
      segm[base + i] = .5;
    }
    return b;
  }
  void release_block(int idx) {
    const int n = num_free;
    num_free += 1;
    heap[n] = idx;
  }

  // ---- VoxelHashTable::Allocate, voxel_hash.cu:46-108 ----------------------------------------
  // returns true if a block was inserted
  bool allocate(const S3& pos) {
    const uint32_t bucket = hash(pos);
    const uint32_t e0 = bucket << 1;
    for (int i = 0; i < 2; ++i) {
      const Entry& b = table[e0 + i];
      if (b.pos == pos && b.idx >= 0) return false;
    }
    uint32_t last = e0 + 1;
    while (table[last].offset) {
      last = (last + table[last].offset) & entry_mask;
      const Entry& b = table[last];
      if (b.pos == pos && b.idx >= 0) return false;
    }
    for (int i = 0; i < 2; ++i) {
      Entry& b = table[e0 + i];
      if (b.idx < 0) {
        if (try_lock(bucket)) {
          const int idx = acquire_block();
          if (idx < 0) return false;
          b.pos = pos;
          b.offset = 0;
          b.idx = idx;
          return true;
        }
        return false;
      }
    }
    stats.slow_requests += 1;  // chained-bucket path (diagnostic counter only)
    last = e0 + 1;
    while (table[last].offset) last = (last + table[last].offset) & entry_mask;
    const uint32_t bucket_last = last >> 1;
    uint32_t next = last;
    for (uint32_t guard = 0; guard < num_entry; ++guard) {  // reference: while (true)
      next = (next + 1) & entry_mask;
      if ((next & 1u) != 1u && table[next].idx < 0) {
        const uint32_t bucket_next = next >> 1;
        if (try_lock(bucket_last) && try_lock(bucket_next)) {
          const int idx = acquire_block();
          if (idx < 0) return false;
          Entry& bl = table[last];
          Entry& bn = table[next];
          const uint32_t wrap = next > last ? 0 : num_entry;
          bl.offset = (int16_t)(next + wrap - last);
          bn.pos = pos;
          bn.offset = 0;
          bn.idx = idx;
          return true;
        }
        return false;
      }
    }
    return false;  // no free slot anywhere (the reference would spin forever)
  }

  // ---- VoxelHashTable::Delete, voxel_hash.cu:110-159 -----------------------------------------
  bool erase(const S3& pos) {
    const uint32_t bucket = hash(pos);
    const uint32_t e0 = bucket << 1;
    {
      Entry& b = table[e0];
      if (b.pos == pos && b.idx >= 0) {
        release_block(b.idx);
        b.offset = 0;
        b.idx = -1;
        return true;
      }
    }
    uint32_t last = e0 + 1;
    Entry& head = table[last];
    if (head.pos == pos && head.idx >= 0) {
      if (try_lock(bucket)) {
        const uint32_t nxt = (last + head.offset) & entry_mask;
        Entry& bn = table[nxt];
        release_block(head.idx);
        head.pos = bn.pos;
        head.offset = bn.offset ? (int16_t)(head.offset + bn.offset) : (int16_t)0;
        head.idx = bn.idx;
        bn.offset = 0;
        bn.idx = -1;
        return true;
      }
      return false;
    }
    while (table[last].offset) {
      Entry& bl = table[last];
      const uint32_t cur = (last + bl.offset) & entry_mask;
      Entry& bc = table[cur];
      if (bc.pos == pos && bc.idx >= 0) {
        if (try_lock(bucket)) {
          bl.offset = bc.offset ? (int16_t)(bl.offset + bc.offset) : (int16_t)0;
          release_block(bc.idx);
          bc.offset = 0;
          bc.idx = -1;
          return true;
        }
        return false;
      }
      last = cur;
    }
    return false;
  }

  // ---- VoxelHashTable::GetBlock(pos, out), voxel_hash.cu:190-218 -----------------------------
  void get_block(const S3& pos, Entry* out) const {
    const uint32_t e0 = hash(pos) << 1;
    for (int i = 0; i < 2; ++i) {
      *out = table[e0 + i];
      if (out->pos == pos && out->idx >= 0) return;
    }
    uint32_t last = e0 + 1;
    while (table[last].offset) {
      last = (last + table[last].offset) & entry_mask;
      *out = table[last];
      if (out->pos == pos && out->idx >= 0) return;
    }
    out->pos = pos;
    out->offset = -1;
    out->idx = -1;
  }

  // ---- is_voxel_visible / is_block_visible, voxel_tsdf.cu:64-96 -------------------------------
  inline bool voxel_visible(int16_t gx, int16_t gy, int16_t gz, const Se3& T, const Intr& K, int W,
                            int H) const {
    const V3 pw{(float)gx * vs, (float)gy * vs, (float)gz * vs};
    const V3 pc = T.apply(pw);
    const V3 ph = K.mul(pc);
    const float u = ph.x / ph.z;
    const float v = ph.y / ph.z;
    return (u >= 0 && u <= (float)(W - 1) && v >= 0 && v <= (float)(H - 1) && ph.z >= 0);
  }
  template <bool Full>
  inline bool block_visible(const S3& bp, const Se3& T, const Intr& K, int W, int H) const {
    const int16_t x = (int16_t)(bp.x << 3), y = (int16_t)(bp.y << 3), z = (int16_t)(bp.z << 3);
    bool vis = Full;
    for (int i = 0; i < 8; ++i) {
      const int16_t cx = (int16_t)(x + ((i >> 0) & 1) * 7);
      const int16_t cy = (int16_t)(y + ((i >> 1) & 1) * 7);
      const int16_t cz = (int16_t)(z + ((i >> 2) & 1) * 7);
      const bool v = voxel_visible(cx, cy, cz, T, K, W, H);
      if (Full)
        vis &= v;
      else
        vis |= v;
    }
    return vis;
  }

  template <class F>
  void parallel_for(size_t n, F&& f) {
    const int nt = threads > 1 ? threads : 1;
    if (nt == 1 || n < 64) {
      f(0, n, 0);
      return;
    }
    std::vector<std::thread> pool;
    const size_t chunk = (n + nt - 1) / nt;
    for (int t = 0; t < nt; ++t) {
      const size_t lo = (size_t)t * chunk;
      const size_t hi = lo + chunk < n ? lo + chunk : n;
      if (lo >= hi) break;
      pool.emplace_back([&f, lo, hi, t]() { f(lo, hi, t); });
    }
    for (auto& th : pool) th.join();
  }

  // ---- block_allocate_kernel + TSDFGrid::Allocate, voxel_tsdf.cu:120-168,454-463 -------------
  // Candidate blocks of pixel rows [y0, y1) in raster order.  With `skip_present` the candidates
  // that already exist in the directory (or fail the frustum test) are dropped: Allocate() returns
  // at once for them without changing any state (voxel_hash.cu:48-65), so dropping them is
  // result-neutral; that is what lets the multithreaded baseline generate candidates in parallel.
  void candidates_rows(const float* depth, int y0, int y1, int W, int H, float md, const Intr& K,
                       const Intr& Ki, const Se3& T, const Se3& Ti, bool skip_present,
                       std::vector<S3>& out) {
    for (int y = y0; y < y1; ++y) {
      for (int x = 0; x < W; ++x) {
        const int idx = y * W + x;
        const float d = depth[idx];
        const V3 pimg{(float)x, (float)y, 1.f};
        const V3 pc = Ki.mul(pimg);
        const float r = std::sqrt(pc.x * pc.x + (pc.y * pc.y + pc.z * pc.z));  // Eigen norm()
        range[idx] = r;
        if (d == 0 || d > md) continue;
        const V3 pcd{pc.x * d, pc.y * d, pc.z * d};
        const V3 pw = Ti.apply(pcd);
        const V3 dc{pc.x / r, pc.y / r, pc.z / r};
        const V3 dw = qrot(Ti.q, dc);
        const V3 start_w{pw.x - dw.x * trunc, pw.y - dw.y * trunc, pw.z - dw.z * trunc};
        const V3 dg{dw.x / vs, dw.y / vs, dw.z / vs};
        const V3 sg{start_w.x / vs, start_w.y / vs, start_w.z / vs};
        const float two_tr = 2 * trunc;
        const V3 rg{two_tr * dg.x, two_tr * dg.y, two_tr * dg.z};
        const int steps =
            f2i(ceilf(fmaxf(fmaxf(fabsf(rg.x), fabsf(rg.y)), fabsf(rg.z)) / RATSDF_BLOCK_LEN));
        const float den = fmaxf((float)steps, 1);
        const V3 st{rg.x / den, rg.y / den, rg.z / den};
        V3 p = sg;
        S3 prev{0, 0, 0};
        bool have_prev = false;
        for (int i = 0; i <= steps; ++i) {
          const S3 g{f2s(roundf(p.x)), f2s(roundf(p.y)), f2s(roundf(p.z))};
          const S3 bp{(int16_t)(g.x >> 3), (int16_t)(g.y >> 3), (int16_t)(g.z >> 3)};
          p.x += st.x;
          p.y += st.y;
          p.z += st.z;
          if (skip_present) {
            if (have_prev && bp == prev) continue;  // same block as the previous sample
            prev = bp;
            have_prev = true;
            Entry tmp;
            get_block(bp, &tmp);
            if (tmp.idx >= 0) continue;
          }
          if (owned(bp) && block_visible<true>(bp, T, K, W, H)) out.push_back(bp);
        }
      }
    }
  }

  void allocate_pass(const float* depth, int H, int W, float md, const Intr& K, const Se3& T) {
    const Se3 Ti = T.inverse();
    const Intr Ki = K.inverse();
    range.resize((size_t)H * W);
    int inserted = 0;
    if (threads <= 1) {
      std::vector<S3> cand;
      for (int y = 0; y < H; ++y) {  // strictly sequential: candidate, Allocate, next candidate
        cand.clear();
        candidates_rows(depth, y, y + 1, W, H, md, K, Ki, T, Ti, false, cand);
        for (const S3& bp : cand)
          if (allocate(bp)) ++inserted;
      }
    } else {
      const int nt = threads;
      std::vector<std::vector<S3>> parts(nt);
      parallel_for((size_t)H, [&](size_t lo, size_t hi, int t) {
        candidates_rows(depth, (int)lo, (int)hi, W, H, md, K, Ki, T, Ti, true, parts[t]);
      });
      for (auto& part : parts)  // row chunks are in raster order
        for (const S3& bp : part)
          if (allocate(bp)) ++inserted;
    }
    stats.allocated_blocks = inserted;
    reset_locks();
  }

  // ---- check_visibility_kernel + GatherBlock, voxel_tsdf.cu:98-118,465-474,847-867 -----------
  void gather_visible(int H, int W, const Intr& K, const Se3& T) {
    const int nt = threads > 1 ? threads : 1;
    std::vector<std::vector<VisBlock>> parts(nt);
    parallel_for(num_entry, [&](size_t lo, size_t hi, int t) {
      auto& out = parts[t];
      for (size_t e = lo; e < hi; ++e) {
        const Entry& b = table[e];
        if (b.idx < 0) continue;
        if (block_visible<false>(b.pos, T, K, W, H)) out.push_back({b, (uint32_t)e});
      }
    });
    visible.clear();
    for (auto& p : parts) visible.insert(visible.end(), p.begin(), p.end());
    stats.visible_blocks = (int)visible.size();
  }

  // ---- tsdf_integrate_kernel, voxel_tsdf.cu:170-251 -------------------------------------------
  int integrate_block(const Entry& blk, const uint8_t* rgb, const float* depth, const float* ht,
                      const float* lt, int H, int W, float md, const Intr& K, const Se3& T) {
    int updated = 0;
    const size_t base = (size_t)blk.idx << 9;
    for (int tz = 0; tz < 8; ++tz)
      for (int ty = 0; ty < 8; ++ty)
        for (int tx = 0; tx < 8; ++tx) {
          const int16_t gx = (int16_t)((int16_t)(blk.pos.x << 3) + tx);
          const int16_t gy = (int16_t)((int16_t)(blk.pos.y << 3) + ty);
          const int16_t gz = (int16_t)((int16_t)(blk.pos.z << 3) + tz);
          const V3 pw{(float)gx * vs, (float)gy * vs, (float)gz * vs};
          const V3 pc = T.apply(pw);
          const V3 ph = K.mul(pc);
          const float pu = ph.x / ph.z;
          const float pv = ph.y / ph.z;
          const int u = f2i(roundf(pu));
          const int v = f2i(roundf(pv));
          if (!(u >= 0 && u < W && v >= 0 && v < H)) continue;
          const int k = v * W + u;
          const float d = depth[k];
          if (d == 0 || d > md) continue;
          const float sdf = range[k] * (d - ph.z);
          if (!(sdf > -trunc)) continue;
          const float ts = fminf(1, sdf / trunc);
          const int vi = tx + ty * 8 + tz * 64;
          ratsdf_rgbw& c = rgbw[base + vi];
          float& t = tsdf[base + vi];
          float& p = segm[base + vi];
          const float wn = (1 - d / md) * 4;
          const float wo = (float)c.weight;
          const float wc = wo + wn;
          const float rn = (float)rgb[3 * k + 0], gn = (float)rgb[3 * k + 1],
                      bn = (float)rgb[3 * k + 2];
          const float rc = ((float)c.r * wo + rn * wn) / wc;
          const float gc = ((float)c.g * wo + gn * wn) / wc;
          const float bc = ((float)c.b * wo + bn * wn) / wc;
          t = (t * wo + ts * wn) / wc;
          c.weight = (uint8_t)f2i(fminf(roundf(wc), 40));
          c.r = (uint8_t)f2i(roundf(rc));
          c.g = (uint8_t)f2i(roundf(gc));
          c.b = (uint8_t)f2i(roundf(bc));
          const float hk = ht ? ht[k] : 1.f;
          const float lk = lt ? lt[k] : 1.f;
          const float pos = expf((wo * logf(p) + wn * logf(hk)) / wc);
          const float neg = expf((wo * logf(1 - p) + wn * logf(lk)) / wc);
          p = pos / (pos + neg);
          ++updated;
        }
    return updated;
  }

  // ---- TSDFGrid::Integrate host sequence, voxel_tsdf.cu:416-452 -------------------------------
  int integrate(const uint8_t* rgb, const float* depth, const float* ht, const float* lt, int H,
                int W, float md, const Intr& K, const Se3& T) {
    stats = ratsdf_frame_stats{};
    const bool prof = getenv("RATSDF_ORACLE_PROFILE") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    allocate_pass(depth, H, W, md, K, T);
    const double t1 = now();
    gather_visible(H, W, K, T);
    const double t2 = now();
    const size_t V = visible.size();
    std::atomic<long> updated{0};
    std::vector<uint8_t> carve(V, 0);
    parallel_for(V, [&](size_t lo, size_t hi, int) {
      long u = 0;
      for (size_t i = lo; i < hi; ++i) {
        const Entry& blk = visible[i].blk;
        u += integrate_block(blk, rgb, depth, ht, lt, H, W, md, K, T);
        // space_carving_kernel, voxel_tsdf.cu:253-276: min |tsdf| over the block after the update
        const float* tp = tsdf + ((size_t)blk.idx << 9);
        float m = fabsf(tp[0]);
        for (int j = 1; j < RATSDF_BLOCK_VOLUME; ++j) m = fminf(m, fabsf(tp[j]));
        carve[i] = (m >= .9f) ? 1 : 0;
      }
      updated += u;
    });
    stats.updated_voxels = (int)updated.load();
    const double t3 = now();
    if (prof)
      fprintf(stderr, "[oracle] alloc %.1f ms, visible %.1f ms, integrate %.1f ms (threads %d)\n",
              (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, threads);
    int deleted = 0;
    for (size_t i = 0; i < V; ++i) {
      if (carve[i] && erase(visible[i].blk.pos)) ++deleted;
    }
    reset_locks();
    stats.deleted_blocks = deleted;
    stats.active_blocks = (int)num_block - num_free;
    totals[0] += 1;
    totals[1] += stats.visible_blocks;
    totals[2] += stats.updated_voxels;
    totals[3] += stats.allocated_blocks;
    totals[4] += stats.deleted_blocks;
    return sticky;
  }

  // ---- RetrieveMutable with a block cache, voxel_hash.cuh:124-143 -----------------------------
  // returns pool voxel index or -1
  long retrieve_index(const S3& point, Entry& cache) const {
    const S3 bp{(int16_t)(point.x >> 3), (int16_t)(point.y >> 3), (int16_t)(point.z >> 3)};
    const int vi = (point.x & 7) + (point.y & 7) * 8 + (point.z & 7) * 64;
    if (cache.pos == bp) {
      if (cache.idx >= 0) return ((long)cache.idx << 9) + vi;
      if (cache.offset < 0) return -1;
    }
    get_block(bp, &cache);
    if (cache.idx >= 0) return ((long)cache.idx << 9) + vi;
    return -1;
  }

  // ---- Retrieve<Voxel>(point, cache): value lookups used by the ray caster ---------------------
  inline float voxel_tsdf_at(const S3& p, Entry& cache) const {
    const long vi = retrieve_index(p, cache);
    return vi >= 0 ? tsdf[vi] : -10.f;  // VoxelTSDF() default, voxel_types.cu:8
  }
  // VoxelHashTable::RetrieveTSDF, voxel_hash.cu:161-188 (corner / weight pairing as written there)
  float retrieve_tsdf(const V3& pt, Entry& cache) const {
    const V3 pl{floorf(pt.x), floorf(pt.y), floorf(pt.z)};
    const V3 ph{pl.x + 1.f, pl.y + 1.f, pl.z + 1.f};
    const V3 al{ph.x - pt.x, ph.y - pt.y, ph.z - pt.z};
    float t[8];
    for (int i = 0; i < 8; ++i) {
      const S3 c{f2s((i >> 2) & 1 ? pl.x : ph.x), f2s((i >> 1) & 1 ? pl.y : ph.y),
                 f2s((i >> 0) & 1 ? pl.z : ph.z)};
      t[i] = voxel_tsdf_at(c, cache);
    }
    const float t00 = t[0] * al.z + t[1] * (1 - al.z);
    const float t01 = t[2] * al.z + t[3] * (1 - al.z);
    const float t10 = t[4] * al.z + t[5] * (1 - al.z);
    const float t11 = t[6] * al.z + t[7] * (1 - al.z);
    const float t0 = t00 * al.y + t01 * (1 - al.y);
    const float t1 = t10 * al.y + t11 * (1 - al.y);
    return t0 * al.x + t1 * (1 - al.x);
  }

  // ---- ray_cast_kernel, voxel_tsdf.cu:278-374 ---------------------------------------------------
  // rows [row0, row1) of the H x W image; the buffers hold those rows only
  void raycast(const Intr& K, int H, int W, const Se3& T, float max_depth, uint8_t* rgba,
               uint8_t* normal, int row0 = 0, int row1 = -1) {
    if (row1 < 0) row1 = H;
    const Se3 Ti = T.inverse();
    const Intr Ki = K.inverse();
    const float step_size = trunc / 2;  // voxel_tsdf.cu:892
    const int max_step = f2i(ceilf(max_depth / step_size));
    parallel_for((size_t)(row1 - row0), [&](size_t ylo, size_t yhi, int) {
      for (size_t yr = ylo; yr < yhi; ++yr)
        for (int x = 0; x < W; ++x) {
          const size_t y = yr + (size_t)row0;
          const size_t idx = yr * W + x;
          uint8_t out_c[4] = {0, 0, 0, 0}, out_n[4] = {0, 0, 0, 0};
          const V3 pc = Ki.mul(V3{(float)x, (float)y, 1.f});
          const float n2 = pc.x * pc.x + (pc.y * pc.y + pc.z * pc.z);
          V3 dc = pc;  // Eigen normalized(): v / sqrt(squaredNorm) when squaredNorm > 0
          if (n2 > 0.f) {
            const float nn = std::sqrt(n2);
            dc = V3{pc.x / nn, pc.y / nn, pc.z / nn};
          }
          const V3 dw = qrot(Ti.q, dc);
          const V3 full{dw.x * step_size / vs, dw.y * step_size / vs, dw.z * step_size / vs};
          V3 stepv = full;
          V3 p{Ti.t.x / vs, Ti.t.y / vs, Ti.t.z / vs};
          Entry cache{{0, 0, 0}, 0, -1};
          auto rnd = [](const V3& v) { return S3{f2s(roundf(v.x)), f2s(roundf(v.y)), f2s(roundf(v.z))}; };
          float prev = voxel_tsdf_at(rnd(p), cache);
          p = V3{p.x + stepv.x, p.y + stepv.y, p.z + stepv.z};
          for (int i = 1; i < max_step; ++i) {
            const S3 g = rnd(p);
            const float cur = voxel_tsdf_at(g, cache);
            const long wi = retrieve_index(g, cache);
            const uint8_t wcur = wi >= 0 ? rgbw[wi].weight : (uint8_t)0;
            if (wcur < 10) {
              p = V3{p.x + stepv.x, p.y + stepv.y, p.z + stepv.z};
              prev = cur;
              continue;
            }
            if (prev > 0 && cur <= 0 && prev - cur <= 2.0f) {
              const V3 p1{p.x - stepv.x, p.y - stepv.y, p.z - stepv.z};
              const float ac = retrieve_tsdf(p, cache);
              const float ap = retrieve_tsdf(p1, cache);
              const float f = ac / (ap - ac);
              const V3 pi{p.x + f * stepv.x, p.y + f * stepv.y, p.z + f * stepv.z};
              const S3 fg = rnd(pi);
              const long vi = retrieve_index(fg, cache);
              const ratsdf_rgbw c = vi >= 0 ? rgbw[vi] : ratsdf_rgbw{0, 0, 0, 0};
              const float prob = vi >= 0 ? segm[vi] : 0.f;
              auto at = [&](int dx, int dy, int dz) {
                return voxel_tsdf_at(S3{(int16_t)(fg.x + dx), (int16_t)(fg.y + dy), (int16_t)(fg.z + dz)},
                                     cache);
              };
              const V3 nr{at(1, 0, 0) - at(-1, 0, 0), at(0, 1, 0) - at(0, -1, 0),
                          at(0, 0, 1) - at(0, 0, -1)};
              const float dotv = nr.x * (-dw.x) + (nr.y * (-dw.y) + nr.z * (-dw.z));
              const float nn = std::sqrt(nr.x * nr.x + (nr.y * nr.y + nr.z * nr.z));
              const float diff = fmaxf(dotv / nn, 0);
              const float alpha = fmaxf(prob - 0.5f, 0) / .5f;
              out_c[0] = (uint8_t)f2i(alpha * 255 + (1 - alpha) * (float)c.r);
              out_c[1] = (uint8_t)f2i((1 - alpha) * (float)c.g);
              out_c[2] = (uint8_t)f2i((1 - alpha) * (float)c.b);
              out_c[3] = 255;
              out_n[0] = (uint8_t)f2i(alpha * 255 + (1 - alpha) * diff * 255);
              out_n[1] = (uint8_t)f2i((1 - alpha) * diff * 255);
              out_n[2] = out_n[1];
              out_n[3] = 255;
              break;
            }
            prev = cur;
            if (cur < 0.5f) {
              stepv = V3{full.x / 10, full.y / 10, full.z / 10};
            } else {
              stepv = full;
            }
            p = V3{p.x + stepv.x, p.y + stepv.y, p.z + stepv.z};
          }
          if (rgba) memcpy(rgba + idx * 4, out_c, 4);
          if (normal) memcpy(normal + idx * 4, out_n, 4);
        }
    });
  }

  // ---- marching_cube_kernel + GatherValidMesh, voxel_tsdf.cu:561-845 ---------------------------
  // Corner / edge conventions of the published table (mcube_table.cuh:14-36): corner i at
  // kCorner[i]; edge e joins kEdge[e]; a cube owns the three edges leaving its corner 0 along +x,
  // +y, +z, and every other edge is the owned edge of a neighbouring cube's corner 0.
  void gather_valid_mesh(std::vector<float>& out_v, std::vector<int32_t>& out_i,
                         std::vector<float>& out_p) {
    static const int kCorner[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 0, 1}, {0, 0, 1},
                                      {0, 1, 0}, {1, 1, 0}, {1, 1, 1}, {0, 1, 1}};
    static const int kEdge[12][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6},
                                     {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};
    static const char* const kCases[256] = {
#include "mc_cases.inc"
    };
    // lower corner and axis of every edge (edge_vertex_map, mcube_table.cuh:34-36), derived
    int edge_lower[12], edge_dim[12];
    for (int e = 0; e < 12; ++e) {
      const int* a = kCorner[kEdge[e][0]];
      const int* b = kCorner[kEdge[e][1]];
      int dim = 0;
      for (int d = 0; d < 3; ++d)
        if (a[d] != b[d]) dim = d;
      edge_dim[e] = dim;
      edge_lower[e] = a[dim] < b[dim] ? kEdge[e][0] : kEdge[e][1];
    }
    static const int kAxis[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};  // corners 1, 4, 3

    // (a sharded map meshes the blocks it owns; blocks imported from a neighbour's subvolume are only read)
    std::vector<Entry> blocks;
    for (uint32_t i = 0; i < num_entry; ++i)
      if (table[i].idx >= 0 && owned(table[i].pos)) blocks.push_back(table[i]);
    const size_t nb = blocks.size();
    const size_t VV = 729;  // BLOCK_VERT_VOLUME = 9^3
    std::vector<float> verts(nb * VV * 9), vprob(nb * VV * 3);
    std::vector<uint8_t> vmask(nb * VV * 3, 0), tmask(nb * 512 * 5, 0);
    std::vector<int32_t> tids(nb * 512 * 15, 0);
    parallel_for(nb, [&](size_t lo, size_t hi, int) {
      std::vector<float> ct(16 * 16 * 16), cp(16 * 16 * 16);
      auto at = [](int z, int y, int x) { return (z * 16 + y) * 16 + x; };
      for (size_t bi = lo; bi < hi; ++bi) {
        const Entry& base = blocks[bi];
        for (int i = 0; i < 2; ++i)
          for (int j = 0; j < 2; ++j)
            for (int k = 0; k < 2; ++k) {
              Entry nb_e;
              get_block(S3{(int16_t)(base.pos.x + k), (int16_t)(base.pos.y + j),
                           (int16_t)(base.pos.z + i)}, &nb_e);
              for (int tz = 0; tz < 8; ++tz)
                for (int ty = 0; ty < 8; ++ty)
                  for (int tx = 0; tx < 8; ++tx) {
                    float t = -10.f, pr = 0.f;
                    if (nb_e.idx >= 0) {
                      const size_t vi = ((size_t)nb_e.idx << 9) + tx + ty * 8 + tz * 64;
                      if ((int)rgbw[vi].weight > 10) {
                        t = tsdf[vi];
                        pr = segm[vi];
                      }
                    }
                    ct[at(i * 8 + tz, j * 8 + ty, k * 8 + tx)] = t;
                    cp[at(i * 8 + tz, j * 8 + ty, k * 8 + tx)] = pr;
                  }
            }
        // three candidate vertices per lattice point of the 9^3 vertex grid
        for (size_t c = 0; c < VV; ++c) {
          const int vx = (int)(c % 9), vy = (int)(c / 9 % 9), vz = (int)(c / 81);
          const float v1[3] = {(float)(int16_t)((int16_t)(base.pos.x << 3) + vx),
                               (float)(int16_t)((int16_t)(base.pos.y << 3) + vy),
                               (float)(int16_t)((int16_t)(base.pos.z << 3) + vz)};
          const float t1 = ct[at(vz, vy, vx)], p1 = cp[at(vz, vy, vx)];
          const size_t cube_idx = bi * VV + c;
          for (int j = 0; j < 3; ++j) {
            const int* o = kAxis[j];
            const float t2 = ct[at(vz + o[2], vy + o[1], vx + o[0])];
            const float p2 = cp[at(vz + o[2], vy + o[1], vx + o[0])];
            const float sfac = (-t1) / (t2 - t1);
            for (int d = 0; d < 3; ++d)
              verts[(cube_idx * 3 + j) * 3 + d] = (v1[d] + sfac * (float)o[d]) * vs;
            vprob[cube_idx * 3 + j] = (p1 + p2) / 2;
          }
        }
        for (int tz = 0; tz < 8; ++tz)
          for (int ty = 0; ty < 8; ++ty)
            for (int tx = 0; tx < 8; ++tx) {
              float lt[8];
              int cubeindex = 0;
              for (int i = 0; i < 8; ++i) {
                lt[i] = ct[at(tz + kCorner[i][2], ty + kCorner[i][1], tx + kCorner[i][0])];
                cubeindex |= (lt[i] < 0) << i;
              }
              const char* cs = kCases[cubeindex];
              const int ntri = (int)(strlen(cs) / 3);
              const size_t thread_idx = bi * 512 + (size_t)(tx + ty * 8 + tz * 64);
              for (int i = 0; i < 5; ++i) {
                const size_t tri = thread_idx * 5 + i;
                if (i >= ntri) continue;  // mask stays 0
                tmask[tri] = 1;
                for (int j = 0; j < 3; ++j) {
                  const char ch = cs[i * 3 + j];
                  const int e = ch <= '9' ? ch - '0' : ch - 'a' + 10;
                  const float diff = fabsf(lt[kEdge[e][1]] - lt[kEdge[e][0]]);
                  if ((double)diff < 1e-3 || diff >= 2) {  // :692-696
                    tmask[tri] = 0;
                    break;
                  }
                  const int* lo_c = kCorner[edge_lower[e]];
                  const size_t coi = (size_t)((lo_c[2] + tz) * 81 + (lo_c[1] + ty) * 9 + (lo_c[0] + tx));
                  const size_t cube_idx = bi * VV + coi;
                  tids[tri * 3 + j] = (int32_t)(cube_idx * 3 + edge_dim[e]);
                  vmask[cube_idx * 3 + edge_dim[e]] = 1;
                }
              }
            }
      }
    });
    // compaction (prefix_sum + compactify_kernel + transform_triangle_id_kernel, :782-813)
    std::vector<int32_t> vmap(vmask.size());
    int32_t run = 0;
    for (size_t i = 0; i < vmask.size(); ++i) {
      vmap[i] = run;
      run += vmask[i];
    }
    out_v.clear();
    out_p.clear();
    out_i.clear();
    out_v.reserve((size_t)run * 3);
    for (size_t i = 0; i < vmask.size(); ++i)
      if (vmask[i]) {
        out_v.insert(out_v.end(), &verts[i * 3], &verts[i * 3] + 3);
        out_p.push_back(vprob[i]);
      }
    for (size_t t = 0; t < tmask.size(); ++t)
      if (tmask[t])
        for (int j = 0; j < 3; ++j) out_i.push_back(vmap[(size_t)tids[t * 3 + j]]);
  }

  template <class Rec, bool Semantic>
  int download(const std::vector<Entry>& blocks, Rec** out, size_t* n) {
    const size_t cnt = blocks.size() * RATSDF_BLOCK_VOLUME;
    Rec* buf = (Rec*)malloc(cnt ? cnt * sizeof(Rec) : 1);
    if (!buf) return RATSDF_ERR_DEVICE;
    for (size_t b = 0; b < blocks.size(); ++b) {
      const Entry& blk = blocks[b];
      const size_t base = (size_t)blk.idx << 9;
      for (int tz = 0; tz < 8; ++tz)
        for (int ty = 0; ty < 8; ++ty)
          for (int tx = 0; tx < 8; ++tx) {
            const int vi = tx + ty * 8 + tz * 64;
            const int16_t gx = (int16_t)((int16_t)(blk.pos.x << 3) + tx);
            const int16_t gy = (int16_t)((int16_t)(blk.pos.y << 3) + ty);
            const int16_t gz = (int16_t)((int16_t)(blk.pos.z << 3) + tz);
            Rec& r = buf[b * RATSDF_BLOCK_VOLUME + vi];
            r.x = (float)gx * vs;
            r.y = (float)gy * vs;
            r.z = (float)gz * vs;
            r.tsdf = tsdf[base + vi];
            if constexpr (Semantic) r.prob = segm[base + vi];
          }
    }
    *out = buf;
    *n = cnt;
    return RATSDF_OK;
  }
};

// =================================== C ABI (prefix ratsdf_oracle_) ============================
extern "C" {

int ratsdf_oracle_create_ex(const ratsdf_config* cfg, ratsdf_engine** out) {
  if (!cfg || !out) return RATSDF_ERR_BAD_ARGUMENT;
  if (!(cfg->voxel_size > 0) || !(cfg->truncation > 0)) return RATSDF_ERR_BAD_ARGUMENT;
  const int bb = cfg->block_bits ? cfg->block_bits : RATSDF_DEFAULT_BLOCK_BITS;
  const int kb = cfg->bucket_bits ? cfg->bucket_bits : RATSDF_DEFAULT_BUCKET_BITS;
  if (bb < 1 || bb > 22 || kb < 2 || kb > 26) return RATSDF_ERR_BAD_ARGUMENT;
  if (cfg->shard_count > 1 && (cfg->shard_rank < 0 || cfg->shard_rank >= cfg->shard_count))
    return RATSDF_ERR_BAD_ARGUMENT;
  ratsdf_engine* e = new ratsdf_engine();
  e->vs = cfg->voxel_size;
  e->trunc = cfg->truncation;
  e->block_bits = bb;
  e->bucket_bits = kb;
  e->num_block = 1u << bb;
  e->num_bucket = 1u << kb;
  e->num_entry = e->num_bucket << 1;
  e->bucket_mask = e->num_bucket - 1;
  e->entry_mask = e->num_entry - 1;
  e->shard_rank = cfg->shard_rank;
  e->shard_count = cfg->shard_count > 1 ? cfg->shard_count : 1;
  e->shard_slab_bits = cfg->shard_slab_bits > 0 ? cfg->shard_slab_bits : 2;
  e->threads = cfg->threads > 1 ? cfg->threads : 1;
  // calloc: pages stay untouched (not resident) until a block is actually used
  e->table = (Entry*)calloc(e->num_entry, sizeof(Entry));
  e->locks = (int*)calloc(e->num_bucket, sizeof(int));
  const size_t nvox = (size_t)e->num_block << 9;
  e->rgbw = (ratsdf_rgbw*)calloc(nvox, sizeof(ratsdf_rgbw));
  e->tsdf = (float*)calloc(nvox, sizeof(float));
  e->segm = (float*)calloc(nvox, sizeof(float));
  e->heap = (int*)malloc(sizeof(int) * e->num_block);
  if (!e->table || !e->locks || !e->rgbw || !e->tsdf || !e->segm || !e->heap) {
    free(e->table); free(e->locks); free(e->rgbw); free(e->tsdf); free(e->segm); free(e->heap);
    delete e;
    return RATSDF_ERR_DEVICE;
  }
  for (uint32_t i = 0; i < e->num_entry; ++i) e->table[i].idx = -1;  // init_hash_table_kernel
  for (uint32_t i = 0; i < e->num_block; ++i) e->heap[i] = (int)i;   // heap_init_kernel
  e->num_free = (int)e->num_block;
  *out = e;
  return RATSDF_OK;
}

int ratsdf_oracle_create(float voxel_size, float truncation, int device, ratsdf_engine** out) {
  ratsdf_config c;
  memset(&c, 0, sizeof(c));
  c.voxel_size = voxel_size;
  c.truncation = truncation;
  c.device = device;
  return ratsdf_oracle_create_ex(&c, out);
}

int ratsdf_oracle_destroy(ratsdf_engine* e) {
  if (!e) return RATSDF_ERR_BAD_ARGUMENT;
  free(e->table); free(e->locks); free(e->rgbw); free(e->tsdf); free(e->segm); free(e->heap);
  delete e;
  return RATSDF_OK;
}

int ratsdf_oracle_integrate(ratsdf_engine* e, const uint8_t* rgb, const float* depth,
                            const float* ht, const float* lt, int height, int width,
                            float max_depth, const ratsdf_intrinsics* K, const ratsdf_pose* P) {
  if (!e || !rgb || !depth || !K || !P || height <= 0 || width <= 0)
    return RATSDF_ERR_BAD_ARGUMENT;
  if (!ht || !lt) ht = lt = nullptr;  // tsdf_module.cc:27-31: either missing -> both ones
  const Intr Ki{K->fx, K->fy, K->cx, K->cy};
  const Se3 T{{P->qx, P->qy, P->qz, P->qw}, {P->tx, P->ty, P->tz}};
  return e->integrate(rgb, depth, ht, lt, height, width, max_depth, Ki, T);
}

int ratsdf_oracle_integrate_device(ratsdf_engine*, const void*, const void*, const void*,
                                   const void*, int, int, float, const ratsdf_intrinsics*,
                                   const ratsdf_pose*) {
  return RATSDF_ERR_NOT_IMPLEMENTED;
}
int ratsdf_oracle_integrate_device_batch(ratsdf_engine*, int, const void* const*, const void* const*,
                                         const void* const*, const void* const*, int, int, float,
                                         const ratsdf_intrinsics*, const ratsdf_pose*) {
  return RATSDF_ERR_NOT_IMPLEMENTED;
}
int ratsdf_oracle_integrate_batch(ratsdf_engine* e, int n, const uint8_t* const* rgb,
                                  const float* const* depth, const float* const* ht,
                                  const float* const* lt, int height, int width, float max_depth,
                                  const ratsdf_intrinsics* K, const ratsdf_pose* P, int) {
  if (!e || n < 0 || (n > 0 && (!rgb || !depth || !K || !P))) return RATSDF_ERR_BAD_ARGUMENT;
  for (int i = 0; i < n; ++i) {  // frame by frame: the definition the batched engine must match
    const int st = ratsdf_oracle_integrate(e, rgb[i], depth[i], (ht && lt) ? ht[i] : nullptr,
                                           (ht && lt) ? lt[i] : nullptr, height, width, max_depth, &K[i],
                                           &P[i]);
    if (st != RATSDF_OK) return st;
  }
  return RATSDF_OK;
}
int ratsdf_oracle_host_alloc(size_t bytes, void** out) {
  if (!out) return RATSDF_ERR_BAD_ARGUMENT;
  *out = bytes ? malloc(bytes) : nullptr;
  return (*out || !bytes) ? RATSDF_OK : RATSDF_ERR_DEVICE;
}
int ratsdf_oracle_host_free(void* p) {
  free(p);
  return RATSDF_OK;
}
int ratsdf_oracle_synchronize(ratsdf_engine* e) { return e ? e->sticky : RATSDF_ERR_BAD_ARGUMENT; }
int ratsdf_oracle_stream(ratsdf_engine*, void** s) {
  if (s) *s = nullptr;
  return RATSDF_ERR_NOT_IMPLEMENTED;
}

int ratsdf_oracle_prepare_device_batch(ratsdf_engine*, int, int, int) { return RATSDF_ERR_NOT_IMPLEMENTED; }
int ratsdf_oracle_profile_enable(ratsdf_engine*, int) { return RATSDF_ERR_NOT_IMPLEMENTED; }
int ratsdf_oracle_pipeline_counters(ratsdf_engine*, int64_t*, int) { return RATSDF_ERR_NOT_IMPLEMENTED; }
int ratsdf_oracle_export_directory_delta_device(ratsdf_engine*, void*, int32_t, void*) { return RATSDF_ERR_NOT_IMPLEMENTED; }
// (nothing to recover from: the oracle has no in-launch waits and no sticky device errors)
int ratsdf_oracle_recover(ratsdf_engine* e) { return e ? RATSDF_OK : RATSDF_ERR_BAD_ARGUMENT; }
// (device-resident block exchange of the across-shard exports: the oracle has no device)
int ratsdf_oracle_export_blocks_device(ratsdf_engine*, int32_t, const void*, void*, void*) { return RATSDF_ERR_NOT_IMPLEMENTED; }
int ratsdf_oracle_import_blocks_device(ratsdf_engine*, int32_t, const void*, const void*) { return RATSDF_ERR_NOT_IMPLEMENTED; }
int ratsdf_oracle_profile_read(ratsdf_engine*, double*, int64_t*) {
  return RATSDF_ERR_NOT_IMPLEMENTED;
}
int ratsdf_oracle_profile_read_frames(ratsdf_engine*, float*, float*, int, int*) {
  return RATSDF_ERR_NOT_IMPLEMENTED;
}

// groups are a device-launch construct of the HIP engine (several maps per launch): no CPU counterpart
int ratsdf_oracle_group_create(ratsdf_engine* const*, int, ratsdf_group** out) {
  if (out) *out = nullptr;
  return RATSDF_ERR_NOT_IMPLEMENTED;
}
int ratsdf_oracle_group_destroy(ratsdf_group*) { return RATSDF_ERR_NOT_IMPLEMENTED; }
int ratsdf_oracle_group_size(ratsdf_group*, int32_t*) { return RATSDF_ERR_NOT_IMPLEMENTED; }
int ratsdf_oracle_group_integrate_device_batch(ratsdf_group*, int, const void* const*,
                                               const void* const*, const void* const*,
                                               const void* const*, int, int, float,
                                               const ratsdf_intrinsics*, const ratsdf_pose*) {
  return RATSDF_ERR_NOT_IMPLEMENTED;
}
int ratsdf_oracle_group_synchronize(ratsdf_group*) { return RATSDF_ERR_NOT_IMPLEMENTED; }
int ratsdf_oracle_group_profile_enable(ratsdf_group*, int) { return RATSDF_ERR_NOT_IMPLEMENTED; }
int ratsdf_oracle_group_profile_read(ratsdf_group*, double*, int64_t*) {
  return RATSDF_ERR_NOT_IMPLEMENTED;
}

int ratsdf_oracle_num_active_blocks(ratsdf_engine* e, int32_t* out) {
  if (!e || !out) return RATSDF_ERR_BAD_ARGUMENT;
  *out = (int32_t)e->num_block - e->num_free;
  return RATSDF_OK;
}

int ratsdf_oracle_last_frame_stats(ratsdf_engine* e, ratsdf_frame_stats* out) {
  if (!e || !out) return RATSDF_ERR_BAD_ARGUMENT;
  *out = e->stats;
  return RATSDF_OK;
}

int ratsdf_oracle_totals(ratsdf_engine* e, int64_t* out5, int reset) {
  if (!e) return RATSDF_ERR_BAD_ARGUMENT;
  for (int i = 0; i < 5; ++i) {
    if (out5) out5[i] = e->totals[i];
    if (reset) e->totals[i] = 0;
  }
  return RATSDF_OK;
}

// check_bound_kernel + GatherVoxels, voxel_tsdf.cu:15-26,532-559; BoundingCube::Scale :28-33
int ratsdf_oracle_query(ratsdf_engine* e, const ratsdf_bounds* b, ratsdf_voxel_tsdf** out,
                        size_t* n) {
  if (!e || !b || !out || !n) return RATSDF_ERR_BAD_ARGUMENT;
  const float scale = (float)(1. / e->vs);
  const int16_t xmin = f2s(b->xmin * scale), xmax = f2s(b->xmax * scale);
  const int16_t ymin = f2s(b->ymin * scale), ymax = f2s(b->ymax * scale);
  const int16_t zmin = f2s(b->zmin * scale), zmax = f2s(b->zmax * scale);
  std::vector<Entry> sel;
  for (uint32_t i = 0; i < e->num_entry; ++i) {
    const Entry& blk = e->table[i];
    const int16_t gx = (int16_t)(blk.pos.x << 3), gy = (int16_t)(blk.pos.y << 3),
                  gz = (int16_t)(blk.pos.z << 3);
    if (blk.idx >= 0 && gx >= xmin && gy >= ymin && gz >= zmin && gx + 8 - 1 <= xmax &&
        gy + 8 - 1 <= ymax && gz + 8 - 1 <= zmax)
      sel.push_back(blk);
  }
  return e->download<ratsdf_voxel_tsdf, false>(sel, out, n);
}

// check_valid_kernel + GatherValid, voxel_tsdf.cu:28-33,476-502
int ratsdf_oracle_gather_valid(ratsdf_engine* e, ratsdf_voxel_tsdf** out, size_t* n) {
  if (!e || !out || !n) return RATSDF_ERR_BAD_ARGUMENT;
  std::vector<Entry> sel;
  for (uint32_t i = 0; i < e->num_entry; ++i)
    if (e->table[i].idx >= 0) sel.push_back(e->table[i]);
  return e->download<ratsdf_voxel_tsdf, false>(sel, out, n);
}

// GatherValidSemantic, voxel_tsdf.cu:49-62,504-530
int ratsdf_oracle_gather_valid_semantic(ratsdf_engine* e, ratsdf_voxel_segm** out, size_t* n) {
  if (!e || !out || !n) return RATSDF_ERR_BAD_ARGUMENT;
  std::vector<Entry> sel;
  for (uint32_t i = 0; i < e->num_entry; ++i)
    if (e->table[i].idx >= 0) sel.push_back(e->table[i]);
  return e->download<ratsdf_voxel_segm, true>(sel, out, n);
}

// TSDFSystem::DownloadAll, tsdf_module.cc:57-64
int ratsdf_oracle_download_all(ratsdf_engine* e, const char* path) {
  if (!e || !path) return RATSDF_ERR_BAD_ARGUMENT;
  ratsdf_voxel_segm* buf = nullptr;
  size_t n = 0;
  const int st = ratsdf_oracle_gather_valid_semantic(e, &buf, &n);
  if (st != RATSDF_OK) return st;
  FILE* f = fopen(path, "wb");
  if (!f) {
    free(buf);
    return RATSDF_ERR_BAD_ARGUMENT;
  }
  fwrite(buf, sizeof(ratsdf_voxel_segm), n, f);
  fclose(f);
  free(buf);
  return RATSDF_OK;
}

int ratsdf_oracle_free_buffer(void* p) {
  free(p);
  return RATSDF_OK;
}

int ratsdf_oracle_raycast(ratsdf_engine* e, const ratsdf_intrinsics* K, int height, int width,
                          const ratsdf_pose* P, float max_depth, uint8_t* rgba, uint8_t* normal) {
  if (!e || !K || !P || height <= 0 || width <= 0 || !(max_depth > 0)) return RATSDF_ERR_BAD_ARGUMENT;
  e->raycast(Intr{K->fx, K->fy, K->cx, K->cy}, height, width,
             Se3{{P->qx, P->qy, P->qz, P->qw}, {P->tx, P->ty, P->tz}}, max_depth, rgba, normal);
  return RATSDF_OK;
}
int ratsdf_oracle_raycast_rows(ratsdf_engine* e, const ratsdf_intrinsics* K, int height, int width,
                               const ratsdf_pose* P, float max_depth, int row0, int row1, uint8_t* rgba,
                               uint8_t* normal) {
  if (!e || !K || !P || height <= 0 || width <= 0 || !(max_depth > 0) || row0 < 0 || row1 > height || row0 > row1)
    return RATSDF_ERR_BAD_ARGUMENT;
  if (row0 == row1) return RATSDF_OK;
  e->raycast(Intr{K->fx, K->fy, K->cx, K->cy}, height, width,
             Se3{{P->qx, P->qy, P->qz, P->qw}, {P->tx, P->ty, P->tz}}, max_depth, rgba, normal, row0, row1);
  return RATSDF_OK;
}
int ratsdf_oracle_raycast_device(ratsdf_engine*, const ratsdf_intrinsics*, int, int,
                                 const ratsdf_pose*, float, void*, void*) {
  return RATSDF_ERR_NOT_IMPLEMENTED;
}

int ratsdf_oracle_gather_valid_mesh(ratsdf_engine* e, float** vertices, size_t* n_vertices,
                                    int32_t** indices, size_t* n_triangles, float** vertex_prob) {
  if (!e || !vertices || !n_vertices || !indices || !n_triangles || !vertex_prob)
    return RATSDF_ERR_BAD_ARGUMENT;
  std::vector<float> v, p;
  std::vector<int32_t> idx;
  e->gather_valid_mesh(v, idx, p);
  float* bv = (float*)malloc(v.size() * 4 + 4);
  float* bp = (float*)malloc(p.size() * 4 + 4);
  int32_t* bi = (int32_t*)malloc(idx.size() * 4 + 4);
  if (!bv || !bp || !bi) return RATSDF_ERR_DEVICE;
  memcpy(bv, v.data(), v.size() * 4);
  memcpy(bp, p.data(), p.size() * 4);
  memcpy(bi, idx.data(), idx.size() * 4);
  *vertices = bv;
  *vertex_prob = bp;
  *indices = bi;
  *n_vertices = p.size();
  *n_triangles = idx.size() / 3;
  return RATSDF_OK;
}

static int write_mesh_files(const char* vp, const char* ip, const char* pp, const float* v, size_t nv,
                            const int32_t* idx, size_t nt, const float* pr) {
  FILE* fv = fopen(vp, "wb");
  FILE* fp = fopen(pp, "wb");
  FILE* fi = fopen(ip, "wb");
  const bool ok = fv && fp && fi;
  if (ok) {
    fwrite(v, 12, nv, fv);
    fwrite(pr, 4, nv, fp);
    fwrite(idx, 12, nt, fi);
  }
  if (fv) fclose(fv);
  if (fp) fclose(fp);
  if (fi) fclose(fi);
  return ok ? RATSDF_OK : RATSDF_ERR_BAD_ARGUMENT;
}

int ratsdf_oracle_download_all_mesh(ratsdf_engine* e, const char* vp, const char* ip,
                                    const char* pp) {
  if (!e || !vp || !ip || !pp) return RATSDF_ERR_BAD_ARGUMENT;
  float *v = nullptr, *pr = nullptr;
  int32_t* idx = nullptr;
  size_t nv = 0, nt = 0;
  int st = ratsdf_oracle_gather_valid_mesh(e, &v, &nv, &idx, &nt, &pr);
  if (st == RATSDF_OK) st = write_mesh_files(vp, ip, pp, v, nv, idx, nt, pr);
  free(v);
  free(pr);
  free(idx);
  return st;
}

int ratsdf_oracle_export_directory_device(ratsdf_engine*, void*, int32_t, void*) {
  return RATSDF_ERR_NOT_IMPLEMENTED;
}

int ratsdf_oracle_test_allocate(ratsdf_engine* e, const int16_t* bp, int32_t n) {
  if (!e || (!bp && n > 0) || n < 0) return RATSDF_ERR_BAD_ARGUMENT;
  for (int i = 0; i < n; ++i) {
    const S3 p{bp[3 * i], bp[3 * i + 1], bp[3 * i + 2]};
    if (e->owned(p)) e->allocate(p);
  }
  e->reset_locks();
  return e->sticky;
}

int ratsdf_oracle_import_blocks(ratsdf_engine* e, int32_t n, const int16_t* bp, const float* tsdf,
                                const ratsdf_rgbw* rgbw, const float* prob) {
  if (!e || n < 0 || (n > 0 && (!bp || !tsdf || !rgbw || !prob))) return RATSDF_ERR_BAD_ARGUMENT;
  for (int i = 0; i < n; ++i) {
    const S3 p{bp[3 * i], bp[3 * i + 1], bp[3 * i + 2]};
    Entry b;
    e->get_block(p, &b);
    for (int pass = 0; b.idx < 0 && pass < 4; ++pass) {  // (an insertion can lose its bucket's lock: next pass)
      e->allocate(p);
      e->reset_locks();
      e->get_block(p, &b);
    }
    if (b.idx < 0) return e->sticky != RATSDF_OK ? e->sticky : RATSDF_ERR_CAPACITY;
    const size_t base = (size_t)b.idx << 9;
    memcpy(e->tsdf + base, tsdf + (size_t)i * 512, 512 * sizeof(float));
    memcpy(e->rgbw + base, rgbw + (size_t)i * 512, 512 * sizeof(ratsdf_rgbw));
    memcpy(e->segm + base, prob + (size_t)i * 512, 512 * sizeof(float));
  }
  return e->sticky;
}

int ratsdf_oracle_test_delete(ratsdf_engine* e, const int16_t* bp, int32_t n) {
  if (!e || (!bp && n > 0) || n < 0) return RATSDF_ERR_BAD_ARGUMENT;
  // A carve pass works on a list built BEFORE any deletion and ordered by hash entry
  // (GatherBlock, voxel_tsdf.cu:847-867): look every block up first, then delete in that order.
  std::vector<std::pair<uint32_t, S3>> order;
  for (int i = 0; i < n; ++i) {
    const S3 p{bp[3 * i], bp[3 * i + 1], bp[3 * i + 2]};
    const uint32_t e0 = e->hash(p) << 1;
    uint32_t found = 0xFFFFFFFFu;
    for (int k = 0; k < 2 && found == 0xFFFFFFFFu; ++k)
      if (e->table[e0 + k].pos == p && e->table[e0 + k].idx >= 0) found = e0 + k;
    uint32_t last = e0 + 1;
    while (found == 0xFFFFFFFFu && e->table[last].offset) {
      last = (last + e->table[last].offset) & e->entry_mask;
      if (e->table[last].pos == p && e->table[last].idx >= 0) found = last;
    }
    if (found == 0xFFFFFFFFu) continue;
    bool dup = false;
    for (auto& o : order) dup = dup || o.first == found;
    if (!dup) order.emplace_back(found, p);
  }
  std::sort(order.begin(), order.end(),
            [](const std::pair<uint32_t, S3>& a, const std::pair<uint32_t, S3>& b) {
              return a.first < b.first;
            });
  for (auto& o : order) e->erase(o.second);
  e->reset_locks();
  return e->sticky;
}

int ratsdf_oracle_test_retrieve(ratsdf_engine* e, const int16_t* pts, int32_t n, ratsdf_rgbw* rgbw,
                                float* tsdf, float* prob, ratsdf_block* blocks) {
  if (!e || (!pts && n > 0) || n < 0) return RATSDF_ERR_BAD_ARGUMENT;
  for (int i = 0; i < n; ++i) {
    const S3 p{pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
    Entry cache{{0, 0, 0}, 0, -1};  // VoxelBlock() default, voxel_mem.cuh:87
    const long vi = e->retrieve_index(p, cache);
    if (rgbw) rgbw[i] = vi >= 0 ? e->rgbw[vi] : ratsdf_rgbw{0, 0, 0, 0};  // voxel_types.cu:3
    if (tsdf) tsdf[i] = vi >= 0 ? e->tsdf[vi] : -10.f;                      // voxel_types.cu:8
    if (prob) prob[i] = vi >= 0 ? e->segm[vi] : 0.f;                        // voxel_types.cu:11
    if (blocks) blocks[i] = {cache.pos.x, cache.pos.y, cache.pos.z, cache.offset, cache.idx};
  }
  return RATSDF_OK;
}

int ratsdf_oracle_test_assign_rgbw(ratsdf_engine* e, const int16_t* pts, const ratsdf_rgbw* vals,
                                   int32_t n) {
  if (!e || ((!pts || !vals) && n > 0) || n < 0) return RATSDF_ERR_BAD_ARGUMENT;
  for (int i = 0; i < n; ++i) {
    const S3 p{pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
    Entry cache{{0, 0, 0}, 0, -1};
    const long vi = e->retrieve_index(p, cache);
    if (vi >= 0) e->rgbw[vi] = vals[i];
  }
  return RATSDF_OK;
}

int ratsdf_oracle_dump_directory(ratsdf_engine* e, int32_t** entry_index, ratsdf_block** blocks,
                                 size_t* n) {
  if (!e || !entry_index || !blocks || !n) return RATSDF_ERR_BAD_ARGUMENT;
  size_t cnt = 0;
  for (uint32_t i = 0; i < e->num_entry; ++i) cnt += e->table[i].idx >= 0;
  int32_t* ei = (int32_t*)malloc(cnt ? cnt * sizeof(int32_t) : 1);
  ratsdf_block* bl = (ratsdf_block*)malloc(cnt ? cnt * sizeof(ratsdf_block) : 1);
  size_t k = 0;
  for (uint32_t i = 0; i < e->num_entry; ++i) {
    const Entry& b = e->table[i];
    if (b.idx < 0) continue;
    ei[k] = (int32_t)i;
    bl[k] = {b.pos.x, b.pos.y, b.pos.z, b.offset, b.idx};
    ++k;
  }
  *entry_index = ei;
  *blocks = bl;
  *n = cnt;
  return RATSDF_OK;
}

int ratsdf_oracle_dump_voxels(ratsdf_engine* e, const int32_t* pool_idx, int32_t n, float* tsdf,
                              ratsdf_rgbw* rgbw, float* prob) {
  if (!e || (!pool_idx && n > 0) || n < 0) return RATSDF_ERR_BAD_ARGUMENT;
  for (int i = 0; i < n; ++i) {
    if (pool_idx[i] < 0 || (uint32_t)pool_idx[i] >= e->num_block) return RATSDF_ERR_BAD_ARGUMENT;
    const size_t base = (size_t)pool_idx[i] << 9;
    const size_t o = (size_t)i << 9;
    if (tsdf) memcpy(tsdf + o, e->tsdf + base, 512 * sizeof(float));
    if (rgbw) memcpy(rgbw + o, e->rgbw + base, 512 * sizeof(ratsdf_rgbw));
    if (prob) memcpy(prob + o, e->segm + base, 512 * sizeof(float));
  }
  return RATSDF_OK;
}

int ratsdf_oracle_dump_heap(ratsdf_engine* e, int32_t* num_free, int32_t* heap) {
  if (!e) return RATSDF_ERR_BAD_ARGUMENT;
  if (num_free) *num_free = e->num_free;
  if (heap) memcpy(heap, e->heap, sizeof(int32_t) * e->num_block);
  return RATSDF_OK;
}

const char* ratsdf_oracle_status_string(int s) {
  switch (s) {
    case RATSDF_OK: return "ok";
    case RATSDF_ERR_BAD_ARGUMENT: return "bad argument";
    case RATSDF_ERR_DEVICE: return "device / allocation error";
    case RATSDF_ERR_POOL_EXHAUSTED: return "voxel block pool exhausted";
    case RATSDF_ERR_CAPACITY: return "internal work list overflow";
    case RATSDF_ERR_NO_DEVICE: return "no device";
    case RATSDF_ERR_NOT_IMPLEMENTED: return "not implemented";
    case RATSDF_ERR_TIMEOUT: return "in-launch wait between workgroups timed out";
    default: return "unknown status";
  }
}
const char* ratsdf_oracle_backend(void) { return "cpu-oracle"; }

}  // extern "C"
