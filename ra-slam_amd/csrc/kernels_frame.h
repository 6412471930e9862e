// kernels_frame.h -- the launches of a frame besides k_integrate.  A frame is TWO launches:
//
//   k_front(f)       reads the directory: visible list (visible_append_role) + allocation requests
//                    from the frame's candidate set (cand_consume_role); first makes sure the queued
//                    head / chain deletes of frame f-1 have happened (carve_resolve_gate); pool
//                    releases of frame f-1 (carve_release_role)
//   k_integrate(f)   workgroup 0: the frame's serial role (statistics of frame f-1, allocation order
//                    of frame f; kernels_integrate.h: serial_role256); everybody else: voxel update of
//                    the visible blocks, then -- once the serial role has published -- commit and
//                    first update of the new blocks; start of the carve pass
//
// k_front leaves most of the chip idle and k_integrate is latency-bound, so when the caller has already
// handed over frame f+1 (ratsdf_integrate_device_batch) its directory-independent candidate pass
// (cand_pixels_role) rides along as extra workgroups: `ahead` describes the share each launch takes.
// A frame nobody looked ahead for (a single call, the first frame of a batch) is two launches as well: its own
// candidate pass runs inside k_front_inline, every pixel workgroup its own consumer (cand_inline_role).
//
// k_alloc_rank is the serial role as a launch of its own (1024 threads, its scratch in LDS): what the
// stand-alone test hooks run, and RATSDF_FUSED_SERIAL=0 puts it back between the two launches for A/B
// measurements (RATSDF_VPL=1 as well).
#pragma once
#include "kernels_integrate.h"

namespace ratsdf {

// One frame of one stream as the kernels see it when their operands live in device memory (launches
// that serve several engines at once, one engine per blockIdx.y: "frame-batched" integration of
// concurrent streams): the parameters of the frame, its input images, and which of the engine's two
// texel / candidate / counter sets it uses.
struct FrameJob {
  FrameParams P;
  const float* depth;
  const uint8_t* rgb;
  const float* ht;
  const float* lt;
  uint32_t par;
  uint32_t pad;
};
typedef const FrameJob __attribute__((address_space(4))) * JobPtr;

// which tiles of the look-ahead candidate pass a launch hosts (same for every engine of the launch)
struct AheadGeom {
  uint32_t first_tile, n_tiles, tiles_per_wg, tiles_x, tiles_x_magic;
};

// the candidate pass of frame `J` of engine `E`, restricted to `g`
__device__ inline CandJob make_cand_job(EnginePtr E, JobPtr J, const AheadGeom& g) {
  CandJob j;
  j.P = ld_const(&J->P);
  j.depth = J->depth;
  j.rgb = J->rgb;
  j.ht = J->ht;
  j.lt = J->lt;
  const uint32_t par = J->par;
  j.texA = E->texA[par];
  j.texB = E->texB[par];
  j.set = ld_const(&E->cand[par]);
  j.first_tile = g.first_tile;
  j.n_tiles = g.n_tiles;
  j.tiles_per_wg = g.tiles_per_wg;
  j.tiles_x = g.tiles_x;
  j.tiles_x_magic = g.tiles_x_magic;
  return j;
}

// ---------------------------------------------------------------------------------------------
// The frame's serial role at the TAIL of k_front (ordinary frames).
//
// Until round 4 the role always ran as workgroup 0 of k_integrate, beside the voxel update: the blocks that
// existed before the frame were through ~7 us into that launch, the role published at 7.3 us, and the
// frame's NEW blocks (a tenth of its blocks) finished at 14 us behind five dependent round trips (request ->
// winners counted -> pool index -> voxel words / texels -> stores).  Everything the role needs exists when
// the last directory workgroup of k_front (visible list, candidate consumers, pool releases) has finished,
// so that workgroup -- whichever it is: they count themselves in, FrameCtl::arrive -- now runs it: winners
// (claim == own rank), their order (smaller ranks counted in LDS), pool indices, directory entries,
// occupancy bits, claim resets, and one work-list item per new block (per-XCD lists of their own behind the
// visible ones: the update of such a block starts from AquireBlock's initial values).  The launch boundary is
// the hand-off: k_integrate sees work lists, nothing in it waits or polls (FrameCtl::front_done tells its
// serial group to stay out).
// Also here, as in serial_role256: the previous frame's statistics and the reset of its counters.
// Frames the tail does NOT take (it returns before it has changed anything, front_done stays 0 and the role
// runs inside k_integrate as before): chained-bucket requests (the resolver), more than kTailReqMax requests
// or kTailWinMax winners (a new view), a previous frame with more deletes than the release role takes.
// What other workgroups of THIS launch wrote is read past the caches (agent-scope loads; the writers use
// write-through stores and drain them before they report); what the tail writes is read by later launches.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kFrontLdsWords = 2 * kSmallCarve + 16;  // LDS of a k_front workgroup, whatever its role
constexpr uint32_t kTailPerThread = 8;
constexpr uint32_t kTailReqMax = kTailPerThread * 256;  // requests of a frame (all in registers at once)
constexpr uint32_t kTailWinMax = kFreshCap;             // winners (ordered in LDS)
constexpr uint32_t kTailBins = 512;                     // bins of the winners' counting sort by raster rank
// LDS words: [0, 32) counters | rank | x,y | z, entry slot | ranks sorted by bin | bin cursors, then pool indices
constexpr uint32_t kTailLdsWords = 32 + 5 * kTailWinMax;
static_assert(kTailBins <= kTailWinMax && kTailBins == 2 * 256, "two bins per thread; the cursors' memory is reused");

// Every wave of a directory workgroup of k_front calls this when its role is done; true (uniform) in the
// workgroup that reports last.  `wg`: its index among the n_wg directory workgroups.
__device__ inline bool front_arrive(FrameCtl* F, uint32_t wg, uint32_t n_wg, Ctl* ctl) {
  __shared__ uint32_t arrived_last;
#ifdef RATSDF_STAMPS
  __shared__ unsigned long long arrive_t[3];
  if (threadIdx.x == 0) arrive_t[0] = wall_clock64();
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores have left the CU ...
  __syncthreads();                                  // ... and every other wave's
#ifdef RATSDF_STAMPS
  if (threadIdx.x == 0) arrive_t[1] = wall_clock64();
#endif
  if (threadIdx.x == 0) {
    const uint32_t sub = wg % kArriveSubs;
    const uint32_t members = (n_wg - sub + kArriveSubs - 1) / kArriveSubs;  // workgroups b < n_wg, b % subs == sub
    uint32_t last = 0;
    if (__hip_atomic_fetch_add(&F->arrive[sub * kListStride], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
        members - 1u) {
      const uint32_t tops = n_wg < kArriveSubs ? n_wg : kArriveSubs;
      last = __hip_atomic_fetch_add(&F->arrive_top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == tops - 1u;
    }
    arrived_last = last;
#ifdef RATSDF_STAMPS
    arrive_t[2] = wall_clock64();
#endif
  }
  __syncthreads();
#ifdef RATSDF_STAMPS
  if (arrived_last && threadIdx.x == 0) {  // timeline of the last arriver, relative to the launch's first stamp
    const unsigned long long T0 = __hip_atomic_load(&ctl->tstamps[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int i = 0; i < 3; ++i) ctl->tstamps[1 + i] += arrive_t[i] - T0;
  }
#endif
  return arrived_last != 0u;
}

__device__ __forceinline__ void front_tail_role(const Table& tab, const FrameParams& P, const CandSet& cand,
                                                const Request* req, uint32_t req_cap, VisItem* vis,
                                                uint32_t seg_cap, const Pool& pool, const CarveBufs& cb, Ctl* ctl,
                                                uint32_t par, ratsdf_frame_stats* stats, uint32_t* lds) {
  static_assert(kTailLdsWords <= kFrontLdsWords, "the tail role works in the launch's role buffer");
  constexpr uint32_t NT = 256;
  constexpr int kU = (int)kTailPerThread;
  const uint32_t tid = threadIdx.x;
  FrameCtl* F = &ctl->fr[par];
  FrameCtl* Fp = &ctl->fr[par ^ 1u];
  uint32_t* win_rank = lds + 32;
  uint32_t* win_w0 = win_rank + kTailWinMax;
  uint32_t* win_ze = win_w0 + kTailWinMax;   // z | (entry & 1) << 16 (the entry's bucket is the block's hash)
  uint32_t* srt = win_ze + kTailWinMax;      // the winners' ranks, grouped by bin
  uint32_t* bins = srt + kTailWinMax;        // kTailBins cursors; later the pool indices the winners pop
  // lds: [0] head / chain deletes of the previous frame that happened, [1] winners, [2] voxels updated,
  //      [4 + l] new blocks filed in list l, [16, 20) wave totals of the scan
  // bin of a raster rank: rank >> sh, below kTailBins for every rank of the frame (pixel * S + sample)
  const uint32_t nranks = (uint32_t)(P.W * P.H) * (uint32_t)P.S;
  const uint32_t sh = nranks > kTailBins ? 32u - (uint32_t)__builtin_clz((nranks - 1u) >> 9) : 0u;

  // ---- one round of loads ----
  const int32_t nf0 = ctl->num_free;                 // (previous launches: plain)
  const uint32_t pend = Fp->pending;
  uint32_t nd = Fp->n_delcand, ns = Fp->n_slow_del;
  const uint32_t p_win = Fp->n_win, p_slow = Fp->n_slow;
  const uint32_t nv = frame_visible_blocks(Fp);
  const uint32_t n_slow = ld_agent(&F->n_slow);      // (this launch's consumers: past the caches)
  uint32_t n = ld_agent(&F->n_req);
  // the frame's requests (the slots exist whatever the count is): block, flags and rank -- the entry a
  // request fills is its home bucket's first or second one (kReqSlot1)
  RequestHead r[kU];
#pragma unroll
  for (int k = 0; k < kU; ++k) {
    const uint32_t i = tid + (uint32_t)k * NT;
    r[k] = ld_agent_request_head(req + (i < req_cap ? i : 0u));
  }
  auto bucket_of = [&](const RequestHead& q) {
    return block_hash((int16_t)(q.w0 & 0xFFFFu), (int16_t)(q.w0 >> 16), (int16_t)(q.w1 & 0xFFFFu), tab.bucket_mask);
  };
  uint4 u = make_uint4(0, 0, 0, 0);
  static_assert(kUpdCounters / NT == 4, "one uint4 of update counters per thread");
  if (pend) u = reinterpret_cast<const uint4*>(cb.upd_wg)[tid];
  if (tid < 32) lds[tid] = 0;
  bins[tid] = 0;
  bins[tid + NT] = 0;
  if (nd > cb.del_cap) nd = cb.del_cap;
  if (ns > cb.slow_cap) ns = cb.slow_cap;
  if (n > req_cap) n = req_cap;
  // uniform: not a frame for the tail -- nothing has been changed
  if (n_slow != 0 || n > kTailReqMax || (pend && nd + ns > kSmallCarve)) return;
#ifdef RATSDF_STAMPS
  unsigned long long tt[5];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  tt[0] = wall_clock64();
#endif
  lds_barrier();
  if (pend && ns) {  // rare: states written by carve_resolve_slow in workgroup 0 of this launch
    for (uint32_t j = tid; j < ns; j += NT)
      if ((ld_agent(reinterpret_cast<const uint32_t*>(&cb.slow[j]) + 1) >> 16) == 2u) atomicAdd(&lds[0], 1u);
  }
  // ---- second dependent round: the claim of every request's bucket ----
  uint32_t c[kU];
#pragma unroll
  for (int k = 0; k < kU; ++k) {
    const uint32_t i = tid + (uint32_t)k * NT;
    c[k] = kInf;
    if (i < n) c[k] = ld_agent(&tab.claim[bucket_of(r[k])]);
  }
  lds_barrier();
  const uint32_t n_del = pend ? nd + lds[0] : 0u;
  const int32_t nf = nf0 + (int32_t)n_del;
  // ... and, riding in the same round, the pool indices the winners will pop: heap[nf - 1 - k] (AquireBlock,
  // voxel_mem.cu:37-41; the release role of this launch may have pushed them moments ago: past the caches)
  int32_t hv[kTailWinMax / NT];
#pragma unroll
  for (uint32_t q = 0; q < kTailWinMax / NT; ++q) {
    const uint32_t j = tid + q * NT;
    hv[q] = -1;
    if ((int32_t)j < nf) hv[q] = (int32_t)ld_agent(reinterpret_cast<const uint32_t*>(&pool.heap[nf - 1 - (int32_t)j]));
  }
  // winners: first requester of a bucket in raster order (claim == own rank)
#pragma unroll
  for (int k = 0; k < kU; ++k) {
    const uint32_t i = tid + (uint32_t)k * NT;
    if (i < n && c[k] == r[k].rank) {
      const uint32_t slot = atomicAdd(&lds[1], 1u);
      if (slot < kTailWinMax) {
        win_rank[slot] = r[k].rank;
        win_w0[slot] = r[k].w0;
        win_ze[slot] = (r[k].w1 & 0xFFFFu) | (((r[k].w1 >> 16) & kReqSlot1) ? 1u << 16 : 0u);
        atomicAdd(&bins[(r[k].rank >> sh) & (kTailBins - 1u)], 1u);
      }
    }
  }
  lds_barrier();
#ifdef RATSDF_STAMPS
  tt[1] = wall_clock64();
#endif
  const uint32_t total = lds[1];
  if (total > kTailWinMax) return;  // uniform: a new view -- still nothing changed
  const uint32_t take = (int64_t)total > (int64_t)nf ? (uint32_t)(nf > 0 ? nf : 0) : total;  // voxel_mem.cu:39

  // ---- the winners' order (= order of the AquireBlock calls): a counting sort by bins of the raster rank.
  // (Until this was a count of the smaller ranks over the whole list per winner -- 270 winners: 74 k compares --
  // the tail spent 6.6 us of vector ALU time here, beside the look-ahead candidate workgroups the launch hosts.)
  {  // exclusive prefix of the bin counts, in place: two bins per thread
    const uint32_t a = bins[2u * tid], b = bins[2u * tid + 1u];
    uint32_t x = a + b;
    const uint32_t lane = tid & 63u, wave = tid >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(x, o);
      if (lane >= (uint32_t)o) x += t;
    }
    if (lane == 63u) lds[16 + wave] = x;
    lds_barrier();
    uint32_t before = 0;
#pragma unroll
    for (uint32_t w = 0; w < NT / 64; ++w)
      if (w < wave) before += lds[16 + w];
    const uint32_t excl = before + x - (a + b);
    bins[2u * tid] = excl;
    bins[2u * tid + 1u] = excl + a;
  }
  lds_barrier();
  constexpr uint32_t kSlots = kTailWinMax / NT;  // winners per thread
  uint32_t mine[kSlots], ord[kSlots];
#pragma unroll
  for (uint32_t q = 0; q < kSlots; ++q) {  // placement: a bin's cursor ends up at the bin's end
    const uint32_t w = tid + q * NT;
    mine[q] = w < total ? win_rank[w] : 0u;
    if (w < total) srt[atomicAdd(&bins[(mine[q] >> sh) & (kTailBins - 1u)], 1u)] = mine[q];
  }
  lds_barrier();
#pragma unroll
  for (uint32_t q = 0; q < kSlots; ++q) {  // winners of smaller rank = those in lower bins + the smaller ones of its own
    const uint32_t w = tid + q * NT;
    ord[q] = kInf;
    if (w < total) {
      const uint32_t bin = (mine[q] >> sh) & (kTailBins - 1u);
      const uint32_t lo = bin ? bins[bin - 1u] : 0u, hi = bins[bin];
      uint32_t k = lo;
      for (uint32_t j = lo; j < hi; ++j) k += srt[j] < mine[q];
      ord[q] = k;
    }
  }
  lds_barrier();  // (the cursors have been read: their memory takes the pool indices)
#pragma unroll
  for (uint32_t q = 0; q < kSlots; ++q) bins[tid + q * NT] = (uint32_t)hv[q];
  lds_barrier();
  // ---- commit: pool index, directory entry, occupancy bit, work-list item ----
#pragma unroll
  for (uint32_t q = 0; q < kSlots; ++q) {
    const uint32_t w = tid + q * NT;
    if (w < total && ord[q] < take) {
      const uint32_t w0 = win_w0[w], ze = win_ze[w], w1 = ze & 0xFFFFu;
      const int bx = (int16_t)(w0 & 0xFFFFu), by = (int16_t)(w0 >> 16), bz = (int16_t)w1;
      const uint32_t e = (block_hash(bx, by, bz, tab.bucket_mask) << 1) + (ze >> 16);
      const int32_t idx = (int32_t)bins[ord[q]];
      uint32_t* pe = reinterpret_cast<uint32_t*>(tab.entries + e);
      pe[0] = w0;                                                       // voxel_hash.cu:72-74
      pe[1] = w1;                                                       // offset 0
      pe[2] = (uint32_t)idx;
      atomicOr(&tab.occ[e >> 6], 1ull << (e & 63));
      mark_dirty(tab, e);
      reinterpret_cast<uint4*>(tab.active)[idx] = make_uint4(w0, w1, (uint32_t)idx, e);
      // Which of the 8 per-XCD lists: the image tile of the pixel that asked first (block_list_of projects the
      // block's centre -- ~100 vector instructions; placement only affects speed, any list is correct).
      // Approximate float arithmetic on purpose; the same on every run.
      const float fpix = floorf((float)mine[q] * (1.f / (float)P.S));
      const float fy = floorf(fpix * (1.f / (float)P.W)), fx = fpix - fy * (float)P.W;
      int tx = (int)(fx * (8.f / (float)P.W)), ty = (int)(fy * (8.f / (float)P.H));
      tx = tx < 0 ? 0 : (tx > 7 ? 7 : tx);
      ty = ty < 0 ? 0 : (ty > 7 ? 7 : ty);
      const uint32_t l = (uint32_t)(tx + 3 * ty) & 7u;
      const uint32_t pos = atomicAdd(&lds[4 + l], 1u);  // < kFreshCap: total <= kTailWinMax
      reinterpret_cast<uint4*>(vis)[(ptrdiff_t)((size_t)l * seg_cap + pos) - (ptrdiff_t)kFreshCap] =
          make_uint4(w0, w1, (uint32_t)idx, e);
    }
  }
  // ---- ResetLocks: every request's bucket (voxel_hash.cu:35-38) ----
#pragma unroll
  for (int k = 0; k < kU; ++k) {
    const uint32_t i = tid + (uint32_t)k * NT;
    if (i < n) tab.claim[bucket_of(r[k])] = kInf;
  }
#ifdef RATSDF_STAMPS
  tt[2] = wall_clock64();
#endif
  // ---- previous frame: voxels-updated sum ----
  if (pend) {
    uint32_t up = u.x + u.y + u.z + u.w;
    if (up) reinterpret_cast<uint4*>(cb.upd_wg)[tid] = make_uint4(0, 0, 0, 0);
    up = wave_sum(up);
    if ((tid & 63) == 0 && up) atomicAdd(&lds[2], up);
  }
  // the frame's candidate lists have been consumed (every consumer has reported): empty them for the frame
  // after next
  if (tid < (uint32_t)kCandSegs) cand.count[tid * kCandCountStride] = 0;
  lds_barrier();
  if (tid < (uint32_t)kNumLists) F->n_fresh[tid] = lds[4 + tid];
  if (tid == 0) {
    const uint32_t upd = lds[2];
    if (pend) {
      if (stats) {
        stats->visible_blocks = (int32_t)nv;
        stats->updated_voxels = (int32_t)upd;
        stats->allocated_blocks = (int32_t)p_win;
        stats->deleted_blocks = (int32_t)n_del;
        stats->active_blocks = tab.num_block - nf;
        stats->slow_requests = (int32_t)p_slow;
        atomicAdd(&ctl->totals[0], 1ull);
        atomicAdd(&ctl->totals[1], (unsigned long long)nv);
        atomicAdd(&ctl->totals[2], (unsigned long long)upd);
        atomicAdd(&ctl->totals[3], (unsigned long long)p_win);
        atomicAdd(&ctl->totals[4], (unsigned long long)n_del);
      }
      zero_frame_ctl(Fp, true);  // counters ready for the frame after next (every reader of this launch has reported)
    }
    if ((int64_t)total > (int64_t)nf) set_error(ctl, RATSDF_ERR_POOL_EXHAUSTED);
    F->alloc_base = (uint32_t)nf;
    F->n_win = take;
    F->n_winlist = total;
    F->pending = 1;  // this frame now owes a carve_finalize
    F->front_done = 1;
    ctl->num_free = nf - (int32_t)take;
    atomicMin(&ctl->free_low, nf - (int32_t)take);  // (Table::active: the slots ever in use; no reply awaited)
    atomicAdd(&ctl->paths[0], 1ull);
#ifdef RATSDF_STAMPS
    tt[3] = wall_clock64();
    const unsigned long long T0 = __hip_atomic_load(&ctl->tstamps[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int i = 0; i < 4; ++i) ctl->tstamps[4 + i] += tt[i] - T0;
    ctl->tstamps[8] += 1;
    ctl->tstamps[9] += n;
    ctl->tstamps[10] += total;
#endif
  }
}

// workgroups [0, n_vis_wg)               visible list of the blocks that exist before this frame
//                                        (longest dependency chain, so it is dispatched first)
// workgroups [.., +kCandSegs * parts)    allocation requests from the frame's candidate lists
// workgroups [.., +kReleaseWGs)          pool releases of the previous frame (carve_release_role)
// workgroups beyond                      look-ahead candidate pass of the next frame (`ahead()`)
// The directory workgroup that finishes last runs the frame's serial role (front_tail_role) when `tail`.
// kTail: the launch may run the frame's serial role at its tail (front_arrive / front_tail_role).  A template
// parameter, not a run-time switch: with the tail's code in the kernel -- even unused -- its scalar-register
// pressure spilled into the hosted candidate pass (864 instead of 140 v_readlane in the kernel; k_front 8.1 ->
// 9.0 us at 640x480, 22.8 -> 24.4 us at 1280x720, same-box A/B against the round-3 build).
template <bool kTail, typename Ahead>
__device__ inline void front_body(const Table& tab, const FrameParams& P, uint32_t n_vis_wg,
                                  const CandSet& cand, uint32_t cand_parts, Request* req,
                                  uint32_t req_cap, SlowRequest* slow, uint32_t slow_cap, VisItem* vis,
                                  uint32_t seg_cap, const Pool& pool, const CarveBufs& cb, Ctl* ctl,
                                  uint32_t par, ratsdf_frame_stats* stats, uint32_t tail, Ahead ahead,
                                  uint32_t* role_lds) {
  static_assert(sizeof(CandLds) <= kFrontLdsWords * 4 && kVisListCap <= 2 * kSmallCarve &&
                    sizeof(ReqBuf) <= kFrontLdsWords * 4,
                "role LDS");
  const uint32_t n_cons_wg = kCandSegs * cand_parts;
  const uint32_t n_dir_wg = n_vis_wg + n_cons_wg + kReleaseWGs;
  if (blockIdx.x >= n_dir_wg) {
    const CandJob job = ahead();
    cand_pixels_role(job, blockIdx.x - n_dir_wg, ctl, *reinterpret_cast<CandLds*>(role_lds));
    return;
  }
  FrameCtl* F = &ctl->fr[par];
  FrameCtl* Fp = &ctl->fr[par ^ 1u];
  // The launch's critical path runs in these workgroups -- chains of dependent round trips with a little
  // arithmetic in between -- while the look-ahead candidate workgroups it hosts keep the vector ALUs busy:
  // ask the instruction arbiter for priority over them.
  if (kTail && (tail & 2u)) __builtin_amdgcn_s_setprio(3);
#ifdef RATSDF_STAMPS
  if (blockIdx.x == 0 && threadIdx.x == 0)
    __hip_atomic_store(&ctl->tstamps[0], (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
  // (diagnostic build, RATSDF_DEBUG=20: workgroup 0 never publishes -- the waiters' bounded wait is what
  // tests/test_gpu_errors.py::test_in_launch_waits_are_bounded exercises)
  bool expired = false;  // uniform per workgroup
  auto gate = [&]() {
    const GateResult g = carve_resolve_gate(tab, cb, ctl, Fp, RATSDF_DBG(P, 20));
    expired = expired || g == kGateExpired;
    return g;
  };
  if (blockIdx.x >= n_vis_wg + n_cons_wg) {
    if (!RATSDF_DBG(P, 12) && gate() != kGateExpired)  // diagnostic ablations 3 / 11 / 12: skip one role
      carve_release_role(tab, pool, cb, ctl, Fp, blockIdx.x - n_vis_wg - n_cons_wg, role_lds);
  } else if (blockIdx.x >= n_vis_wg) {
    if (!RATSDF_DBG(P, 11)) {
      const uint32_t c = blockIdx.x - n_vis_wg;
      cand_consume_role(tab, P, cand, c % kCandSegs, c / kCandSegs, cand_parts, req, req_cap, slow,
                        slow_cap, ctl, F, gate, *reinterpret_cast<ReqBuf*>(role_lds));
    }
  } else {
    if (!RATSDF_DBG(P, 3)) visible_append_role(tab, P, blockIdx.x, n_vis_wg, vis, seg_cap, ctl, F, gate, role_lds);
  }
  // (a workgroup whose gate expired does not report: the directory may be half-edited, the tail must not
  // run on it -- the sticky error says the frame is incomplete)
  if constexpr (kTail) {
    if ((tail & 1u) && !expired && front_arrive(F, blockIdx.x, n_dir_wg, ctl))
      front_tail_role(tab, P, cand, req, req_cap, vis, seg_cap, pool, cb, ctl, par, stats, role_lds);
  }
}

template <bool kTail>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(8))) void k_front(
    Table tab, FrameParams P, uint32_t n_vis_wg, CandSet cand, uint32_t cand_parts, Request* req,
    uint32_t req_cap,
    SlowRequest* slow, uint32_t slow_cap, VisItem* vis, uint32_t seg_cap, Pool pool, CarveBufs cb,
    Ctl* ctl, uint32_t par, ratsdf_frame_stats* stats, uint32_t tail, CandJob ahead) {
  // one LDS buffer for whichever role the workgroup plays
  __shared__ __attribute__((aligned(16))) uint32_t role_lds[kFrontLdsWords];
  front_body<kTail>(tab, P, n_vis_wg, cand, cand_parts, req, req_cap, slow, slow_cap, vis, seg_cap, pool, cb,
                    ctl, par, stats, tail, [&]() { return ahead; }, role_lds);
}

// k_front for a frame nobody looked ahead for (kernels_cand.h: cand_inline_role): the frame's own candidate pass rides
// in the launch, every pixel workgroup its own consumer.
//   workgroups [0, n_vis_wg)            visible list
//   workgroups [.., +n_now_wg)          this frame's pixels -> candidates -> requests
//   workgroups [.., +kReleaseWGs)       pool releases of the previous frame
//   workgroups beyond                   look-ahead candidate pass of the next frame (`ahead`), if the caller named one
// A kernel of its own so that the steady-state k_front carries none of this (its scalar registers are all in use).
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(8))) void k_front_inline(
    Table tab, uint32_t n_vis_wg, uint32_t n_now_wg, Request* req, uint32_t req_cap, SlowRequest* slow,
    uint32_t slow_cap, VisItem* vis, uint32_t seg_cap, Pool pool, CarveBufs cb, Ctl* ctl, uint32_t par, CandJob now,
    CandJob ahead) {
  static_assert((kInlineReqOffsetWords * 4u + sizeof(ReqBuf)) <= kFrontLdsWords * 4u, "role LDS");
  __shared__ __attribute__((aligned(16))) uint32_t role_lds[kFrontLdsWords];
  const uint32_t n_dir_wg = n_vis_wg + n_now_wg + kReleaseWGs;
  if (blockIdx.x >= n_dir_wg) {
    cand_pixels_role(ahead, blockIdx.x - n_dir_wg, ctl, *reinterpret_cast<CandLds*>(role_lds));
    return;
  }
  FrameCtl* F = &ctl->fr[par];
  FrameCtl* Fp = &ctl->fr[par ^ 1u];
  auto gate = [&]() { return carve_resolve_gate(tab, cb, ctl, Fp, RATSDF_DBG(now.P, 20)); };
  if (blockIdx.x >= n_vis_wg + n_now_wg) {
    if (gate() != kGateExpired)
      carve_release_role(tab, pool, cb, ctl, Fp, blockIdx.x - n_vis_wg - n_now_wg, role_lds);
  } else if (blockIdx.x >= n_vis_wg) {
    cand_inline_role(now, blockIdx.x - n_vis_wg, tab, req, req_cap, slow, slow_cap, ctl, F, gate, role_lds);
  } else {
    visible_append_role(tab, now.P, blockIdx.x, n_vis_wg, vis, seg_cap, ctl, F, gate, role_lds);
  }
}

// the same launch for several engines: engine blockIdx.y, operands from its record and its job
template <bool kTail>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(8))) void k_front_g(
    EnginePtr engs, JobPtr cur, JobPtr nxt, uint32_t n_vis_wg, uint32_t cand_parts, uint32_t tail, AheadGeom ag) {
  __shared__ __attribute__((aligned(16))) uint32_t role_lds[kFrontLdsWords];
  EnginePtr E = engs + blockIdx.y;
  JobPtr J = cur + blockIdx.y;
  const uint32_t par = J->par;
  const RankBufs rb = ld_const(&E->rb);
  front_body<kTail>(ld_const(&E->tab), ld_const(&J->P), n_vis_wg, ld_const(&E->cand[par]), cand_parts, rb.req,
             rb.req_cap, E->slow, E->slow_cap, E->vis, E->seg_cap, ld_const(&E->pool),
             ld_const(&E->cb[par ^ 1u]), E->ctl, par, E->stats, tail,
             [&]() { return make_cand_job(E, nxt + blockIdx.y, ag); }, role_lds);
}

// The serial bookkeeping of a frame in its steady-state shape (few deletes, few requests, no chained
// buckets involved): carve_finalize of the previous frame + alloc_rank_role of this one, fused so that
// every input is requested in ONE round of loads at the top (a single workgroup that walks
// "counter -> list -> table -> ..." pays a full memory round trip per arrow, which is what these
// kernels consist of).  Anything unusual falls back to the general functions.
// `scratch`: the workgroup's dynamic LDS.  All 1024 threads.
__device__ inline void serial_frame_role(const Table& tab, const Pool& pool, const RankBufs& rb,
                                         const CarveBufs& cb, Ctl* ctl, uint32_t par,
                                         ratsdf_frame_stats* stats, unsigned long long* skeys) {
#ifdef RATSDF_STAMPS
  unsigned long long ts[6];
#define SSTAMP(i) ts[i] = (unsigned long long)clock64()
#else
#define SSTAMP(i) do { } while (0)
#endif
  constexpr uint32_t NT = 1024;
  static_assert(NT == kUpdCounters, "one update counter per thread");
  constexpr uint32_t RPT = kSmallRank / NT;   // requests per thread held in registers
  uint32_t* scratch = reinterpret_cast<uint32_t*>(skeys);
  uint32_t* lds = scratch;  // [32] slow-delete counter, [33] winner counter, [34] update sum
  const uint32_t tid = threadIdx.x;
  FrameCtl* Fp = &ctl->fr[par ^ 1u];
  FrameCtl* F = &ctl->fr[par];

  SSTAMP(0);
  // ---- one round of loads ----
  const int32_t nf0 = ctl->num_free;
  const uint32_t pend = Fp->pending;
  uint32_t nd = Fp->n_delcand, ns = Fp->n_slow_del;
  const uint32_t p_win = Fp->n_win, p_slow = Fp->n_slow;
  const uint32_t nv = frame_visible_blocks(Fp);
  const uint32_t n_slow = F->n_slow;
  uint32_t n = F->n_req;
  const uint32_t u = cb.upd_wg[tid & (kUpdCounters - 1)];  // NT == kUpdCounters
  Request r[RPT];  // the first NT requests ride in the first round; more only if there are more
  r[0] = rb.req[tid < rb.req_cap ? tid : 0];
  unsigned long long tot[5] = {0, 0, 0, 0, 0};
  if (tid == 0 && stats) {
#pragma unroll
    for (int i = 0; i < 5; ++i) tot[i] = ctl->totals[i];
  }
  if (nd > cb.del_cap) nd = cb.del_cap;
  if (ns > cb.slow_cap) ns = cb.slow_cap;
  if (n > rb.req_cap) n = rb.req_cap;

  const bool fast = (!pend || nd + ns <= kSmallCarve) && n_slow == 0 && n <= kSmallRank;
  SSTAMP(1);
  if (!fast) {  // uniform
    int32_t nf = nf0;
    nf += (int32_t)carve_finalize(tab, pool, cb, ctl, Fp, stats, nf, scratch);
    alloc_rank_role(tab, rb.req, rb.req_cap, rb.req_k, rb.slow, rb.slow_cap, rb.xlocks,
                    rb.bitmap, rb.summary, rb.prefix, rb.nwords, rb.sort_scratch, ctl, F, nf, skeys);
    return;
  }

  // second (and last) dependent round: the claim of every request's bucket
  uint32_t c[RPT];
#pragma unroll
  for (uint32_t k = 0; k < RPT; ++k) {
    const uint32_t i = tid + k * NT;
    c[k] = kInf;
    if (k > 0) {
      r[k] = Request{0, 0, 0, 0, 0, 0};
      if (i < n) r[k] = rb.req[i];
    }
    if (i < n) c[k] = tab.claim[block_hash(r[k].x, r[k].y, r[k].z, tab.bucket_mask)];
  }
  if (tid < 3) lds[32 + tid] = 0;
  lds_barrier();
  SSTAMP(2);
  // ---- previous frame: its pool releases were pushed by the release role of this frame's k_front
  // (same "few deletes" condition); here only the count and the voxels-updated sum are needed ----
  if (pend) {
    for (uint32_t j = tid; j < ns; j += NT)  // head / chain deletes: rare
      if (cb.slow[j].state == 2) atomicAdd(&lds[32], 1u);
    uint32_t up = u;
    if (u) cb.upd_wg[tid] = 0;
    up = wave_sum(up);
    if ((tid & 63) == 0 && up) atomicAdd(&lds[34], up);
  }
  // ---- this frame: the winners' ranks go to a compact list; k_integrate turns a rank into the
  // winner's position in raster order (= order of the AquireBlock calls) by counting the smaller ones
#pragma unroll
  for (uint32_t k = 0; k < RPT; ++k) {
    const uint32_t i = tid + k * NT;
    if (i < n && c[k] == r[k].rank) {
      rb.req[i].flags = kReqWinner;
      mark_dirty(tab, r[k].entry);  // (directory delta: the entry the commit will fill)
      const uint32_t slot = atomicAdd(&lds[33], 1u);
      rb.win_ranks[slot] = r[k].rank;
    }
  }
  lds_barrier();
  SSTAMP(3);
  const uint32_t n_del = pend ? nd + lds[32] : 0u;
  const uint32_t total = lds[33];
  const uint32_t upd = lds[34];
  const int32_t nf = nf0 + (int32_t)n_del;
  SSTAMP(4);
#ifdef RATSDF_STAMPS
  if (tid == 0) { ctl->stamps[16] += n_del; ctl->stamps[17] += total; ctl->stamps[18] += n; }
#endif
  if (tid == 0) {
    if (pend) {
      if (stats) {
        stats->visible_blocks = (int32_t)nv;
        stats->updated_voxels = (int32_t)upd;
        stats->allocated_blocks = (int32_t)p_win;
        stats->deleted_blocks = (int32_t)n_del;
        stats->active_blocks = tab.num_block - nf;
        stats->slow_requests = (int32_t)p_slow;
        ctl->totals[0] = tot[0] + 1;
        ctl->totals[1] = tot[1] + nv;
        ctl->totals[2] = tot[2] + upd;
        ctl->totals[3] = tot[3] + p_win;
        ctl->totals[4] = tot[4] + n_del;
      }
      zero_frame_ctl(Fp, tab.tail_on != 0);  // counters ready for the frame after next
    }
    uint32_t take = total;
    if ((int64_t)total > (int64_t)nf) {  // voxel_mem.cu:39 assert(idx >= 1)
      set_error(ctl, RATSDF_ERR_POOL_EXHAUSTED);
      take = (uint32_t)nf;
    }
    F->alloc_base = (uint32_t)nf;
    F->n_win = take;
    F->n_winlist = total;
    F->pending = 1;  // this frame now owes a carve_finalize
    ctl->num_free = nf - (int32_t)take;
    atomicMin(&ctl->free_low, nf - (int32_t)take);  // (Table::active: the slots ever in use; no reply awaited)
  }
  SSTAMP(5);
#ifdef RATSDF_STAMPS
  if (tid == 0)
    for (int i = 0; i < 6; ++i) atomicAdd(&ctl->stamps[8 + i], ts[i]);
#endif
}

// workgroup 0: carve_finalize of the previous frame, then this frame's allocation order
// workgroups beyond: look-ahead candidate pass of the next frame
__global__ __launch_bounds__(1024) void k_alloc_rank(Table tab, Pool pool, RankBufs rb, CarveBufs cb,
                                                     Ctl* ctl, uint32_t par,
                                                     ratsdf_frame_stats* stats, uint32_t* cand_count,
                                                     CandJob ahead) {
  if (blockIdx.x != 0) {
    __shared__ CandLds L;
    cand_pixels_role(ahead, blockIdx.x - 1, ctl, L);
    return;
  }
  extern __shared__ __attribute__((aligned(16))) unsigned long long skeys[];
  // the frame's candidate lists have been consumed by k_front: empty them for the frame after next
  if (cand_count && threadIdx.x < kCandSegs) cand_count[threadIdx.x * kCandCountStride] = 0;
  serial_frame_role(tab, pool, rb, cb, ctl, par, stats, skeys);
}

// several engines: the serial workgroups of all of them run side by side (one per blockIdx.y)
__global__ __launch_bounds__(1024) void k_alloc_rank_g(EnginePtr engs, JobPtr cur, JobPtr nxt,
                                                       AheadGeom ag) {
  EnginePtr E = engs + blockIdx.y;
  if (blockIdx.x != 0) {
    __shared__ CandLds L;
    const CandJob ahead = make_cand_job(E, nxt + blockIdx.y, ag);
    cand_pixels_role(ahead, blockIdx.x - 1, E->ctl, L);
    return;
  }
  extern __shared__ __attribute__((aligned(16))) unsigned long long skeys[];
  const uint32_t par = cur[blockIdx.y].par;
  uint32_t* cand_count = E->cand[par].count;
  if (threadIdx.x < kCandSegs) cand_count[threadIdx.x * kCandCountStride] = 0;
  RankBufs rb = ld_const(&E->rb);
  const FrameParams P = ld_const(&cur[blockIdx.y].P);
  rb.nwords = ((uint32_t)(P.W * P.H) * (uint32_t)P.S + 31u) / 32u;
  serial_frame_role(ld_const(&E->tab), ld_const(&E->pool), rb, ld_const(&E->cb[par ^ 1u]), E->ctl, par, E->stats,
                    skeys);
}

// first frame of a batch, several engines: its candidate pass as a launch of its own
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80))) void k_cand_g(EnginePtr engs,
                                                                                     JobPtr cur,
                                                                                     AheadGeom ag) {
  __shared__ CandLds L;
  EnginePtr E = engs + blockIdx.y;
  const CandJob job = make_cand_job(E, cur + blockIdx.y, ag);
  cand_pixels_role(job, blockIdx.x, E->ctl, L);
}

// voxel update of several engines (kernels_integrate.h: integrate_body)
template <int VPL, bool kTail>
__global__ __launch_bounds__(VPL == 1 ? 512 : RATSDF_INTEG_NT) __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(VPL <= 2 ? 8 : (VPL == 4 ? 5 : 3)))) void k_integrate_g(
    EnginePtr engs, JobPtr cur, JobPtr nxt, uint32_t n_int_wg, uint32_t n_serial_wg, uint32_t n_ahead_wg,
    uint32_t commit_rot, AheadGeom ag) {
  constexpr bool tail_on = kTail;
  __shared__ __attribute__((aligned(16))) uint32_t role_lds[kIntegLdsWords];
  EnginePtr E = engs + blockIdx.y;
  JobPtr J = cur + blockIdx.y;
  const uint32_t par = J->par;
  if (__builtin_expect(blockIdx.x >= n_serial_wg + n_ahead_wg, 1)) {  // role order as in k_integrate
    const uint32_t ibid = blockIdx.x - n_serial_wg - n_ahead_wg;
    const bool fused = n_serial_wg != 0;
    IntegArgs A;
    A.rgbw = E->pool.rgbw;
    A.tsdf = E->pool.tsdf;
    A.segm = E->pool.segm;
    A.texA = E->texA[par];
    A.texB = E->texB[par];
    A.vis = E->vis;
    A.seg_cap = E->seg_cap;
    A.F = &E->ctl->fr[par];
    A.upd_wg = E->cb[par].upd_wg;
    A.par = par;
    const FrameParams P = ld_const(&J->P);
#include "integrate_body.inc"
    return;
  }
  if (blockIdx.x >= n_serial_wg) {  // look-ahead workgroups
    if (VPL != 1) {
      const CandJob ahead = make_cand_job(E, nxt + blockIdx.y, ag);
      cand_pixels_role(ahead, blockIdx.x - n_serial_wg, E->ctl, *reinterpret_cast<CandLds*>(role_lds));
    }
    return;
  }
  if (blockIdx.x == 0) {  // the frame's serial role
    const uint32_t nwords = ((uint32_t)(J->P.W * J->P.H) * (uint32_t)J->P.S + 31u) / 32u;
    serial_workgroup(E, par, nwords, role_lds, RATSDF_DBG(J->P, 21));
  } else {
    serial_helper(E, par, role_lds, blockIdx.x, RATSDF_DBG(J->P, 22));
  }
}

}  // namespace ratsdf
