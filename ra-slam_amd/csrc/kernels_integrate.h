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

// Wave-wide sum by DPP (data-parallel primitives: the VALU reads a neighbouring lane's register
// directly): six steps, no LDS.  __shfl_xor compiles to ds_bpermute_b32 -- an LDS instruction plus four
// VALU instructions of index arithmetic per step, each step waiting for the LDS round trip of the one
// before it.  The result is valid in lane 63 and is broadcast from there through a scalar register.
// (The voxel update itself needs no such reduction any more: finish_block works with ballots.)
//   quad_perm [1,0,3,2], [2,3,0,1]: within 4 lanes; row_half_mirror, row_mirror: within a row of 16;
//   row_bcast:15 (rows 1, 3 take lane 15 of the row before), row_bcast:31 (rows 2, 3 take lane 31)
template <typename Op>
__device__ inline uint32_t wave_reduce_dpp(uint32_t v, uint32_t identity, Op op) {
  v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, 0xB1, 0xF, 0xF, false));
  v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, 0x4E, 0xF, 0xF, false));
  v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, 0x141, 0xF, 0xF, false));
  v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, 0x140, 0xF, 0xF, false));
  v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, 0x142, 0xA, 0xF, false));
  v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, 0x143, 0xC, 0xF, false));
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ inline uint32_t wave_sum(uint32_t v) {
  return wave_reduce_dpp(v, 0u, [](uint32_t a, uint32_t b) { return a + b; });
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

// ---------------------------------------------------------------------------------------------
// The frame's serial role INSIDE k_integrate (256 threads, workgroup 0 of the launch).
//
// As a launch of its own (k_alloc_rank, kernels_frame.h) the role costs the frame ~5 us in which one
// workgroup walks two dependent memory round trips and 255 CUs idle, plus a launch boundary.  Nothing
// the voxel update of the blocks that already exist needs comes out of it: only the commit of this
// frame's NEW blocks does (pool indices = order of the winners), and in frames that touch chained
// buckets the deletes of space carving (the resolver edits the directory).  So the role runs beside
// the update, publishes F->serial_done, and only those two consumers wait for it (bounded, like
// carve_resolve_gate).
// Hand-off WITHOUT cache maintenance in the ordinary frame: a release fence here would write back the
// whole L2 of this XCD (which the update workgroups next door keep filling with dirty voxel lines), an
// acquire in every committing workgroup would drop theirs.  Instead the few words the commit needs
// (winner flags, winners' ranks, alloc_base / n_win / n_winlist) are written with agent-scope stores
// (write-through), drained (vmcnt) before the flag goes out the same way, and read with agent-scope
// loads.  The general path stores plainly; it publishes serial_done = 2 after a real release and its
// consumers add an acquire.  What the previous frame still owes (carve_finalize) reads that frame's own
// lists and counters (per-parity buffers), never the ones this frame's update appends to.
// Same results as serial_frame_role / alloc_rank_role / carve_finalize: the fast path below is the
// 1024-thread one re-cut for 256 threads; everything unusual calls the general functions with their
// scratch in device memory instead of LDS (rare: first frames of a view, chained buckets).
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kFusedSlowCap = 1024;  // chained-bucket requests resolved with their keys in the role's LDS
constexpr uint32_t kFusedLockSlots = 2 * kFusedSlowCap;  // ... and their lock set: 512 distinct requests (a third
                                                         // of a pass's requests are, on the maps measured)
// LDS words of a k_integrate workgroup: the candidate pass's lists, or the resolver's keys and lock set
// behind the role's 8 counters (16 KiB: eight workgroups per CU use 128 of its 160 KiB)
constexpr uint32_t kFusedResolverWords = 8 + 2 * kFusedSlowCap + kFusedLockSlots;
constexpr uint32_t kIntegLdsWords =
    (sizeof(CandLds) + 3) / 4 > kFusedResolverWords ? (sizeof(CandLds) + 3) / 4 : kFusedResolverWords;

// A safety net, not a deadline: the role's general path with every capacity exhausted (16 384 chained
// requests sorted in device memory by 256 threads) takes tens of milliseconds.
constexpr unsigned long long kSerialWaitTicks = 200000000;  // 2 s of the 100 MHz wall clock

__device__ inline Request ld_agent_request(const Request* p) {
  const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
  unsigned long long w[2];
  w[0] = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  w[1] = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  Request r;
  __builtin_memcpy(&r, w, sizeof(r));
  return r;
}

// A wave waits until the frame's serial role has published (one lane polls; bounded, sticky error on
// expiry); after the general path (serial_done == 2) the caller's plain loads need an acquire.
// The polling backs off: in the ordinary frame the flag comes within ~2 us and the first polls are
// 0.25 us apart; in a frame that takes the role's general path (the first frames of a view: hundreds
// of microseconds) 8 192 waves polling one cache line at that cadence saturate its L2 channel and
// slow down the very role they are waiting for, so the gap doubles up to ~30 us.
__device__ inline uint32_t poll_serial_done(FrameCtl* F, Ctl* ctl) {
  uint32_t v = 0;
  const unsigned long long t0 = (unsigned long long)wall_clock64();
  uint32_t naps = 0;
  while ((v = ld_agent(&F->serial_done)) == 0u) {
    if ((unsigned long long)wall_clock64() - t0 > kSerialWaitTicks) {
      set_error(ctl, RATSDF_ERR_TIMEOUT);
      break;
    }
    if (naps < 16) {
      __builtin_amdgcn_s_sleep(8);
    } else {  // 4, 8, 16 ... 31 us (s_sleep 127 = 8 128 cycles)
      const uint32_t reps = naps < 24 ? 1u : (naps < 32 ? 2u : (naps < 40 ? 4u : 8u));
      for (uint32_t k = 0; k < reps; ++k) __builtin_amdgcn_s_sleep(127);
    }
    ++naps;
  }
  return v;
}
__device__ inline void wait_serial_done(FrameCtl* F, Ctl* ctl) {
  uint32_t v = 0;
  if ((threadIdx.x & 63u) == 0) v = poll_serial_done(F, ctl);
  v = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
  if (v != 1u) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // uniform
}
// The same for a whole workgroup at once (every wave of the workgroup calls it at the same point):
// wave 0 polls, the others wait at an LDS barrier -- a quarter of the pollers.  `lds`: one word.
__device__ inline void wait_serial_done_wg(FrameCtl* F, Ctl* ctl, uint32_t* lds) {
  if (threadIdx.x == 0) *lds = poll_serial_done(F, ctl);
  lds_barrier();
  const uint32_t v = *lds;
  if (v != 1u) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // uniform
  lds_barrier();  // (the word may be reused)
}

// returns 1 (ordinary frame: agent-scope stores only) or 2 (general path: plain stores)
// The general paths.  Their operands are read from the engine record where they are used (references
// into constant memory, not local copies): held in registers all along, the ~30 pointers pushed scalar
// spills into scratch memory, and a kernel that uses scratch pays for it in every wave it launches
// (-25% on the whole frame).
__device__ __forceinline__ void serial_general(EnginePtr E, uint32_t par, uint32_t nwords, int32_t nf,
                                               bool resolved) {
  Ctl* ctl = E->ctl;
  const Table& tab = *(const Table*)(&E->tab);
  const RankBufs& rb = *(const RankBufs*)(&E->rb);
  const CarveBufs& cb = *(const CarveBufs*)(&E->cb[par ^ 1u]);
  const Pool& pool = *(const Pool*)(&E->pool);
  uint32_t* scratch = E->serial_scratch;
  nf += (int32_t)carve_finalize(tab, pool, cb, ctl, &ctl->fr[par ^ 1u], E->stats, nf, scratch);
  alloc_rank_role(tab, rb.req, rb.req_cap, rb.req_k, rb.slow, rb.slow_cap, rb.xlocks,
                  rb.bitmap, rb.summary, rb.prefix, nwords, rb.sort_scratch, ctl, &ctl->fr[par], nf,
                  reinterpret_cast<unsigned long long*>(scratch), resolved);
}

// The pass over the frame's requests that decides the winners: request -> its bucket's claim -> (winner:
// flag in the request, rank into win_ranks).  Chunks of kClaimChunk requests, four per thread in flight;
// this workgroup takes chunks first, first + stride, ...  The list position of a winner comes from the
// workgroup's own LDS counter (`cursor` null: the serial workgroup alone, positions 0, 1, ...) or, when
// several workgroups share the pass, from a device-wide cursor bumped once per chunk.
// lds: [1] winners of this workgroup so far (caller zeroes), [4] / [5] scratch of the shared mode.
constexpr uint32_t kClaimPerThread = 4;
constexpr uint32_t kClaimChunk = kClaimPerThread * 256;
constexpr uint32_t kClaimPre = 3;  // requests per thread that ride in the serial role's first round of loads
                                   // (768 requests: a frame of the 640x480 stream files 700 - 1 000)
constexpr uint32_t kHelpMin = 4 * kClaimChunk;  // requests from which the seven neighbours help
constexpr uint32_t kSerialGroup = 8;            // workgroups of the serial group (the first is the role)
// (what the pass needs of a request: block, flags, rank -- its first three words)
struct RequestHead {
  uint32_t w0, w1, rank;  // x | y << 16, z | flags << 16, raster rank
};
__device__ inline RequestHead ld_agent_request_head(const Request* p) {
  const uint32_t* q = reinterpret_cast<const uint32_t*>(p);
  const unsigned long long a = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(q), __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
  return RequestHead{(uint32_t)a, (uint32_t)(a >> 32),
                     __hip_atomic_load(q + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)};
}
__device__ inline RequestHead ld_request_head(const Request* p) {
  const uint32_t* q = reinterpret_cast<const uint32_t*>(p);
  const uint2 a = *reinterpret_cast<const uint2*>(q);
  return RequestHead{a.x, a.y, q[2]};
}
__device__ __forceinline__ void claim_pass(const Table& tab, const RankBufs& rb, uint32_t n, uint32_t* lds,
                                           uint32_t first, uint32_t stride, uint32_t* cursor,
                                           const RequestHead (&pre)[kClaimPre]) {
  constexpr uint32_t NT = 256;
  constexpr int kU = (int)kClaimPerThread;
  const uint32_t tid = threadIdx.x;
  for (uint32_t base = first * kClaimChunk; base < n; base += stride * kClaimChunk) {  // uniform
    RequestHead r[kU];
    uint32_t c[kU];
#pragma unroll
    for (int k = 0; k < kU; ++k) {  // (agent scope: requests the resolver appended are read past this CU's L1)
      const uint32_t i = base + (uint32_t)k * NT + tid;
      r[k] = RequestHead{0, 0, 0};
      if (base == 0 && k < kClaimPre) r[k] = pre[k];  // (rode in the caller's first round of loads)
      else if (i < n) r[k] = ld_agent_request_head(rb.req + i);
    }
#pragma unroll
    for (int k = 0; k < kU; ++k) {
      const uint32_t i = base + (uint32_t)k * NT + tid;
      c[k] = kInf;
      if (i < n) {
        // (shared pass: past this XCD's L2 -- the resolver of the same launch, on workgroup 0's XCD, may have
        // reset this claim; its resets are write-through stores, drained before help_go went out)
        const uint32_t* pc = &tab.claim[block_hash((int16_t)(r[k].w0 & 0xFFFFu), (int16_t)(r[k].w0 >> 16),
                                                   (int16_t)(r[k].w1 & 0xFFFFu), tab.bucket_mask)];
        c[k] = cursor ? ld_agent(pc) : *pc;
      }
    }
    // the winners' ranks go to a compact list; the committing waves turn a rank into the winner's
    // position in raster order (= order of the AquireBlock calls) by counting the smaller ones.
    // A request the resolver placed is a winner as it stands (its bucket's claim is not its own).
    uint32_t at[kU];
    if (cursor) {  // uniform
      if (tid == 0) lds[4] = 0;
      lds_barrier();
    }
#pragma unroll
    for (int k = 0; k < kU; ++k) {
      const uint32_t i = base + (uint32_t)k * NT + tid;
      at[k] = kInf;
      if (i >= n) continue;
      const bool placed = ((r[k].w1 >> 16) & kReqPlaced) != 0;
      if (placed || c[k] == r[k].rank) {
        if (!placed) {  // word 1 = z | flags << 16: the winner flag as one agent-scope word store
          st_agent(reinterpret_cast<uint32_t*>(rb.req + i) + 1, (r[k].w1 & 0xFFFFu) | ((uint32_t)kReqWinner << 16));
          // (directory delta: the entry this winner's commit will fill -- its home bucket's first or second one)
          if (tab.delta_on)
            mark_dirty(tab, (block_hash((int16_t)(r[k].w0 & 0xFFFFu), (int16_t)(r[k].w0 >> 16),
                                        (int16_t)(r[k].w1 & 0xFFFFu), tab.bucket_mask) << 1) +
                                (((r[k].w1 >> 16) & kReqSlot1) ? 1u : 0u));
        }
        at[k] = atomicAdd(&lds[cursor ? 4 : 1], 1u);
      }
    }
    uint32_t off = 0;
    if (cursor) {
      lds_barrier();
      if (tid == 0) {
        const uint32_t cnt = lds[4];
        lds[5] = cnt ? __hip_atomic_fetch_add(cursor, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
      }
      lds_barrier();
      off = lds[5];
    }
#pragma unroll
    for (int k = 0; k < kU; ++k)
      if (at[k] != kInf) st_agent(&rb.win_ranks[off + at[k]], r[k].rank);
  }
}

__device__ __forceinline__ uint32_t serial_role256(EnginePtr E, uint32_t par, uint32_t nwords,
                                      uint32_t* lds /* >= 8 words */) {
  constexpr uint32_t NT = 256;
  constexpr uint32_t UPT = kUpdCounters / NT;  // update counters per thread
  const uint32_t tid = threadIdx.x;
  Ctl* ctl = E->ctl;
  FrameCtl* Fp = &ctl->fr[par ^ 1u];
  FrameCtl* F = &ctl->fr[par];
  const Table tab = ld_const(&E->tab);
  const RankBufs rb = ld_const(&E->rb);
  const CarveBufs cb = ld_const(&E->cb[par ^ 1u]);  // the PREVIOUS frame's lists and counters
  ratsdf_frame_stats* stats = E->stats;
  // the frame's candidate lists have been consumed by k_front: empty them for the frame after next
  if (tid < kCandSegs) E->cand[par].count[tid * kCandCountStride] = 0;

  // ---- one round of loads ----
  int32_t nf0;
  uint32_t pend, nd, ns, p_win, p_slow, nv, n;
  RequestHead pre[kClaimPre];  // the first requests (the slots exist whatever the count is)
  auto first_round = [&](auto after_resolver) {
    nf0 = ctl->num_free;
    pend = Fp->pending;
    nd = Fp->n_delcand;
    ns = Fp->n_slow_del;
    p_win = Fp->n_win;
    p_slow = Fp->n_slow;
    nv = frame_visible_blocks(Fp);
#pragma unroll
    for (uint32_t k = 0; k < kClaimPre; ++k) {  // the first requests ride in the first round
      const Request* q = rb.req + (tid + k * NT < rb.req_cap ? tid + k * NT : 0);
      if (decltype(after_resolver)::value) pre[k] = ld_agent_request_head(q);  // what it appended: past this CU's L1
      else pre[k] = ld_request_head(q);
    }
    n = decltype(after_resolver)::value ? ld_agent(&F->n_req) : F->n_req;
  };
  const uint32_t n_slow = F->n_slow;
  RATSDF_STAMP(ctl->stamps, 8);
  first_round(std::false_type{});
  // Chained-bucket requests (a map of tens of thousands of blocks has a few in most frames): the
  // resolver replays them in rank order against the claim table and appends what it places to the
  // request list as winners; with its sort keys in the workgroup's LDS it costs a few dependent loads
  // per request, and the rest of the frame is the ordinary one below.  (With its scratch in device
  // memory and the LDS ranking of alloc_rank_role behind it, this took ~300 us per frame at 1280x720 /
  // 2 mm on a 400 MB map: the general path is for frames that are unusual in size, not in kind.)
  if (__builtin_expect(n_slow != 0, 0)) {  // uniform; the resolver's one call site in this role
    // (ordinary: keys and lock set in this workgroup's LDS; past kFusedSlowCap requests: in device memory)
    unsigned long long* lds_keys = reinterpret_cast<unsigned long long*>(lds + 8);
    if (n_slow > kFusedSlowCap ||
        !resolve_slow_requests<true>(tab, rb.req, rb.req_cap, rb.slow, rb.slow_cap, rb.xlocks, ctl, F, lds_keys,
                                     rb.sort_scratch, kFusedSlowCap, lds + 8 + 2 * kFusedSlowCap, kFusedLockSlots))
      (void)resolve_slow_requests<false>(tab, rb.req, rb.req_cap, rb.slow, rb.slow_cap, rb.xlocks, ctl, F, lds_keys,
                                         rb.sort_scratch, 0u);
    __syncthreads();
    first_round(std::true_type{});  // again rather than held in registers across the resolver (what it
                                    // placed sits behind the frame's own requests)
  }
  if (nd > cb.del_cap) nd = cb.del_cap;
  if (ns > cb.slow_cap) ns = cb.slow_cap;
  if (n > rb.req_cap) n = rb.req_cap;

  const bool fast = (!pend || nd + ns <= kSmallCarve) && n <= kFusedRank;
  if (__builtin_expect(!fast, 0)) {  // uniform: the general functions, scratch in device memory
    if (tid == 0) {
      st_agent(&F->help_go, 2u);  // (the neighbours stay out)
      atomicAdd(&ctl->paths[3], 1ull);
    }
    serial_general(E, par, nwords, nf0, true);
    return 2u;
  }

  // second dependent round: the claim of every request's bucket (claim_pass).  Four requests per thread are
  // in flight at once (the heads of the first three rode in the first round): n requests cost this workgroup
  // ceil(n / 1024) times two dependent round trips -- or an eighth of that from 4 096 requests on, where the
  // seven workgroups dispatched beside it share the pass (a whole new view files 30 k).
  // (the previous frame's update counters ride in this round: nothing depends on them but a sum)
  uint4 u = make_uint4(0, 0, 0, 0);
  static_assert(UPT == 4, "one uint4 of update counters per thread");
  if (pend) u = reinterpret_cast<const uint4*>(cb.upd_wg)[tid];
  if (tid < 3) lds[tid] = 0;  // [0] slow deletes that happened, [1] winners, [2] voxels updated
  lds_barrier();
  RATSDF_STAMP(ctl->stamps, 9);
  // A frame with thousands of requests: the seven workgroups dispatched beside this one take seven eighths
  // of the pass (serial_helper); this workgroup tells them (help_go), takes its share, and waits for them.
  const bool helped = n > kHelpMin;  // uniform
  if (tid == 0) st_agent(&F->help_go, helped ? 1u : 2u);
  if (__builtin_expect(helped, 0)) {
    claim_pass(tab, rb, n, lds, 0, kSerialGroup, &F->help_winners, pre);
    if (tid == 0) {  // bounded wait for the helpers' "done and drained"
      const unsigned long long t0 = (unsigned long long)wall_clock64();
      while (ld_agent(&F->help_done) < kSerialGroup - 1) {
        if ((unsigned long long)wall_clock64() - t0 > kSerialWaitTicks) {
          set_error(ctl, RATSDF_ERR_TIMEOUT);
          break;
        }
        __builtin_amdgcn_s_sleep(8);
      }
      lds[1] = ld_agent(&F->help_winners);
    }
  } else {
    claim_pass(tab, rb, n, lds, 0, 1, nullptr, pre);
  }
  RATSDF_STAMP(ctl->stamps, 10);
  if (pend) {  // previous frame: count of its head / chain deletes, voxels-updated sum
    for (uint32_t j = tid; j < ns; j += NT)
      if (cb.slow[j].state == 2) atomicAdd(&lds[0], 1u);
    uint32_t up = u.x + u.y + u.z + u.w;
    if (up) reinterpret_cast<uint4*>(cb.upd_wg)[tid] = make_uint4(0, 0, 0, 0);
    up = wave_sum(up);
    if ((tid & 63) == 0 && up) atomicAdd(&lds[2], up);
  }
  lds_barrier();
  const uint32_t n_del = pend ? nd + lds[0] : 0u;
  const uint32_t total = lds[1];
  const uint32_t upd = lds[2];
  const int32_t nf = nf0 + (int32_t)n_del;
  if (tid == 0) {
    if (pend) {
      if (stats) {
        stats->visible_blocks = (int32_t)nv;
        stats->updated_voxels = (int32_t)upd;
        stats->allocated_blocks = (int32_t)p_win;
        stats->deleted_blocks = (int32_t)n_del;
        stats->active_blocks = tab.num_block - nf;
        stats->slow_requests = (int32_t)p_slow;
        // fire-and-forget adds: nobody else touches the totals while a frame is in flight
        atomicAdd(&ctl->totals[0], 1ull);
        atomicAdd(&ctl->totals[1], (unsigned long long)nv);
        atomicAdd(&ctl->totals[2], (unsigned long long)upd);
        atomicAdd(&ctl->totals[3], (unsigned long long)p_win);
        atomicAdd(&ctl->totals[4], (unsigned long long)n_del);
      }
      zero_frame_ctl(Fp, tab.tail_on != 0);  // counters ready for the frame after next
    }
    uint32_t take = total;
    if ((int64_t)total > (int64_t)nf) {  // voxel_mem.cu:39 assert(idx >= 1)
      set_error(ctl, RATSDF_ERR_POOL_EXHAUSTED);
      take = (uint32_t)nf;
    }
    st_agent(&F->alloc_base, (uint32_t)nf);
    st_agent(&F->n_win, take);
    st_agent(&F->n_winlist, total);
    F->pending = 1;  // this frame now owes a carve_finalize
    ctl->num_free = nf - (int32_t)take;
    atomicMin(&ctl->free_low, nf - (int32_t)take);  // (Table::active: the slots ever in use; no reply awaited)
    atomicAdd(&ctl->paths[n_slow ? 2 : 1], 1ull);
  }
  RATSDF_STAMP(ctl->stamps, 11);
  RATSDF_STAMP(ctl->stamps, 12);
  RATSDF_STAMP(ctl->stamps, 13);
  return 1u;
}

// workgroup 0 of a fused launch: the role, then the hand-off (every storing wave drained, barrier,
// one agent-scope release, the flag)
// (`withhold`: diagnostic build only, RATSDF_DEBUG=21 -- the flag never goes out, the consumers' bounded
// waits expire: tests/test_gpu_errors.py::test_in_launch_waits_are_bounded)
__device__ __forceinline__ void serial_workgroup(EnginePtr E, uint32_t par, uint32_t nwords, uint32_t* lds,
                                                 bool withhold = false) {
  // (an ordinary frame: the role ran at the tail of k_front, kernels_frame.h: front_tail_role)
  if (E->tab.tail_on && E->ctl->fr[par].front_done) return;  // uniform
  // The frame's critical path runs in these four waves, each sharing its SIMD with seven waves of the voxel
  // update: ask the instruction arbiter for the highest wave priority.
  __builtin_amdgcn_s_setprio(3);
#ifdef RATSDF_STAMPS
  if (threadIdx.x == 0) E->ctl->stamps[par * 3u] = (unsigned long long)wall_clock64();
#endif
  const uint32_t how = serial_role256(E, par, nwords, lds);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    if (how != 1u) {  // uniform
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (!withhold) st_agent(&E->ctl->fr[par].serial_done, how);
#ifdef RATSDF_STAMPS
    E->ctl->stamps[par * 3u + 1u] = (unsigned long long)wall_clock64();  // (timeline: ratsdf_debug_wave_stamps)
#endif
  }
}

// Workgroups 1 .. 7 of the serial group: wait for the serial workgroup's word; when it asks, take every
// eighth chunk of the pass over the requests; report when every store has drained.  (`withhold`:
// diagnostic build only, RATSDF_DEBUG=22 -- never report, the serial workgroup's bounded wait expires.)
__device__ __forceinline__ void serial_helper(EnginePtr E, uint32_t par, uint32_t* lds, uint32_t wg,
                                              bool withhold = false) {
  Ctl* ctl = E->ctl;
  FrameCtl* F = &ctl->fr[par];
  if (E->tab.tail_on && F->front_done) return;  // uniform: no serial role in this launch (front_tail_role did the frame's)
  if (threadIdx.x == 0) {
    uint32_t v = 0;
    const unsigned long long t0 = (unsigned long long)wall_clock64();
    while ((v = ld_agent(&F->help_go)) == 0u) {
      if ((unsigned long long)wall_clock64() - t0 > kSerialWaitTicks) break;  // (the role reports its own expiry)
      __builtin_amdgcn_s_sleep(8);
    }
    lds[0] = v;
    lds[1] = 0;
  }
  lds_barrier();
  if (lds[0] != 1u) return;  // uniform: an ordinary frame
  const Table tab = ld_const(&E->tab);
  const RankBufs rb = ld_const(&E->rb);
  uint32_t n = ld_agent(&F->n_req);  // (agent scope, here and for the requests: the chained-bucket resolver
  if (n > rb.req_cap) n = rb.req_cap;  //  of this launch may have appended some)
  const RequestHead none[kClaimPre] = {};  // (chunk 0 is the serial workgroup's)
  claim_pass(tab, rb, n, lds, wg, kSerialGroup, &F->help_winners, none);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0 && !withhold)
    (void)__hip_atomic_fetch_add(&F->help_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}


// round-half-away-from-zero of a NON-NEGATIVE float (roundf for x >= 0, NaN stays NaN); the generic
// roundf additionally restores the sign (v_bfi)
__device__ inline float round_nonneg(float x) {
  const float t = truncf(x);
  return t + ((x - t) >= .5f ? 1.f : 0.f);
}

// Update of one voxel block by the waves that own it (WPB waves, `part` = which one).  Called with the
// whole wave (all 64 lanes) or not at all.  Returns the wave's summary word (the same in every lane):
// voxels updated | (a voxel with |tsdf| < 0.9 exists) << 16 | (a voxel that is not a NaN exists) << 17.
#ifdef RATSDF_STAMPS
#define WSTAMP(i) do { if (wstamps && (threadIdx.x & 63) == 0) wstamps[i] = (unsigned long long)clock64(); } while (0)
#else
#define WSTAMP(i) do { } while (0)
#endif
template <int VPL>
__device__ inline void integrate_block(const Pool& pool, const FrameParams& P, const VisItem& item,
                                       bool fresh_in, uint32_t vi0, const float4* texA,
                                       const uint32_t* texB, uint32_t* out_word,
                                       unsigned long long* wstamps = nullptr) {
  bool fresh = fresh_in;
  const int tx0 = vi0 & 7, ty = (vi0 >> 3) & 7, tz = vi0 >> 6;
  const size_t v = ((size_t)item.idx << 9) + vi0;
  uint32_t tv[VPL], sv[VPL], cv[VPL];
  if (RATSDF_DBG(P, 5)) fresh = true;  // diagnostic: no voxel loads / stores
  if (!RATSDF_DBG(P, 5)) VecIO<VPL>::load(pool.rgbw + v, cv);
  else
    for (int j = 0; j < VPL; ++j) cv[j] = 0;
  // (FrameParams::segm_live == 0: a map that has never seen ht / lt holds 0.5 in every voxel and a TSDF-only frame
  // leaves it there, so the probability of existing blocks is neither loaded nor stored)
  const bool segm_live = P.segm_live != 0;  // uniform
  if (!fresh) {
    VecIO<VPL>::load(reinterpret_cast<const uint32_t*>(pool.tsdf + v), tv);
    if (segm_live) {
      VecIO<VPL>::load(reinterpret_cast<const uint32_t*>(pool.segm + v), sv);
    } else {
      // (whatever the registers hold: the probability computed from it is never stored.  An empty asm "defines"
      // them without an instruction -- loading the constant 0.5 here cost the kernel registers it does not have)
#pragma unroll
      for (int j = 0; j < VPL; ++j) asm volatile("" : "=v"(sv[j]));
    }
  }
  const int gy = (int16_t)((int16_t)(item.y << 3) + ty);
  const int gz = (int16_t)((int16_t)(item.z << 3) + tz);
  const float wy = (float)gy * P.vs, wz = (float)gz * P.vs;
  float phz[VPL];
  uint32_t kk[VPL];
  bool inb[VPL];
  V3 ph[VPL];
  bool fast = true;
  if (VPL == 2 && !RATSDF_DBG(P, 14)) {
    // The lane's two voxels differ in x only: quat_rotate / se3_apply / intr_mul (device_math.h)
    // written out on 2-wide vectors, operation for operation (same association, no contraction), so
    // that the pair goes through packed FP32 instructions without the shuffles of automatic
    // vectorisation.
    typedef float v2f __attribute__((ext_vector_type(2)));
    const Quat q = P.T.q;
    const int gx0 = (int16_t)((int16_t)(item.x << 3) + tx0), gx1 = (int16_t)((int16_t)(item.x << 3) + tx0 + 1);
    const v2f vx = v2f{(float)gx0, (float)gx1} * P.vs;                  // :183-187
    // uv = 2 * cross(q.xyz, v)
    float uvx = q.y * wz - q.z * wy;
    v2f uvy = q.z * vx - q.x * wz;
    v2f uvz = q.x * wy - q.y * vx;
    uvx += uvx;
    uvy += uvy;
    uvz += uvz;
    // c = cross(q.xyz, uv)
    const v2f cx = q.y * uvz - q.z * uvy;
    const v2f cy = q.z * uvx - q.x * uvz;
    const v2f cz = q.x * uvy - q.y * uvx;
    const v2f rx = (vx + q.w * uvx) + cx;
    const v2f ry = (wy + q.w * uvy) + cy;
    const v2f rz = (wz + q.w * uvz) + cz;
    const v2f pcx = rx + P.T.t.x, pcy = ry + P.T.t.y, pcz = rz + P.T.t.z;  // :190
    const v2f phx = P.K.fx * pcx + P.K.cx * pcz;                        // :193
    const v2f phy = P.K.fy * pcy + P.K.cy * pcz;
    fast = recip_safe(pcz[0]) && recip_safe(pcz[1]);
    v2f qu, qv;
    if (fast) {  // hnormalized(): see the loop below; make_recip + div_shared on the pair
      auto fma2 = [](v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); };
      const v2f r0 = {__builtin_amdgcn_rcpf(pcz[0]), __builtin_amdgcn_rcpf(pcz[1])};
      const v2f r1 = fma2(fma2(-pcz, r0, v2f{1.f, 1.f}), r0, r0);
      auto div2 = [&](v2f a) {
        const v2f q0 = a * r1;
        const v2f q1 = fma2(fma2(-pcz, q0, a), r1, q0);
        return fma2(fma2(-pcz, q1, a), r1, q1);
      };
      qu = div2(phx);
      qv = div2(phy);
    } else {
      // a depth outside the shared-reciprocal range, e.g. a voxel in the camera plane (z == 0): plain
      // IEEE quotients; 0 / 0 = NaN picks pixel 0 like the reference's float -> int conversion, which
      // the short pixel pick below gets from a zero quotient (its conversion of a NaN is not 0)
      auto q = [](float a, float z) {
        const float r = a / z;
        return r == r ? r : 0.f;
      };
      qu = v2f{q(phx[0], pcz[0]), q(phx[1], pcz[1])};
      qv = v2f{q(phy[0], pcz[0]), q(phy[1], pcz[1])};
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      // u = (int)roundf(qu), 0 <= u < W (:196-205), in its short form: |u| = v_cvt_rpi_i32_f32(|qu|)
      // (floor(|x| + 0.5) evaluated exactly = |roundf(x)| for every float, NaN -> 0 and saturation as
      // the conversion of roundf(x) gives them), and a negative coordinate lies in the image only if it
      // rounds to 0, i.e. unless qu <= -0.5.  Equal to the long form for every one of the 2^32 floats
      // (tools/probes/round_probe.hip, k_pick) at a third of its instructions.
      const uint32_t u = rpi_abs(qu[j]);
      const uint32_t w = rpi_abs(qv[j]);
      inb[j] = u < (uint32_t)P.W && w < (uint32_t)P.H && !(qu[j] <= -.5f) && !(qv[j] <= -.5f);
      uint32_t k = w * (uint32_t)P.W + u;
      if (RATSDF_DBG(P, 15)) k = (k >> 6) << 6;  // diagnostic: 64-texel granularity (few cache lines per gather)
      kk[j] = inb[j] ? k : 0u;
      phz[j] = pcz[j];
    }
  } else {
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
      const int gx = (int16_t)((int16_t)(item.x << 3) + tx0 + j);         // :183-184
      const V3 pw{(float)gx * P.vs, wy, wz};                              // :187
      const V3 pc3 = se3_apply(P.T, pw);                                  // :190
      ph[j] = intr_mul(P.K, pc3);                                         // :193
      fast = fast && recip_safe(ph[j].z);
    }
  }
#pragma unroll
  for (int j = 0; j < (VPL == 2 && !RATSDF_DBG(P, 14) ? 0 : VPL); ++j) {
    // hnormalized(): two quotients with the same divisor (shared-divisor form, device_math.h: the
    // correctly rounded quotient for |z| in (1e-18, 1e18) and |x / z| below the overflow threshold,
    // i.e. for every pose with coordinates below ~1e18 m); the plain IEEE divisions for a depth
    // outside that range -- z == 0 happens: a voxel in the camera plane.  One test for the lane's
    // voxels together.
    float qu, qv;
    if (fast) {
      const Recip rz = make_recip(ph[j].z);
      qu = div_shared(ph[j].x, rz);
      qv = div_shared(ph[j].y, rz);
    } else {
      qu = ph[j].x / ph[j].z;
      qv = ph[j].y / ph[j].z;
    }
    const int u = f2i(roundf(qu));                                      // :196-199
    const int w = f2i(roundf(qv));                                      // :202
    // 0 <= u < W && 0 <= w < H (:205) as two unsigned compares; no branch around the index
    inb[j] = (uint32_t)u < (uint32_t)P.W && (uint32_t)w < (uint32_t)P.H;
    const uint32_t k = (uint32_t)w * (uint32_t)P.W + (uint32_t)u;
    kk[j] = inb[j] ? k : 0u;
    phz[j] = ph[j].z;
  }
  WSTAMP(1);
  float4 ta[VPL];
  uint32_t tb[VPL];
#pragma unroll
  for (int j = 0; j < VPL; ++j) {  // all gathers in flight together
#ifdef RATSDF_STAMPS
    if (RATSDF_DBG(P, 4)) {        // diagnostic: no gathers
      ta[j] = make_float4(2.f, 1.f, -0.5f, 2.f);
      tb[j] = 0x00808080u;
      continue;
    }
#endif
    ta[j] = texA[kk[j]];           // depth, range, log ht - log lt, w_new
    tb[j] = texB[kk[j]];           // rgb
  }
#ifdef RATSDF_STAMPS
  if (wstamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif
  WSTAMP(2);
  if (fresh) {  // AquireBlock initial values, voxel_mem.cu:43-51 (rgb stays as found)
    // (materialised here: hoisted out of the block loop the two constants took registers the update
    // needs, and were spilled to scratch memory instead of being re-created)
    uint32_t minus_one = __float_as_uint(-1.f), half = __float_as_uint(.5f);
    asm volatile("" : "+v"(minus_one), "+v"(half));
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
      tv[j] = minus_one;
      sv[j] = half;
      cv[j] = (cv[j] & 0x00FFFFFFu) | 0x01000000u;
    }
  }
  uint32_t nupd = 0;
  bool upd2[2] = {false, false};  // VPL == 2: which of the lane's two voxels updated
  const Recip rtrunc = make_recip(P.trunc);
  if (VPL == 2 && !RATSDF_DBG(P, 14)) {
    // The lane's two voxels side by side in 2-wide vectors: the same operations in the same order as
    // the loop below (the results are bit-identical), but as packed FP32 instructions (v_pk_mul /
    // v_pk_add / v_pk_fma_f32: two lanes of arithmetic per issue slot), which halves the instruction
    // count of the blend -- the kernel is bound by VALU issue, not by bytes.  Both voxels are computed
    // whenever either updates; what does not update is not stored.
    typedef float v2f __attribute__((ext_vector_type(2)));
    auto fma2 = [](v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); };
    auto div2 = [&](v2f a, v2f d, v2f r1) {  // div_shared (device_math.h), two quotients at once
      const v2f q0 = a * r1;
      const v2f e2 = fma2(-d, q0, a);
      const v2f q1 = fma2(e2, r1, q0);
      const v2f e3 = fma2(-d, q1, a);
      return fma2(e3, r1, q1);
    };
    const v2f d = {ta[0].x, ta[1].x};
    const v2f sdf = v2f{ta[0].y, ta[1].y} * (d - v2f{phz[0], phz[1]});            // :216
    bool (&upd)[2] = upd2;
#pragma unroll
    for (int j = 0; j < 2; ++j)  // (colour-word test: see the loop below)
      upd[j] = inb[j] && !(d[j] == 0 || d[j] > P.md) && sdf[j] > -P.trunc && tb[j] != 0xFFFFFFFFu;  // :211,217
    if (upd[0] || upd[1]) {
      v2f ts = div2(sdf, v2f{rtrunc.d, rtrunc.d}, v2f{rtrunc.r1, rtrunc.r1});    // :218
      ts = v2f{fminf(1, ts[0]), fminf(1, ts[1])};
      const v2f wn = {ta[0].w, ta[1].w};                                          // :226 (per pixel)
      const uint32_t c0 = cv[0], c1 = cv[1];
      const v2f wo = {(float)(c0 >> 24), (float)(c1 >> 24)};                      // :227
      const v2f wc = wo + wn;                                                     // :228
      const uint32_t n0 = tb[0], n1 = tb[1];
      // make_recip(wc)
      const v2f r0 = {__builtin_amdgcn_rcpf(wc[0]), __builtin_amdgcn_rcpf(wc[1])};
      const v2f e = fma2(-wc, r0, v2f{1.f, 1.f});
      const v2f r1 = fma2(e, r0, r0);
      // one colour channel at a time, packed into the output words at once (short live ranges: the
      // kernel has 64 VGPRs).  roundf of a non-negative float = v_cvt_rpi_i32_f32 (floor(x + 0.5),
      // evaluated exactly by the hardware: equal to roundf for every non-negative float, checked
      // exhaustively by tools/probes/round_probe.hip); quotients and the weight are non-negative and
      // at most 255 / 44, never NaN (wc >= 1).
      auto rpi = [](float x) {
        int r;
        asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
        return (uint32_t)r;
      };
      uint32_t o0, o1;
      {
        const v2f r_old = {(float)(c0 & 0xFFu), (float)(c1 & 0xFFu)};
        const v2f r_new = {(float)(n0 & 0xFFu), (float)(n1 & 0xFFu)};
        const v2f q = div2(r_old * wo + r_new * wn, wc, r1);                       // :234-235
        o0 = rpi(q[0]);                                                            // :239-240
        o1 = rpi(q[1]);
      }
      {
        const v2f g_old = {(float)((c0 >> 8) & 0xFFu), (float)((c1 >> 8) & 0xFFu)};
        const v2f g_new = {(float)((n0 >> 8) & 0xFFu), (float)((n1 >> 8) & 0xFFu)};
        const v2f q = div2(g_old * wo + g_new * wn, wc, r1);
        o0 |= rpi(q[0]) << 8;
        o1 |= rpi(q[1]) << 8;
      }
      {
        const v2f b_old = {(float)((c0 >> 16) & 0xFFu), (float)((c1 >> 16) & 0xFFu)};
        const v2f b_new = {(float)((n0 >> 16) & 0xFFu), (float)((n1 >> 16) & 0xFFu)};
        const v2f q = div2(b_old * wo + b_new * wn, wc, r1);
        o0 |= rpi(q[0]) << 16;
        o1 |= rpi(q[1]) << 16;
      }
      {
        const uint32_t w0 = rpi(wc[0]), w1 = rpi(wc[1]);                           // :238
        o0 |= (w0 < 40u ? w0 : 40u) << 24;
        o1 |= (w1 < 40u ? w1 : 40u) << 24;
      }
      const v2f t_old = {__uint_as_float(tv[0]), __uint_as_float(tv[1])};
      const v2f t_new = div2(t_old * wo + ts * wn, wc, r1);                        // :236
      const uint32_t ow[2] = {o0, o1};
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (!upd[j]) continue;
        tv[j] = __float_as_uint(t_new[j]);
        cv[j] = ow[j];
        // probability: see the loop below.  (P.segm_live == 0: not loaded, not stored, and what is computed here from
        // an unloaded register is dropped -- a uniform branch around these twelve instructions cost the kernel two
        // vector registers it does not have, i.e. scratch)
        const float pr = __uint_as_float(sv[j]);
        const float odds = pr * __builtin_amdgcn_rcpf(1.f - pr);
        const float L = __builtin_amdgcn_logf(odds) * 0.69314718f;
        const float x = (wo[j] * L + wn[j] * ta[j].z) * r1[j];
        const float ex = __builtin_amdgcn_exp2f(x * -1.44269504f);
        sv[j] = __float_as_uint(__builtin_amdgcn_rcpf(1.f + ex));
      }
    }
    nupd = upd[0] || upd[1];  // (non-zero = something to store)
  } else {
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    const float d = ta[j].x;
    const float sdf = ta[j].y * (d - phz[j]);                           // :216
    // (the colour word never has its top byte set; testing it here keeps the compiler from sinking
    // that gather into the branch, where it would be a second, dependent memory round trip)
    if (inb[j] && !(d == 0 || d > P.md) && sdf > -P.trunc && tb[j] != 0xFFFFFFFFu) {  // :211,217
      // sdf in (-trunc, range * max_depth]: the shared-divisor quotient is the IEEE one
      const float ts = fminf(1, div_shared(sdf, rtrunc));               // :218
      const float wn = ta[j].w;                                         // :226 (per pixel)
      const uint32_t c = cv[j];
      const float wo = (float)(c >> 24);                                // :227
      const float wc = wo + wn;                                         // :228
      const float r_old = (float)(c & 0xFFu), g_old = (float)((c >> 8) & 0xFFu),
                  b_old = (float)((c >> 16) & 0xFFu);
      const uint32_t cn = tb[j];
      const float r_new = (float)(cn & 0xFFu), g_new = (float)((cn >> 8) & 0xFFu),
                  b_new = (float)((cn >> 16) & 0xFFu);
      // wc = weight (1..40) + w_new (0..4): the quotients of :234-240 share it.  Numerators are
      // bounded (bytes times weights, |tsdf| <= 1), so no operand leaves the range in which the
      // shared-divisor form is the correctly rounded quotient; NaN propagates as in IEEE division.
      const Recip rwc = make_recip(wc);
      const float rc = div_shared(r_old * wo + r_new * wn, rwc);        // :234-235
      const float gc = div_shared(g_old * wo + g_new * wn, rwc);
      const float bc = div_shared(b_old * wo + b_new * wn, rwc);
      const float t_old = __uint_as_float(tv[j]);
      tv[j] = __float_as_uint(div_shared(t_old * wo + ts * wn, rwc));   // :236
      // quotients and the weight are non-negative: roundf without the sign step, and the byte pack
      // of an integer-valued float is exact in any rounding mode
      uint32_t o = 0;
      o = __builtin_amdgcn_cvt_pk_u8_f32(round_nonneg(rc), 0, o);        // :239-240
      o = __builtin_amdgcn_cvt_pk_u8_f32(round_nonneg(gc), 1, o);
      o = __builtin_amdgcn_cvt_pk_u8_f32(round_nonneg(bc), 2, o);
      o = __builtin_amdgcn_cvt_pk_u8_f32(fminf(round_nonneg(wc), 40), 3, o);  // :238
      cv[j] = o;
      if (!RATSDF_DBG(P, 6)) {  // diagnostic 6: no transcendental part
        // probability (:242-248): p' = pos / (pos + neg) with pos = exp((wo log p + wn log ht) / wc),
        // neg = exp((wo log(1-p) + wn log lt) / wc)  ==  1 / (1 + exp(-(wo L + wn Ln) / wc)) with the
        // log-odds L = log(p / (1-p)) and the per-pixel Ln = log ht - log lt (texel).  Same function,
        // one log and one exp instead of two each; hardware v_log / v_exp / v_rcp (~1e-7 relative;
        // tolerance on the probability is 1e-4, and it feeds nothing else).  0, 1, inf and NaN
        // behave as in the reference's form (ht = 0 -> p' = 0; ht = lt = 0 -> NaN; ...).
        const float pr = __uint_as_float(sv[j]);
        const float odds = pr * __builtin_amdgcn_rcpf(1.f - pr);
        const float L = __builtin_amdgcn_logf(odds) * 0.69314718f;
        const float x = (wo * L + wn * ta[j].z) * rwc.r1;
        const float e = __builtin_amdgcn_exp2f(x * -1.44269504f);
        sv[j] = __float_as_uint(__builtin_amdgcn_rcpf(1.f + e));
      }
      ++nupd;
    }
  }
  }
  WSTAMP(3);
  if ((nupd || fresh) && !RATSDF_DBG(P, 5) && !RATSDF_DBG(P, 7)) {
    VecIO<VPL>::store(reinterpret_cast<uint32_t*>(pool.tsdf + v), tv);
    if (segm_live || fresh) VecIO<VPL>::store(reinterpret_cast<uint32_t*>(pool.segm + v), sv);
    VecIO<VPL>::store(pool.rgbw + v, cv);
  }
  // space_carving_kernel, :253-276: "min |tsdf| over the block >= 0.9" with fminf's NaN rule (a NaN
  // never wins) is "no voxel with |tsdf| < 0.9, and at least one that is a number": two per-lane
  // predicates, combined over the wave by ballots in finish_block (two v_cmp each, the rest on the
  // scalar unit) instead of a 512-way min reduction.
  // (the builtin on the predicates themselves, one ballot per comparison: anything that passes through
  // an integer or a logical OR of lanes' predicates first makes the compiler move a predicate that
  // already sits in a scalar register pair into a vector register and compare it again)
  auto ballot = [](bool p) { return (unsigned long long)__builtin_amdgcn_ballot_w64(p); };
  unsigned long long low = 0, num = 0;
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    const float t = __uint_as_float(tv[j]);
    low |= ballot(fabsf(t) < .9f);
    num |= ballot(t == t);
  }
  // the wave's word (uniform): voxels updated | any-low << 16 | any-number << 17
  uint32_t cnt = 0;
  if (VPL == 2 && !RATSDF_DBG(P, 14)) {
    cnt = (uint32_t)__popcll(ballot(upd2[0])) + (uint32_t)__popcll(ballot(upd2[1]));
  } else {
#pragma unroll
    for (int b = 0; b < 4; ++b) cnt += (uint32_t)__popcll(ballot((nupd >> b) & 1u)) << b;
  }
  *out_word = cnt | (low != 0ull ? 1u << 16 : 0u) | (num != 0ull ? 1u << 17 : 0u);
}

// End of a block's update: combine the WPB waves of the block (carve predicates, voxels updated)
// through LDS; one thread adds the update count to its workgroup's counter (a single device-wide counter
// would cost more than the whole update: ~90 atomics/us per address) and files the block for carving
// when min |tsdf| >= 0.9 (space_carving_kernel, voxel_tsdf.cu:253-276).
// Inside a wave everything is a ballot (integrate_block: v_cmp into a scalar register pair + s_bcnt1 on
// the scalar unit; the voxel count is the sum of the popcounts of the bits of the per-lane count).  Until round 3
// this was two 64-lane butterflies of __shfl_xor = ds_bpermute (LDS) -- ~70 VALU instructions and six
// dependent LDS round trips per block and wave, a fifth of the update.
// sred: [2][8] words, used alternately by consecutive calls (`phase` = call parity), so one
// LDS-only barrier per call is enough: a wave can only be one call ahead of the slowest reader.
// Word = voxels updated | any-low << 16 | any-number << 17.
template <int WPB>
__device__ inline void finish_block(EnginePtr E, FrameCtl* F, uint32_t* upd_wg, uint32_t par,
                                    bool carve_after_serial, const VisItem& item,
                                    bool active, uint32_t word, uint32_t wv, uint32_t part,
                                    uint32_t lane, uint32_t phase, uint32_t counter, uint32_t (*sred)[8]) {
  bool fin = active && lane == 0;
  if (WPB > 1) {
    if (lane == 0) sred[phase][wv] = word;
    lds_barrier();  // not __syncthreads(): that would also wait for the voxel stores in flight
    fin = fin && part == 0;
    if (fin) {
#pragma unroll
      for (int i = 1; i < WPB; ++i) {
        const uint32_t o = sred[phase][wv + i];
        word = ((word + o) & 0xFFFFu) | ((word | o) & 0x30000u);
      }
    }
  }
  if (fin) {
    const uint32_t n = word & 0xFFFFu;
    if (n) atomicAdd(&upd_wg[counter & (kUpdCounters - 1)], n);
    if ((word & 0x30000u) == 0x20000u) {  // rare: operands come from the engine record, not from registers held all along
      // the resolver of this frame's chained-bucket requests (serial role, possibly still running
      // beside this update) reads and edits the directory as it was BEFORE the frame's carving
      if (carve_after_serial) wait_serial_done(F, E->ctl);
      const Table tab = ld_const(&E->tab);
      const CarveBufs cb = ld_const(&E->cb[par]);
      carve_candidate(tab, cb, E->ctl, F, item);
    }
  }
}

// threads per update workgroup (VPL >= 2): 256 = one voxel block per workgroup at VPL 2
#ifndef RATSDF_INTEG_NT
#define RATSDF_INTEG_NT 256
#endif

// What every wave of the voxel update needs, by value (scalar registers).
struct IntegArgs {
  uint32_t* rgbw;
  float* tsdf;
  float* segm;
  const float4* texA;
  const uint32_t* texB;
  const VisItem* vis;
  uint32_t seg_cap;
  FrameCtl* F;
  uint32_t* upd_wg;
  uint32_t par;  // frame parity: which of the engine's two counter / list sets the frame uses
};

// Work lists: `vis` is kNumLists segments of seg_cap items holding the visible blocks that existed
// before the frame, bucketed by image tile (block_list_of); workgroup b serves list b & 7, which
// keeps a tile's texels in one XCD's L2.  This frame's new blocks come from the request list: every
// workgroup commits and integrates its share.
// SGPR cap: above 80 SGPRs the hardware admits only 6-7 instead of 8 workgroups of 256 threads per
// CU (MI355X_MICROARCH.md, residency formula), which pushed the last 20 % of the blocks into a
// second round of waves.
// Workgroups [0, n_int_wg) update voxel blocks; workgroups beyond host a share of the NEXT frame's
// candidate pass (`ahead`, kernels_cand.h): the update is bound by memory latency and leaves the
// vector ALUs mostly idle, the candidate pass is ALU work on other inputs.
// kTail: the engine's front-tail option (front_tail_role, kernels_frame.h) -- a template parameter so that the
// default kernel carries none of that path (same-box A/B: +0.2 us at 640x480, +0.6 us at 1280x720 with it in).
template <int VPL, bool kTail>
__global__ __launch_bounds__(VPL == 1 ? 512 : RATSDF_INTEG_NT) __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(VPL <= 2 ? 8 : (VPL == 4 ? 5 : 3)))) void k_integrate(
    IntegArgs A, FrameParams P, EnginePtr E, uint32_t n_int_wg, uint32_t n_serial_wg, uint32_t n_ahead_wg,
    uint32_t commit_rot, CandJob ahead) {
  constexpr bool tail_on = kTail;
  __shared__ __attribute__((aligned(16))) uint32_t role_lds[kIntegLdsWords];
  // Grid: [n_serial_wg: 0, or 8 of which the first is the frame's serial role][n_ahead_wg look-ahead
  // workgroups of the next frame's candidate pass][n_int_wg update workgroups].  The first two groups
  // are multiples of 8, so that the update workgroups keep their list <-> XCD mapping, and they come
  // FIRST: they start at once and are done before the last update workgroups are.
  if (__builtin_expect(blockIdx.x >= n_serial_wg + n_ahead_wg, 1)) {
    const uint32_t ibid = blockIdx.x - n_serial_wg - n_ahead_wg;
    const bool fused = n_serial_wg != 0;
#include "integrate_body.inc"
    return;
  }
  if (blockIdx.x >= n_serial_wg) {
    if (VPL != 1) cand_pixels_role(ahead, blockIdx.x - n_serial_wg, E->ctl, *reinterpret_cast<CandLds*>(role_lds));
    return;
  }
  if (blockIdx.x == 0)
    serial_workgroup(E, A.par, ((uint32_t)(P.W * P.H) * (uint32_t)P.S + 31u) / 32u, role_lds, RATSDF_DBG(P, 21));
  else
    serial_helper(E, A.par, role_lds, blockIdx.x, RATSDF_DBG(P, 22));
}

}  // namespace ratsdf
