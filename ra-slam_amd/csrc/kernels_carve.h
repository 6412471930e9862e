// kernels_carve.h -- the carve pass (space_carving_kernel's Delete calls, utils/tsdf/voxel_tsdf.cu:
// 253-276,878-883; VoxelHashTable::Delete, voxel_hash.cu:110-159; ReleaseBlock, voxel_mem.cu:56-61)
// without a launch of its own.
//
// The pass is made deterministic the same way as allocation: deletions happen in ascending
// hash-entry order (the order of the reference's visible list).  It has three parts with different
// dependencies, and each lives where its inputs become available:
//   carve_candidate      (inside k_integrate, by the thread that holds a block's min |tsdf|)
//                        a block in slot 0 of its home bucket is deleted on the spot -- that path
//                        takes no lock and touches nothing else (voxel_hash.cu:114-123); a list-head
//                        or chain block claims its home bucket with atomicMin(entry index) in the
//                        carve claim table and is queued
//   carve_resolve_slow   (first workgroup of the NEXT k_front, only when something is queued; the
//                        other directory readers of that launch wait for it)
//                        one winner per home bucket = the first in entry order (the bucket lock of
//                        voxel_hash.cu:125-158); winners unlink; claims are released (ResetLocks)
//   carve_finalize       (in the next frame's serial role, before its own pass; or k_settle when no frame
//                        follows) ReleaseBlock in ascending entry order of the deleted blocks:
//                          few deletes (the steady state): entries in an LDS list, every delete
//                          counts the smaller ones; many deletes: entry-indexed bitmap + popcount
//                          prefix (self-cleaning);
//                        free-list bookkeeping, frame statistics, and the consumed FrameCtl is
//                        zeroed for the frame after next.
// Nothing between k_integrate and the next allocation pass needs the free list, and the next
// k_front only needs the directory, so the order above is equivalent to the reference's
// "integrate, then carve" with the carve's serial part off the frame's critical path.
#pragma once
#include "kernels_visible.h"

namespace ratsdf {

constexpr uint32_t kSmallCarve = 2048;
constexpr uint32_t kUpdCounters = 1024;  // voxels-updated counters (workgroup index mod this)

// Barrier for data exchanged through LDS only.  __syncthreads() also waits until every global store
// of the workgroup has been acknowledged (~1 us): the serial role has global stores in flight almost
// all the time and only ever hands LDS counters across these barriers.
__device__ inline void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }


// Sum over each aligned group of 16 lanes, in every lane of the group, by DPP (registers of neighbouring
// lanes read directly; the __shfl_xor butterfly it replaces is four ds_bpermute_b32 = four dependent LDS
// round trips): pairs, quads, then the two mirror patterns -- a sum does not care which lane a partial
// sum came from.
__device__ inline uint32_t dpp_sum16(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false);  // row_half_mirror
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false);  // row_mirror
  return v;
}

__device__ inline void occ_clear(const Table& tab, uint32_t e) {
  atomicAnd(&tab.occ[e >> 6], ~(1ull << (e & 63)));
}
__device__ inline uint32_t ld_agent(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// write-through store (global_store ... sc1): leaves this XCD's L2 at once, so a reader on any CU that
// has not cached the line sees it after the storing wave's s_waitcnt vmcnt(0)
__device__ inline void st_agent(uint32_t* p, uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// A block whose min |tsdf| >= 0.9 after the update (voxel_tsdf.cu:253-276).  One thread.
__device__ inline void carve_candidate(const Table& tab, const CarveBufs& cb, Ctl* ctl, FrameCtl* F,
                                       const VisItem& it) {
  const uint32_t bucket = block_hash(it.x, it.y, it.z, tab.bucket_mask);
  // Lock-free cases: slot 0 of the home bucket (voxel_hash.cu:114-123), and the list head (slot 1)
  // of a bucket WITHOUT a chain.  The reference takes the bucket lock for the head (:125-140), but
  // with an empty chain no other delete of the pass can want that lock (slot 0 never does, chain
  // elements do not exist), and its copy-next-into-head step degenerates to clearing the entry.  The
  // link is read now, not from the visible-list copy: this frame's allocation pass may have appended
  // a chain element since.
  bool simple = it.entry == (bucket << 1);
  if (!simple && it.entry == (bucket << 1) + 1u) {
    const uint32_t w1 = ld_agent(reinterpret_cast<const uint32_t*>(tab.entries + it.entry) + 1);
    simple = (w1 >> 16) == 0;
  }
  if (simple) {
    uint32_t* pe = reinterpret_cast<uint32_t*>(tab.entries + it.entry);
    pe[1] = key1(it.z);  // offset = 0
    pe[2] = (uint32_t)-1;
    occ_clear(tab, it.entry);
    reinterpret_cast<uint32_t*>(tab.active + it.idx)[2] = (uint32_t)-1;  // the slot is empty
    const uint32_t slot = atomicAdd(&F->n_delcand, 1u);
    if (slot < cb.del_cap) {
      cb.del[slot] = DelItem{it.entry, it.idx};
    } else {
      set_error(ctl, RATSDF_ERR_CAPACITY);
    }
  } else {
    atomicMin(&tab.dclaim[bucket], it.entry);
    const uint32_t slot = atomicAdd(&F->n_slow_del, 1u);
    if (slot < cb.slow_cap) {
      cb.slow[slot] = SlowDelete{it.x, it.y, it.z, 0, it.entry, -1};
    } else {
      set_error(ctl, RATSDF_ERR_CAPACITY);
    }
  }
}

// Head / chain deletes.  All threads of one workgroup; every claim of the pass has been placed (the
// kernel that placed them has finished).
// Everything this function stores is read by OTHER workgroups of the same launch once the gate below
// has let them through (directory entries, carve claims, the state of the queued items), so every store
// is an agent-scope (write-through) store: the hand-off then needs no cache maintenance at all
// (carve_resolve_gate).  Its own loads are plain: what it reads was written by earlier launches.
__device__ inline void carve_resolve_slow(const Table& tab, const CarveBufs& cb, Ctl* ctl, FrameCtl* F) {
  const uint32_t tid = threadIdx.x, nt = block_threads();
  SlowDelete* slow = cb.slow;
  uint32_t ns = F->n_slow_del;
  if (ns > cb.slow_cap) ns = cb.slow_cap;
  // decide every winner before any claim is released (the decision is parked in the item itself --
  // word 1 = z | state << 16 -- written and re-read past the caches by the same thread)
  for (uint32_t j = tid; j < ns; j += nt) {
    const SlowDelete s = slow[j];
    const uint32_t bucket = block_hash(s.x, s.y, s.z, tab.bucket_mask);
    const uint32_t win = ld_agent(&tab.dclaim[bucket]) == s.entry ? 1u : 0u;
    st_agent(reinterpret_cast<uint32_t*>(&slow[j]) + 1, (uint32_t)(uint16_t)s.z | (win << 16));
  }
  __syncthreads();
  for (uint32_t j = tid; j < ns; j += nt) {
    SlowDelete s = slow[j];
    s.state = (uint16_t)(ld_agent(reinterpret_cast<const uint32_t*>(&slow[j]) + 1) >> 16);
    const uint32_t bucket = block_hash(s.x, s.y, s.z, tab.bucket_mask);
    st_agent(&tab.dclaim[bucket], kInf);  // ResetLocks
    if (!s.state) continue;
    const uint32_t k0 = key0(s.x, s.y), k1 = key1(s.z);
    uint32_t last = (bucket << 1) + 1;
    uint32_t* ph = reinterpret_cast<uint32_t*>(tab.entries + last);
    const EntryWords h = load_entry(tab.entries, last);
    int32_t freed = -1;
    if (entry_matches(h, k0, k1)) {                                     // voxel_hash.cu:125-140
      const uint32_t nxt = (last + (uint32_t)entry_offset(h)) & tab.entry_mask;
      uint32_t* pn = reinterpret_cast<uint32_t*>(tab.entries + nxt);
      const EntryWords nw = load_entry(tab.entries, nxt);
      freed = h.idx;
      const int noff = entry_offset(nw);
      const int16_t hoff = noff ? (int16_t)(entry_offset(h) + noff) : (int16_t)0;
      // (a head that is its own successor -- empty chain -- is just cleared: in the reference the copy
      // onto itself is followed by the clear of the same entry)
      if (nxt != last) {
        st_agent(ph + 0, nw.w0);
        st_agent(ph + 1, (nw.w1 & 0xFFFFu) | ((uint32_t)(uint16_t)hoff << 16));
        st_agent(ph + 2, (uint32_t)nw.idx);
      }
      st_agent(pn + 1, nw.w1 & 0xFFFFu);
      st_agent(pn + 2, (uint32_t)-1);
      occ_clear(tab, nxt);  // the head keeps its bit unless it was its own successor
      // Table::active: the deleted block's slot is empty, the block that moved into the head lives at `last` now
      if (freed >= 0) st_agent(reinterpret_cast<uint32_t*>(tab.active + freed) + 2, (uint32_t)-1);
      if (nxt != last && nw.idx >= 0) st_agent(reinterpret_cast<uint32_t*>(tab.active + nw.idx) + 3, last);
      if (nxt != last) mark_dirty(tab, last);  // (the block that moved into the head: same position, new entry words)
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
          st_agent(pl + 1, (lw.w1 & 0xFFFFu) | ((uint32_t)(uint16_t)link << 16));
          mark_dirty(tab, last);  // (the predecessor's link has changed)
          freed = cw.idx;
          st_agent(pcur + 1, cw.w1 & 0xFFFFu);
          st_agent(pcur + 2, (uint32_t)-1);
          occ_clear(tab, cur);
          if (freed >= 0) st_agent(reinterpret_cast<uint32_t*>(tab.active + freed) + 2, (uint32_t)-1);
          break;
        }
        last = cur;
      }
    }
    if (freed >= 0) {  // word 1 of the item = z | state << 16
      st_agent(reinterpret_cast<uint32_t*>(&slow[j]) + 3, (uint32_t)freed);
      st_agent(reinterpret_cast<uint32_t*>(&slow[j]) + 1, (uint32_t)(uint16_t)s.z | (2u << 16));
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains before the barrier
  __syncthreads();
  if (tid == 0) st_agent(&F->slow_resolved, 1u);
}

// Directory readers of the next launch: make sure the queued head / chain deletes of the previous
// frame have happened.  Workgroup 0 of the launch does them; the others wait for its flag.
//
// The hand-off uses NO cache maintenance (until round 2: __threadfence() + release store in workgroup
// 0 and an agent-scope acquire fence in every waiting workgroup; in frames with queued deletes -- 60 %
// of the frames of a 68 k-block map -- that cost the launch a constant +38 us: the release wrote back
// the L2 of workgroup 0's XCD, which the look-ahead candidate workgroups of the same launch keep
// filling with dirty texel lines, and ~500 workgroups invalidated the L1 of CUs they share with
// those).  Instead:
//   writer   every store of carve_resolve_slow is an agent-scope (sc1, write-through) store; every
//            storing wave drains (s_waitcnt vmcnt(0)), the workgroup's barrier, then thread 0 stores
//            the flag the same way;
//   readers  poll the flag with relaxed agent-scope loads (one lane; an acquire per poll would flush
//            the CU's L1 each time), then a workgroup barrier, then PLAIN loads.  That is sound because
//            no workgroup of this launch loads a line of the directory (entries, claims, queued
//            items) before it has passed the gate -- L1 and L2 start the launch invalidated and the
//            write-through stores leave no stale copy in the writer's L2 -- with two exceptions: the
//            occupancy words visible_append_role requests before the gate, which it therefore loads
//            again at agent scope (atomics on them happen at the memory side), and the lines the
//            resolver itself loaded on workgroup 0's CU, which that workgroup invalidates (its own
//            L1 only) before it publishes the flag.
// The wait relies on nothing HIP promises (waiters all have a higher index than workgroup 0, which the
// dispatcher has so far always started first), so it is BOUNDED, like the serial role's: after
// kGateTimeoutTicks of the 100 MHz wall clock the waiter gives up and records RATSDF_ERR_TIMEOUT --
// an error the caller sees at the next synchronisation, never a hung GPU.  The bound is a safety net,
// not a deadline (the resolver takes microseconds; 65 536 queued chain deletes with every chain walk
// missing the caches, on a device shared with other queues or serialised under a profiler, stay far
// below it).  Uniform per workgroup.
// Returns kGateOpen (nothing was queued), kGateWaited (data requested before the call may be stale)
// or kGateExpired: the caller must NOT read the directory -- it may be half-edited -- and skips its
// work; the frame is then incomplete, which the sticky error reports.
constexpr unsigned long long kGateTimeoutTicks = 200000000;  // 2 s, as kSerialWaitTicks
__device__ inline GateResult carve_resolve_gate(const Table& tab, const CarveBufs& cb, Ctl* ctl,
                                                FrameCtl* Fprev, bool withhold = false) {
  if (Fprev->n_slow_del == 0) return kGateOpen;  // the steady state
  __shared__ uint32_t gate_state;
  if (blockIdx.x == 0) {
    if (ld_agent(&Fprev->slow_resolved) == 0) carve_resolve_slow(tab, cb, ctl, Fprev);  // uniform
    // (carve_resolve_slow ends with every wave drained and a barrier.)  The resolver's own PLAIN loads
    // have left lines in THIS CU's L1 that its write-through stores then changed underneath: one
    // invalidate of this L1 (buffer_inv sc1, no write-back) BEFORE the flag goes out, so that neither
    // this workgroup nor a waiting workgroup resident on the same CU can hit them afterwards.
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (!withhold) st_agent(&Fprev->slow_resolved, 2u);
    }
    __syncthreads();
    return kGateWaited;
  }
  if (threadIdx.x == 0) {
    uint32_t st = kGateWaited;
    const unsigned long long t0 = (unsigned long long)wall_clock64();
    while (ld_agent(&Fprev->slow_resolved) != 2u) {
      if ((unsigned long long)wall_clock64() - t0 > kGateTimeoutTicks) {
        set_error(ctl, RATSDF_ERR_TIMEOUT);
        st = kGateExpired;
        break;
      }
      __builtin_amdgcn_s_sleep(16);
    }
    gate_state = st;
  }
  __syncthreads();
  return (GateResult)gate_state;
}

// The positions a finished frame deleted go to the directory-delta log (Table::del_log): every simple delete of
// the frame's list and every head / chain delete that happened.  All threads of one workgroup; one returning
// atomic for the lot.  `lds`: 2 words.
__device__ inline void log_deleted_positions(const Table& tab, const CarveBufs& cb, uint32_t nd, uint32_t ns,
                                             uint32_t* lds) {
  const uint32_t tid = threadIdx.x, nt = block_threads();
  if (!tab.delta_on) return;  // uniform: nobody consumes deltas
  if (tid == 0) lds[0] = 0;
  __syncthreads();
  uint32_t mine = 0;  // slow deletes of this thread that happened
  for (uint32_t j = tid; j < ns; j += nt) mine += (ld_agent(reinterpret_cast<const uint32_t*>(&cb.slow[j]) + 1) >> 16) == 2u;
  const uint32_t first = mine ? atomicAdd(&lds[0], mine) : 0u;
  __syncthreads();
  const uint32_t n_slow_done = lds[0];
  if (tid == 0) lds[1] = nd + n_slow_done ? atomicAdd(tab.del_count, nd + n_slow_done) : 0u;
  __syncthreads();
  const uint32_t base = lds[1];
  // (a simple delete leaves the block's position in its cleared entry -- carve_candidate rewrites the second and
  // third word only -- and nothing can have filled that entry again: this runs before the next allocation pass
  // commits anything)
  for (uint32_t i = tid; i < nd; i += nt) {
    const EntryWords ew = load_entry(tab.entries, cb.del[i].entry);
    if (base + i < tab.del_cap) tab.del_log[base + i] = make_uint2(ew.w0, ew.w1 & 0xFFFFu);
  }
  uint32_t at = base + nd + first;
  for (uint32_t j = tid; j < ns; j += nt) {
    const SlowDelete sd = cb.slow[j];
    if ((ld_agent(reinterpret_cast<const uint32_t*>(&cb.slow[j]) + 1) >> 16) != 2u) continue;
    if (at < tab.del_cap) tab.del_log[at] = make_uint2(key0(sd.x, sd.y), key1(sd.z));
    ++at;
  }
}

// Pool releases of a finished frame with few deletes (the steady state), spread over kReleaseWGs
// workgroups of the next k_front: ReleaseBlock in ascending entry order (voxel_mem.cu:56-60) means
// delete w goes to heap[num_free + (number of deleted entries below its own)].  Every workgroup
// holds the whole list of deleted entries in LDS; 16 lanes share a delete and each scans a 16th of
// the list.  num_free is the value the frame's own allocation pass left (nothing changes it until
// the next frame's serial role, which adds the number of releases).  The slow (head / chain) deletes of the
// frame have been resolved before this runs (carve_resolve_gate).
constexpr uint32_t kReleaseWGs = 16;

// `scratch`: 2 * kSmallCarve + 4 words of LDS, 16-byte aligned.
__device__ inline void carve_release_role(const Table& tab, const Pool& pool, const CarveBufs& cb, Ctl* ctl,
                                          FrameCtl* Fp, uint32_t wg, uint32_t* scratch) {
  uint32_t* del_entry = scratch;
  int32_t* del_pool = reinterpret_cast<int32_t*>(scratch + kSmallCarve);
  uint32_t& n_extra = scratch[2 * kSmallCarve];
  const uint32_t tid = threadIdx.x, nt = block_threads();
  // one round of loads: counters, free count and (speculatively) the head of the delete list
  const uint32_t pend = Fp->pending;
  uint32_t nd = Fp->n_delcand, ns = Fp->n_slow_del;
  const int32_t nf = ctl->num_free;
  DelItem first = cb.del[tid < cb.del_cap ? tid : 0];
  if (!pend) return;  // uniform
  if (nd > cb.del_cap) nd = cb.del_cap;
  if (ns > cb.slow_cap) ns = cb.slow_cap;
  if (nd + ns > kSmallCarve || nd + ns == 0) return;  // many deletes: carve_finalize does them
  if (tid == 0) n_extra = 0;
  __syncthreads();
  if (tid < nd) {
    del_entry[tid] = first.entry;
    del_pool[tid] = first.idx;
  }
  for (uint32_t i = tid + nt; i < nd; i += nt) {
    const DelItem d = cb.del[i];
    del_entry[i] = d.entry;
    del_pool[i] = d.idx;
  }
  for (uint32_t j = tid; j < ns; j += nt) {  // head / chain deletes that happened: rare
    const SlowDelete sd = cb.slow[j];
    if (sd.state == 2) {
      const uint32_t slot = nd + atomicAdd(&n_extra, 1u);
      del_entry[slot] = sd.entry;
      del_pool[slot] = sd.freed;
    }
  }
  __syncthreads();
  const uint32_t n = nd + n_extra;
  const uint32_t n64 = (n + 63u) & ~63u;
  if (tid < n64 - n) del_entry[n + tid] = kInf;  // n64 <= kSmallCarve (a multiple of 64)
  __syncthreads();
  const uint4* v = reinterpret_cast<const uint4*>(del_entry);
  const uint32_t nchunks = n64 >> 2, sub = tid & 15u;
  const uint32_t per_pass = (nt >> 4) * kReleaseWGs;
  for (uint32_t base = 0; base < n; base += per_pass) {  // uniform
    const uint32_t item = base + wg * (nt >> 4) + (tid >> 4);
    const uint32_t mine = item < n ? del_entry[item] : 0u;
    uint32_t k = 0;
    for (uint32_t c = sub; c < nchunks; c += 16) {
      const uint4 x = v[c];
      k += (x.x < mine) + (x.y < mine) + (x.z < mine) + (x.w < mine);
    }
    k = dpp_sum16(k);  // over the 16 lanes that share the delete
    // (write-through: the frame's serial role at the tail of the same launch pops these, front_tail_role)
    if (sub == 0 && item < n) {
      if (tab.tail_on) st_agent(reinterpret_cast<uint32_t*>(&pool.heap[(uint32_t)nf + k]), (uint32_t)del_pool[item]);
      else pool.heap[(uint32_t)nf + k] = del_pool[item];
    }
  }
  if (wg == 0 && tab.delta_on) {  // uniform: the frame's deleted positions, for the directory delta
    __syncthreads();
    log_deleted_positions(tab, cb, nd, ns, scratch + 2 * kSmallCarve + 4);
  }
}

// ReleaseBlock calls of a finished frame in ascending entry order + bookkeeping.  All threads of one
// workgroup.  `scratch`: 2 * kSmallCarve + 80 words of LDS (16-byte aligned).  `nf` = free blocks before the releases.
// Returns the number of blocks released (uniform).
// kLog: the frame's deleted positions go to the directory-delta log from here (k_settle: no frame followed, so
// no release role has logged them).  Inside a frame's launches (kLog false) the release role of k_front logs
// them; only a frame with more deletes than that role takes leaves them unlogged, and then the log is marked
// unusable (the next delta export reports an overflow and the caller takes a whole directory): the logging
// code itself stays out of k_integrate, which has no register to spare.
template <bool kLog = false>
__device__ inline uint32_t carve_finalize(const Table& tab, const Pool& pool, const CarveBufs& cb,
                                          Ctl* ctl, FrameCtl* F, ratsdf_frame_stats* stats, int32_t nf,
                                          uint32_t* scratch) {
  if (F->pending == 0) return 0;  // uniform
  const uint32_t tid = threadIdx.x, nt = block_threads();
  uint32_t* lds = scratch;                    // [0,32): scan scratch, [32]: counter, [33]: result
  uint32_t* del_entry = scratch + 64;
  int32_t* del_pool = reinterpret_cast<int32_t*>(scratch + 64 + kSmallCarve + 16);
  uint32_t nd = F->n_delcand, ns = F->n_slow_del;
  if (nd > cb.del_cap) nd = cb.del_cap;
  if (ns > cb.slow_cap) ns = cb.slow_cap;
  // voxels updated: per-workgroup counters of k_integrate (consumed here)
  uint32_t upd_part = 0;
  for (uint32_t i = tid; i < kUpdCounters; i += nt) {
    const uint32_t u = cb.upd_wg[i];
    if (u) {
      upd_part += u;
      cb.upd_wg[i] = 0;
    }
  }
  if (tid == 0) lds[32] = 0;
  __syncthreads();
  if (kLog) log_deleted_positions(tab, cb, nd, ns, lds + 40);
  else if (nd + ns > kSmallCarve && tid == 0 && tab.delta_on) atomicOr(tab.del_count, 0x80000000u);
  uint32_t n_del = 0;
  if (nd + ns <= kSmallCarve) {
    for (uint32_t i = tid; i < nd; i += nt) {
      const DelItem d = cb.del[i];
      del_entry[i] = d.entry;
      del_pool[i] = d.idx;
    }
    for (uint32_t j = tid; j < ns; j += nt) {
      const SlowDelete s = cb.slow[j];
      if (s.state == 2) {
        const uint32_t slot = nd + atomicAdd(&lds[32], 1u);
        del_entry[slot] = s.entry;
        del_pool[slot] = s.freed;
      }
    }
    __syncthreads();
    n_del = nd + lds[32];
    // rank of a delete = number of deleted entries below its own (all in LDS)
    lds_rank_all(del_entry, n_del, [&](uint32_t w, uint32_t k) {
      pool.heap[(uint32_t)nf + k] = del_pool[w];                          // voxel_mem.cu:56-60
    });
  } else {
    for (uint32_t i = tid; i < nd; i += nt) bitmap_set(cb.bitmap, cb.summary, cb.del[i].entry);
    for (uint32_t j = tid; j < ns; j += nt)
      if (cb.slow[j].state == 2) bitmap_set(cb.bitmap, cb.summary, cb.slow[j].entry);
    __syncthreads();
    const uint32_t nwords = tab.num_entry >> 5;
    const uint32_t chunk = bitmap_chunk(nwords, nt);
    const uint32_t sum = chunk_popcount(cb.bitmap, cb.summary, nwords, chunk);
    uint32_t total = 0;
    const uint32_t excl = block_exclusive_scan(sum, lds, &total);
    bitmap_write_prefix(cb.bitmap, cb.summary, cb.prefix, nwords, chunk, sum, excl);
    __syncthreads();
    n_del = total;
    // (four deletes per thread in flight: item -> prefix word / bitmap word are dependent loads)
    constexpr int kU = 4;
    for (uint32_t base = tid; base < nd; base += nt * kU) {  // uniform
      DelItem d[kU];
      uint32_t pw[kU], bw[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const uint32_t i = base + (uint32_t)u * nt;
        d[u] = cb.del[i < nd ? i : 0];
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const uint32_t w = d[u].entry >> 5;
        pw[u] = cb.prefix[w < nwords ? w : 0];
        bw[u] = cb.bitmap[w < nwords ? w : 0];
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const uint32_t i = base + (uint32_t)u * nt;
        if (i < nd) pool.heap[(uint32_t)nf + pw[u] + __popc(bw[u] & ((1u << (d[u].entry & 31)) - 1u))] = d[u].idx;
      }
    }
    for (uint32_t j = tid; j < ns; j += nt) {
      const SlowDelete s = cb.slow[j];
      if (s.state == 2) pool.heap[(uint32_t)nf + bitmap_rank(cb.bitmap, cb.prefix, s.entry)] = s.freed;
    }
    __syncthreads();  // every reader of the delete bitmap is done: leave it clean for the next pass
    bitmap_clean(cb.bitmap, cb.summary, nwords);
  }
  uint32_t upd = 0;
  (void)block_exclusive_scan(upd_part, lds, &upd);
  if (tid == 0) {
    // (the frame's counters are read here, not at the top: ten values held across the bitmap passes
    // pushed the kernels that inline this function into scratch memory)
    const uint32_t n_win = F->n_win, n_slow_req = F->n_slow;
    const uint32_t nv = frame_visible_blocks(F);
    if (stats) {
      stats->visible_blocks = (int32_t)nv;
      stats->updated_voxels = (int32_t)upd;
      stats->allocated_blocks = (int32_t)n_win;
      stats->deleted_blocks = (int32_t)n_del;
      stats->active_blocks = tab.num_block - (nf + (int32_t)n_del);
      stats->slow_requests = (int32_t)n_slow_req;
      ctl->totals[0] += 1;
      ctl->totals[1] += nv;
      ctl->totals[2] += upd;
      ctl->totals[3] += n_win;
      ctl->totals[4] += n_del;
    }
    // counters ready for the frame after next
    zero_frame_ctl(F, tab.tail_on != 0);
  }
  __syncthreads();
  return n_del;
}

// explicit delete list (test hook, utils/tests/voxel_mem_test.cu:56-78): every listed block that
// exists becomes a carve candidate; k_settle then deletes them in entry order
__global__ void k_delete_list(Table tab, const int16_t* pos, int n, CarveBufs cb, Ctl* ctl,
                              uint32_t par) {
  const int i = blockIdx.x * block_threads() + threadIdx.x;
  FrameCtl* F = &ctl->fr[par];
  if (i == 0) F->pending = 1;
  if (i >= n) return;
  EntryWords w;
  const int x = pos[3 * i], y = pos[3 * i + 1], z = pos[3 * i + 2];
  const uint32_t e = find_block(tab, x, y, z, &w);
  if (e == kInf) return;
  carve_candidate(tab, cb, ctl, F, VisItem{(int16_t)x, (int16_t)y, (int16_t)z, 0, w.idx, e});
}

// The tail of the last frame when no other frame follows it (queries, statistics, test hooks and
// ratsdf_synchronize call this first): queued head / chain deletes + carve_finalize.
__global__ __launch_bounds__(1024) void k_settle(Table tab, Pool pool, CarveBufs cb, Ctl* ctl,
                                                 uint32_t par, ratsdf_frame_stats* stats) {
  __shared__ __attribute__((aligned(16))) uint32_t scratch[2 * kSmallCarve + 80];
  FrameCtl* F = &ctl->fr[par];
  if (F->n_slow_del != 0 && ld_agent(&F->slow_resolved) == 0) {  // uniform
    carve_resolve_slow(tab, cb, ctl, F);
    __threadfence();
    __syncthreads();
  }
  const int32_t nf = ctl->num_free;
  const uint32_t n_del = carve_finalize<true>(tab, pool, cb, ctl, F, stats, nf, scratch);
  if (threadIdx.x == 0 && n_del) ctl->num_free = nf + (int32_t)n_del;
}

}  // namespace ratsdf
