"""Multi-GPU host logic: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on
MI355X, "gloo" on CPU for tests).  The reference has no multi-GPU code (single process, single GPU,
modules/tsdf_module.h:152-164); this is the new design of SURVEY 8e:

* block-ownership sharding ("spatial subvolumes"): every rank sees the full frame but inserts only
  the blocks it owns, owner = floormod(block.x >> slab_bits, world).  Integration needs no
  communication.  The per-rank engines are created with shard_rank / shard_count.
* frame-batched streams: every rank integrates its own stream into its own map.
* in both modes the ranks periodically ALL-GATHER their block directories (12-byte entries) so any
  rank can tell which rank holds which block (queries across seams, global statistics).  Messages
  are small (10^3..10^5 entries), so one fixed-capacity all-gather + one count all-gather is used:
  DirectoryExchange sends whole directories, DirectoryDeltaExchange only what changed since the
  previous exchange (every rank keeps replicas).
"""
import numpy as np

from ._abi import BLOCK_DTYPE


def owner_of(block_x, world, slab_bits=2):
    """Rank owning a block with x block-coordinate `block_x` (numpy-friendly, floor modulo)."""
    return np.mod(np.asarray(block_x, dtype=np.int64) >> slab_bits, world)


class DirectoryExchange:
    """All-gather of block directories with fixed-capacity buffers.

    device=None -> CPU tensors (gloo); otherwise a torch cuda device (nccl/RCCL) and the engine
    writes its directory straight into the send buffer (ratsdf_export_directory_device).
    `capacity` defaults to the size of the engine's block pool (2^block_bits entries): a directory
    can then never be truncated.  A smaller capacity is legal; result() raises if any rank's
    directory did not fit (the count that travels is the true one, not the clamped one)."""

    def __init__(self, capacity=None, device=None, engine=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        if capacity is None:
            capacity = 1 << (engine.block_bits if engine is not None else 18)
        self.capacity = int(capacity)
        self.device = device
        kw = dict(dtype=torch.int32, device=device if device is not None else "cpu")
        self.send = torch.zeros(self.capacity * 3, **kw)
        self.count = torch.zeros(1, **kw)
        self.recv = torch.zeros(self.world * self.capacity * 3, **kw)
        self.counts = torch.zeros(self.world, **kw)
        self._engine_stream = None

    def fill_from_engine_device(self, engine):
        """HIP engine -> send buffer, on the engine's stream (no host round trip).  The collective
        runs on torch's current stream: all_gather() orders the two with events."""
        engine.export_directory_device(self.send.data_ptr(), self.capacity, self.count.data_ptr())
        self._engine_stream = self.torch.cuda.ExternalStream(engine.stream(), device=self.device)

    def fill_from_numpy(self, blocks):
        """blocks: structured array of BLOCK_DTYPE (e.g. Engine.dump_directory()[1])."""
        n = min(len(blocks), self.capacity)
        raw = np.zeros(self.capacity * 3, dtype=np.int32)
        raw[:n * 3] = np.ascontiguousarray(blocks[:n]).view(np.int32).reshape(-1)
        self.send.copy_(self.torch.from_numpy(raw))
        self.count.fill_(len(blocks))  # the true count, like the device export
        self._engine_stream = None

    def all_gather(self):
        torch = self.torch
        es = self._engine_stream
        if es is not None:  # engine stream -> (event) -> torch's current stream
            ev = torch.cuda.Event()
            ev.record(es)
            torch.cuda.current_stream(self.device).wait_event(ev)
        if self.world == 1:
            self.recv.copy_(self.send)
            self.counts.copy_(self.count)
        else:
            self.dist.all_gather_into_tensor(self.recv, self.send)
            self.dist.all_gather_into_tensor(self.counts, self.count)
        if es is not None:  # ... and back: the engine may not overwrite `send` before the collective read it
            ev2 = torch.cuda.Event()
            ev2.record(torch.cuda.current_stream(self.device))
            es.wait_event(ev2)

    def result(self):
        """List (one per rank) of structured BLOCK_DTYPE arrays."""
        counts = self.counts.cpu().numpy()
        if int(counts.max(initial=0)) > self.capacity:
            raise OverflowError(f"block directory of rank {int(counts.argmax())} has {int(counts.max())} "
                                f"entries, exchange capacity is {self.capacity}")
        raw = self.recv.cpu().numpy().reshape(self.world, self.capacity * 3)
        out = []
        for r in range(self.world):
            n = int(counts[r])
            out.append(raw[r, :n * 3].copy().view(BLOCK_DTYPE))
        return out


class DirectoryDeltaExchange(DirectoryExchange):
    """The same exchange carrying only what CHANGED since the previous one ("all-gather of
    the block directory delta ... every N frames"): a rank sends the entries it has added (or whose
    pool index changed) and the positions it has deleted; every rank keeps a replica of every rank's
    directory and applies the deltas.  The first exchange is a full one by construction (everything
    is new).  result() returns the replicas in DirectoryExchange.result()'s form (entries sorted by
    position), so multi.query() takes either.

    Wire format per rank, ONE collective: int32 {added, deleted} followed by `delta_capacity`
    12-byte entries (the added ones first, then the deleted ones: position only).

    Nothing on the exchange path waits for the device:
      * the engine exports its whole directory into a pool-sized device buffer (one small kernel; the
        buffer holds 2^block_bits entries, so the export cannot overflow and never touches the engine's
        sticky error);
      * the delta is taken with fixed-shape tensor ops on that device (sort, searchsorted, cumulative
        sums, scatter into the payload) -- no .item(), no boolean-mask indexing, whose result sizes the
        host would have to read;
      * the counts travel inside the payload; a delta that does not fit `delta_capacity` is clamped in
        the buffer and its TRUE size is what travels, so every rank learns of the overflow from the
        same collective and raises the same OverflowError -- no rank is left waiting in a collective;
      * the received deltas are applied to the replicas one exchange LATER (or when result() is
        called): by then the collective has long finished, so reading its counts does not stall the
        host while the next batch of frames waits to be enqueued.
    """

    def __init__(self, capacity=None, device=None, engine=None, delta_capacity=None):
        super().__init__(capacity, device, engine)
        torch = self.torch
        dev = device if device is not None else "cpu"
        i32 = dict(dtype=torch.int32, device=dev)
        # the payload: by default a quarter of the pool (a delta is a frame batch's worth of blocks).  The
        # FIRST exchange carries whole directories and uses pool-sized buffers of its own, once.
        self.delta_capacity = int(delta_capacity) if delta_capacity else max(self.capacity // 4, 1024)
        self.delta_capacity = min(self.delta_capacity, self.capacity)
        self._i32 = i32
        self.full = torch.zeros(self.capacity * 3, **i32)          # export target (whole directory)
        self.full_count = torch.zeros(1, **i32)
        self._first = True
        self._C = self.capacity                                    # payload entries of the exchange in hand
        self.send = torch.zeros(2 + 3 * (self._C + 1), **i32)      # header, C entries, one dump slot
        self.recv2 = [torch.zeros(self.world * (2 + 3 * (self._C + 1)), **i32)]
        self._slot = 0
        self._todo = None                                          # received, not yet applied
        self._prev = torch.zeros((self.capacity, 3), **i32)        # my directory as last sent, sorted
        self._prev_key = torch.full((self.capacity,), 2 ** 62, dtype=torch.int64, device=dev)
        self._replica = [torch.zeros((0, 3), **i32) for _ in range(self.world)]
        self.last_sent = (0, 0)
        self._last_counts = None
        self.resyncs = 0                                           # exchanges restarted because an engine's delete log overflowed

    def _restart(self):
        """Back to the state before the first exchange: the next one carries whole directories again (in
        pool-sized buffers) and the replicas are rebuilt from it.  After a delta that did not fit, the
        truncated entries are in nobody's replica and the sender's `_prev` has already moved past them: a
        caller that catches the OverflowError and carries on gets consistent replicas from the next exchange
        on instead of silently diverged ones.  Every rank takes this path together (they all see the same
        counts), so the buffer sizes of the next collective agree."""
        torch = self.torch
        self._first = True
        self._C = self.capacity
        self.send = torch.zeros(2 + 3 * (self._C + 1), **self._i32)
        self.recv2 = [torch.zeros(self.world * (2 + 3 * (self._C + 1)), **self._i32)]
        self._slot = 0
        self._todo = None
        self._prev_key = torch.full_like(self._prev_key, 2 ** 62)
        self._replica = [rep[:0] for rep in self._replica]

    def _pos_key(self, rows):
        """int64 key of the block position (x, y, z int16 in the first 6 bytes of an entry)."""
        i64 = self.torch.int64
        return ((rows[:, 1].to(i64) & 0xFFFF) << 32) | (rows[:, 0].to(i64) & 0xFFFFFFFF)

    def fill_from_engine_device(self, engine):
        engine.export_directory_device(self.full.data_ptr(), self.capacity, self.full_count.data_ptr())
        self._engine_stream = self.torch.cuda.ExternalStream(engine.stream(), device=self.device)

    def fill_delta_from_engine_device(self, engine):
        """The engine's own record of what changed since the previous export (ratsdf_export_directory_delta_device:
        a dirty bit per directory entry + a log of deleted positions) straight into the payload: two small kernels on
        the engine's stream instead of a sort of the whole directory here (_make_delta).  Not for the first exchange
        (`fill_from_engine` chooses)."""
        body = self.send[2:]
        engine.export_directory_delta_device(body.data_ptr(), self._C, self.send.data_ptr())
        self._engine_stream = self.torch.cuda.ExternalStream(engine.stream(), device=self.device)
        self._have_delta = True

    def fill_from_engine(self, engine):
        """What bench.py --gpus N calls once per step: the whole directory the first time (and after an overflow),
        the engine's delta log from then on."""
        if self._first:
            self.fill_from_engine_device(engine)
            engine.export_directory_delta_device(0, 0, 0)   # the whole directory is on its way: forget the changes so far
        else:
            self.fill_delta_from_engine_device(engine)

    def fill_from_numpy(self, blocks):
        n = len(blocks)
        if n > self.capacity:   # (cannot happen with the default capacity = the engine's pool)
            raise ValueError(f"block directory has {n} entries, the export buffer holds {self.capacity}")
        raw = np.zeros(self.capacity * 3, dtype=np.int32)
        raw[:n * 3] = np.ascontiguousarray(blocks).view(np.int32).reshape(-1)
        self.full.copy_(self.torch.from_numpy(raw))
        self.full_count.fill_(n)
        self._engine_stream = None

    def _make_delta(self):
        """payload from (current export) minus (previous export); fixed shapes, no host read."""
        torch = self.torch
        cap, C = self.capacity, self._C
        BIG = 2 ** 62
        rows = self.full.reshape(cap, 3)
        valid = torch.arange(cap, device=rows.device) < self.full_count.to(torch.int64)
        key = torch.where(valid, self._pos_key(rows), torch.full_like(self._prev_key, BIG))
        key, order = torch.sort(key)
        now = rows[order]
        valid = key < BIG
        # added / changed: my position is new, or its pool index / offset word differs
        at = torch.searchsorted(self._prev_key, key).clamp_(max=cap - 1)
        same = (self._prev_key[at] == key) & (self._prev[at, 2] == now[:, 2]) & (self._prev[at, 1] == now[:, 1])
        add = valid & ~same
        # gone: a previous position that no longer exists
        pvalid = self._prev_key < BIG
        at2 = torch.searchsorted(key, self._prev_key).clamp_(max=cap - 1)
        gone = pvalid & ~(key[at2] == self._prev_key)
        na, nd = add.sum(), gone.sum()
        # scatter into the payload: added entries at [0, na), deleted at [na, na + nd); whatever does not
        # fit goes to the dump slot C (the header carries the true counts, so everybody sees the overflow)
        pa = torch.cumsum(add, 0) - 1
        pd = na + torch.cumsum(gone, 0) - 1
        body = self.send[2:].reshape(C + 1, 3)
        body.zero_()
        ia = torch.where(add & (pa < C), pa, torch.full_like(pa, C))
        body.index_copy_(0, ia, now)
        idl = torch.where(gone & (pd < C), pd, torch.full_like(pd, C))
        body.index_copy_(0, idl, self._prev)
        self.send[0] = na.to(torch.int32)
        self.send[1] = nd.to(torch.int32)
        self._prev = now.clone()
        self._prev_key = key

    def _apply(self, recv, C):
        """host side of an exchange that has completed: counts, overflow, replicas"""
        torch = self.torch
        rows = recv.reshape(self.world, 2 + 3 * (C + 1))
        counts = rows[:, :2].cpu()
        self._last_counts = counts
        if any(int(counts[r, 1]) == 0x7FFFFFFF for r in range(self.world)):
            # An engine could not log its deletes (ratsdf_export_directory_delta_device: a frame that carves more
            # than kSmallCarve blocks inside a batch -- a new view, a large move -- is finalised where logging is
            # not possible, and the log itself is finite): its delta is unusable.  Ordinary operation, not an
            # error: every rank reads the same counts, so every rank restarts together and the next exchange
            # carries whole directories; the replicas stay as they were until then.
            self._restart()
            self.resyncs += 1
            return True
        over = [(r, int(counts[r, 0]), int(counts[r, 1])) for r in range(self.world)
                if int(counts[r, 0]) + int(counts[r, 1]) > C]
        if over:   # every rank sees the same counts: every rank raises, nobody is left in a collective
            r, na, nd = over[0]
            # the truncated entries are in nobody's replica and _make_delta has already moved on: the next
            # exchange of a caller that catches the error resends the whole directory (every rank resets,
            # because every rank is here)
            self._restart()
            raise OverflowError(f"directory delta of rank {r}: {na} added + {nd} deleted entries, "
                                f"delta_capacity is {C}")
        self.last_sent = (int(counts[self.rank, 0]), int(counts[self.rank, 1]))
        for r in range(self.world):
            na, nd = int(counts[r, 0]), int(counts[r, 1])
            body = rows[r, 2:].reshape(C + 1, 3)
            rep = self._replica[r]
            drop = body[:na + nd]                      # replaced and deleted positions
            if len(rep) and len(drop):
                rep = rep[~torch.isin(self._pos_key(rep), self._pos_key(drop))]
            rep = torch.cat([rep, body[:na].clone()])
            self._replica[r] = rep[torch.argsort(self._pos_key(rep))]
        return False

    def flush(self):
        """apply what has been received (reads the last collective's counts: waits for it); True = the exchange
        restarted instead (an engine's delete log had overflowed): the next one carries whole directories"""
        if self._todo is not None:
            (todo, C), self._todo = self._todo, None
            return self._apply(todo, C)
        return False

    def all_gather(self):
        torch = self.torch
        es = self._engine_stream
        if es is not None:  # engine stream -> (event) -> torch's current stream
            ev = torch.cuda.Event()
            ev.record(es)
            torch.cuda.current_stream(self.device).wait_event(ev)
        if getattr(self, "_have_delta", False):   # the engine has written header and payload itself
            self._have_delta = False
        else:
            self._make_delta()
        recv = self.recv2[self._slot % len(self.recv2)]
        self._slot += 1
        C = self._C
        if self.world == 1:
            recv.copy_(self.send)
        else:
            self.dist.all_gather_into_tensor(recv, self.send)
        if es is not None:  # the engine may not overwrite the export buffer before the delta was taken
            ev2 = torch.cuda.Event()
            ev2.record(torch.cuda.current_stream(self.device))
            es.wait_event(ev2)
        if self.flush():      # the PREVIOUS exchange (finished long ago): no stall
            return            # ... it restarted the exchange: this one's deltas build on replicas nobody has
        self._todo = (recv, C)
        if self._first:       # from now on: deltas, in the smaller payload buffers
            self._first = False
            self.flush()      # (the pool-sized buffers are released)
            self._C = self.delta_capacity
            self.send = torch.zeros(2 + 3 * (self._C + 1), **self._i32)
            self.recv2 = [torch.zeros(self.world * (2 + 3 * (self._C + 1)), **self._i32) for _ in range(2)]

    def result(self):
        """List (one per rank) of structured BLOCK_DTYPE arrays, sorted by block position."""
        self.flush()
        return [np.ascontiguousarray(rep.cpu().numpy()).reshape(-1).view(BLOCK_DTYPE) for rep in self._replica]


def blocks_in_bounds(blocks, bounds, voxel_size):
    """Which directory entries a Query with `bounds` selects: the reference scales the box to voxel
    units with a C cast to short (BoundingCube::Scale, utils/tsdf/voxel_tsdf.cuh:19-34) and takes a
    block iff all of its 8^3 voxels lie inside, bounds inclusive (check_bound_kernel,
    utils/tsdf/voxel_tsdf.cu:15-26)."""
    scale = np.float32(1.0 / voxel_size)
    b = [int(np.trunc(np.float32(v) * scale)) for v in bounds]   # (xmin, xmax, ymin, ymax, zmin, zmax)
    b = [((v + 32768) % 65536) - 32768 for v in b]               # wrap like the cast to short
    gx = blocks["x"].astype(np.int64) * 8
    gy = blocks["y"].astype(np.int64) * 8
    gz = blocks["z"].astype(np.int64) * 8
    return ((gx >= b[0]) & (gx + 7 <= b[1]) & (gy >= b[2]) & (gy + 7 <= b[3]) & (gz >= b[4]) &
            (gz + 7 <= b[5]))


def query(engine, bounds, per_rank, device=None):
    """TSDFSystem::Query (modules/tsdf_module.cc:39-43) across the ranks of a sharded map: every rank
    gets the records (VOXEL_TSDF_DTYPE) of all blocks inside `bounds`, whichever rank holds them.

    `per_rank` is the all-gathered directory (DirectoryExchange.result()): it tells every rank,
    without communication, how many records each rank will contribute -- so buffers are sized
    exactly, ranks that own nothing inside the box send nothing, and a box owned by a single rank
    needs one broadcast instead of an all-gather.  Record order: by rank, then the reference's order
    (ascending hash-entry index) within a rank."""
    import torch
    import torch.distributed as dist
    from ._abi import VOXEL_TSDF_DTYPE
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    nrec = [int(blocks_in_bounds(b, bounds, engine.voxel_size).sum()) * 512 for b in per_rank]
    owners = [r for r in range(world) if nrec[r] > 0]
    mine = engine.query(bounds) if nrec[rank] > 0 else np.empty(0, dtype=VOXEL_TSDF_DTYPE)
    if len(mine) != nrec[rank]:
        raise RuntimeError(f"rank {rank}: directory is stale ({len(mine)} records, directory says {nrec[rank]})")
    if world == 1 or not owners:
        return mine
    dev = device if device is not None else "cpu"
    if len(owners) == 1:
        src = owners[0]
        buf = torch.empty(nrec[src] * 4, dtype=torch.float32, device=dev)
        if rank == src:
            buf.copy_(torch.from_numpy(mine.view(np.float32).copy()))
        dist.broadcast(buf, src=src)
        return buf.cpu().numpy().view(VOXEL_TSDF_DTYPE)
    width = max(nrec) * 4
    send = torch.zeros(width, dtype=torch.float32, device=dev)
    if len(mine):
        send[:len(mine) * 4].copy_(torch.from_numpy(mine.view(np.float32).copy()))
    recv = torch.empty(world * width, dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(recv, send)
    rows = recv.cpu().numpy().reshape(world, width)
    return np.concatenate([rows[r, :nrec[r] * 4] for r in range(world)]).view(VOXEL_TSDF_DTYPE)


def download_all(engine, device=None):
    """TSDFSystem::DownloadAll (modules/tsdf_module.cc:57-64) across the ranks of a sharded map: every
    rank gets the 20-byte records (VOXEL_SEGM_DTYPE, GatherValidSemantic) of ALL ranks' blocks,
    concatenated by rank, the reference's order (ascending hash-entry index) within a rank.  Exact by
    construction: blocks are disjoint between ranks and a record depends on its own block only.
    (Ray casting and marching cubes read NEIGHBOUR blocks -- voxel_tsdf.cu:278-374,561-715 -- and need
    the halo plan of DESIGN.md section 6 instead.)"""
    import torch
    import torch.distributed as dist
    from ._abi import VOXEL_SEGM_DTYPE
    mine = engine.gather_valid_semantic()
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return mine
    dev = device if device is not None else "cpu"
    n = torch.tensor([len(mine)], dtype=torch.int64, device=dev)
    ns = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(ns, n)
    ns = [int(v) for v in ns.cpu()]
    width = max(ns) * 5
    send = torch.zeros(max(width, 1), dtype=torch.float32, device=dev)
    if len(mine):
        send[:len(mine) * 5].copy_(torch.from_numpy(mine.view(np.float32).copy()))
    recv = torch.empty(world * max(width, 1), dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(recv, send)
    rows = recv.cpu().numpy().reshape(world, max(width, 1))
    return np.concatenate([rows[r, :ns[r] * 5] for r in range(world)]).view(VOXEL_SEGM_DTYPE)


def check_sharded_directories(per_rank, slab_bits=2):
    """Every block sits on its owner and on no other rank; returns the global block count."""
    world = len(per_rank)
    seen = set()
    for r, blocks in enumerate(per_rank):
        own = owner_of(blocks["x"], world, slab_bits)
        if not np.all(own == r):
            raise AssertionError(f"rank {r} holds blocks it does not own")
        for p in zip(blocks["x"].tolist(), blocks["y"].tolist(), blocks["z"].tolist()):
            if p in seen:
                raise AssertionError(f"block {p} present on two ranks")
            seen.add(p)
    return len(seen)


# ---- consumers that read NEIGHBOUR blocks, across subvolume seams ------------------------------------------------
def _pos_set(blocks):
    return set(zip(blocks["x"].tolist(), blocks["y"].tolist(), blocks["z"].tolist()))


def halo_plan(per_rank):
    """Which blocks every rank needs from every other rank to mesh its own subvolume.

    Marching cubes on block B reads B + {0, 1}^3 (GatherValidMesh, voxel_tsdf.cu:582-620: the 2x2x2 block
    neighbourhood behind the 16^3 tile).  Ownership is by x-slab, so a neighbour that belongs to someone else has
    x + 1: rank r needs every block (x + 1, y + dy, z + dz), dy, dz in {0, 1}, of its own blocks (x, y, z) that
    exists in another rank's directory.  Computed from the replicated directories alone, hence the same on every
    rank: nobody has to ask.  Returns plan[q][r] = sorted list of positions rank q sends to rank r."""
    world = len(per_rank)
    sets = [_pos_set(b) for b in per_rank]
    plan = [[[] for _ in range(world)] for _ in range(world)]
    for r in range(world):
        want = set()
        for (x, y, z) in sets[r]:
            for dy in (0, 1):
                for dz in (0, 1):
                    want.add((x + 1, y + dy, z + dz))
        want -= sets[r]
        for q in range(world):
            if q != r:
                plan[q][r] = sorted(want & sets[q])
    return plan


def export_blocks(engine, positions):
    """(positions [n, 3] int16, tsdf, rgbw, prob [n, 512]) of the listed blocks of this engine's map"""
    from ._abi import RGBW_DTYPE
    _, blocks = engine.dump_directory()
    idx_of = {(int(x), int(y), int(z)): int(i) for x, y, z, i in zip(blocks["x"], blocks["y"], blocks["z"], blocks["idx"])}
    pos = np.array(positions, dtype=np.int16).reshape(-1, 3)
    if len(pos) == 0:
        return (pos, np.zeros((0, 512), np.float32), np.zeros((0, 512), RGBW_DTYPE), np.zeros((0, 512), np.float32))
    t, c, p = engine.dump_voxels(np.array([idx_of[tuple(int(v) for v in q)] for q in pos], dtype=np.int32))
    return pos, t, c, p


def _gather_block_data(part, nmax, world, device):
    """all-gather of the voxel data of up to `nmax` blocks per rank (an export_blocks() tuple; tsdf | rgbw | prob as
    3 x 512 int32 words per block): list of [nmax, 1536] int32 numpy arrays, one per rank.  One
    all_gather_into_tensor on `device` (cuda under "nccl"; None = CPU tensors for gloo)."""
    import torch
    import torch.distributed as dist
    pos, t, c, p = part
    send = np.zeros((nmax, 3 * 512), dtype=np.int32)
    if len(pos):
        send[:len(pos), 0:512] = t.view(np.int32)
        send[:len(pos), 512:1024] = np.ascontiguousarray(c).view(np.int32).reshape(-1, 512)
        send[:len(pos), 1024:1536] = p.view(np.int32)
    if world == 1:
        return [send]
    dev = device if device is not None else "cpu"
    s_t = torch.from_numpy(send).to(dev).reshape(-1)
    r_t = torch.empty(world * s_t.numel(), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(r_t, s_t)
    got = r_t.cpu().numpy().reshape(world, nmax, 3 * 512)
    return [got[q] for q in range(world)]


def _gather_arrays(arrays, world, device):
    """all-gather of a few flat numpy arrays whose lengths differ between ranks: list over ranks of lists of arrays
    (same dtypes as `arrays`).  Two collectives (byte lengths, padded payload) on `device`."""
    import torch
    import torch.distributed as dist
    dev = device if device is not None else "cpu"
    raw = [np.ascontiguousarray(a).reshape(-1).view(np.uint8) for a in arrays]
    n = torch.tensor([len(r) for r in raw], dtype=torch.int64, device=dev)
    ns = torch.zeros(world * len(raw), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(ns, n)
    ns = ns.cpu().numpy().reshape(world, len(raw))
    width = max(int(ns.sum(axis=1).max()), 1)
    send = torch.zeros(width, dtype=torch.uint8, device=dev)
    flat = np.concatenate(raw)
    if len(flat):
        send[:len(flat)].copy_(torch.from_numpy(flat.copy()))
    recv = torch.empty(world * width, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(recv, send)
    rows = recv.cpu().numpy().reshape(world, width)
    out = []
    for q in range(world):
        o, parts = 0, []
        for a, m in zip(arrays, ns[q]):
            parts.append(rows[q, o:o + int(m)].copy().view(np.asarray(a).dtype))
            o += int(m)
        out.append(parts)
    return out


# ---------------------------------------------------------------------------------------------------------------
# The same exchange with the voxel data staying in device memory (ratsdf_export_blocks_device /
# ratsdf_import_blocks_device): under RCCL the all-gather's buffers are device tensors anyway, so the owner's engine
# writes its seam / frustum blocks straight into the send buffer and the receiver's scratch engine reads the rows it
# needs straight out of the receive buffer -- 6 KiB per block that never visit host memory (the host path above is
# what the gloo tests and the CPU oracle use).  Records: 1536 int32 words per block, tsdf | rgbw | prob.
# ---------------------------------------------------------------------------------------------------------------
def device_exchange(engine, device):
    """True when the across-shard exports can keep voxel data on the device: HIP engine + a cuda device for the
    collectives' buffers"""
    return device is not None and str(device).startswith("cuda") and engine.lib.backend().startswith("hip")


def _pos_tensor(positions, device):
    import torch
    return torch.from_numpy(np.ascontiguousarray(np.array(positions, dtype=np.int16).reshape(-1, 3))).to(device)


def export_blocks_device(engine, positions, rows, device):
    """[rows, 1536] int32 tensor on `device`: record i = the voxels of block positions[i] of the engine's map (rows
    beyond len(positions) are zero).  Every listed block must exist."""
    import torch
    n = len(positions)
    out = torch.zeros((max(rows, n, 1), 1536), dtype=torch.int32, device=device)
    if n:
        pos = _pos_tensor(positions, device)
        missing = torch.zeros(1, dtype=torch.int32, device=device)
        torch.cuda.synchronize(device)           # (the buffers exist and are zeroed before the engine's stream writes)
        engine.export_blocks_device(n, pos.data_ptr(), out.data_ptr(), missing.data_ptr())
        engine.synchronize()
        if int(missing.item()) != 0:
            raise RuntimeError(f"export_blocks_device: {int(missing.item())} of {n} listed blocks are not in the map")
    return out


def import_blocks_device(scratch, positions, records, chunk=4096):
    """`records` ([n, 1536] int32 device tensor, contiguous) into `scratch` as blocks `positions`"""
    import torch
    n = len(positions)
    if n == 0:
        return
    assert records.is_contiguous() and records.shape[0] >= n and records.shape[1] == 1536
    pos = _pos_tensor(positions, records.device)
    torch.cuda.synchronize(records.device)       # (whatever produced the records has finished)
    for lo in range(0, n, chunk):
        m = min(chunk, n - lo)
        scratch.import_blocks_device(m, pos.data_ptr() + lo * 6, records.data_ptr() + lo * 1536 * 4)


def copy_blocks_device(src, dst, positions, device, chunk=4096):
    """blocks `positions` of engine `src` into engine `dst` (same device), `chunk` blocks (24 MiB) at a time"""
    for lo in range(0, len(positions), chunk):
        part = positions[lo:lo + chunk]
        import_blocks_device(dst, part, export_blocks_device(src, part, len(part), device))


def _all_gather_rows(send, world):
    """[world, rows, 1536]: every rank's send buffer (one all_gather_into_tensor on the buffers' device)"""
    import torch
    import torch.distributed as dist
    if world == 1:
        return send.unsqueeze(0)
    recv = torch.empty((world,) + tuple(send.shape), dtype=send.dtype, device=send.device)
    dist.all_gather_into_tensor(recv.reshape(-1), send.reshape(-1))
    return recv


def _import_plan_rows(scratch, recv, out_lists, plan, rank, skip_self=True):
    """what the plan says rank `rank` needs from every other rank, out of the gathered buffers into `scratch`"""
    import torch
    for q in range(len(out_lists)):
        if (skip_self and q == rank) or not plan[q][rank]:
            continue
        row_of = {pp: i for i, pp in enumerate(out_lists[q])}
        rows = torch.tensor([row_of[pp] for pp in plan[q][rank]], dtype=torch.int64, device=recv.device)
        import_blocks_device(scratch, plan[q][rank], recv[q].index_select(0, rows).contiguous())


def mesh_rank_device(engine, scratch, recv, out_lists, plan, rank, device):
    """One rank's share of mesh_across_shards with the voxel data on the device: own blocks engine -> scratch, halo rows
    recv -> scratch, mesh.  `recv` = _all_gather_rows of every rank's export_blocks_device(out_lists[q])."""
    _, blocks = engine.dump_directory()
    own = list(zip(blocks["x"].tolist(), blocks["y"].tolist(), blocks["z"].tolist()))
    copy_blocks_device(engine, scratch, own, device)
    _import_plan_rows(scratch, recv, out_lists, plan, rank)
    return scratch.gather_valid_mesh()


def raycast_rank_device(engine, scratch, recv, out_lists, plan, rank, device, intrinsics, height, width, pose,
                        max_depth, rows):
    """One rank's strip of raycast_across_shards with the voxel data on the device"""
    copy_blocks_device(engine, scratch, plan[rank][rank], device)
    _import_plan_rows(scratch, recv, out_lists, plan, rank)
    return scratch.raycast_rows(intrinsics, height, width, pose, max_depth, rows[0], rows[1])


def mesh_with_halo(engine, scratch, halo):
    """The mesh of the blocks `engine` owns, with the cells on its subvolume's +x seam closed: `scratch` (a fresh
    engine with the same voxel size and shard parameters) receives the engine's own blocks and the neighbours'
    blocks `halo` (an export_blocks() tuple) through ratsdf_import_blocks and is meshed -- a sharded engine meshes
    only what it owns, so every cell of the whole map is emitted by exactly one rank (the owner of the block that
    holds the cell's minimum corner), and the map itself is not touched."""
    _, blocks = engine.dump_directory()
    own = export_blocks(engine, np.stack([blocks["x"], blocks["y"], blocks["z"]], axis=1))
    for part in (own, halo):
        for lo in range(0, len(part[0]), 4096):
            scratch.import_blocks(part[0][lo:lo + 4096], part[1][lo:lo + 4096], part[2][lo:lo + 4096],
                                  part[3][lo:lo + 4096])
    return scratch.gather_valid_mesh()


def mesh_across_shards(engine, make_scratch, per_rank, device=None):
    """TSDFSystem::DownloadAllMesh (tsdf_module.cc:66-86) of a map that is spread over ranks by block ownership:
    halo exchange (one all-gather of the seam blocks' voxel data, 6 KiB per block), per-rank meshing with the
    seam closed, one all-gather of the meshes.  `per_rank` = the replicated directories
    (DirectoryExchange.result()); `make_scratch()` builds an empty engine like `engine`.  Returns (vertices [n, 3],
    triangles [m, 3], vertex probability [n]) of the whole map on every rank; as a multiset of triangles it is the
    mesh of the same map held by one engine.  `device`: where the collectives' buffers live -- a torch cuda device
    under backend "nccl" (RCCL takes device tensors only), None = CPU tensors (gloo), as multi.query."""
    import torch
    import torch.distributed as dist
    from ._abi import RGBW_DTYPE
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    plan = halo_plan(per_rank)
    # what I send to anybody, once; everybody knows everybody's list, so the buffers have agreed sizes
    out_lists = [sorted(set(p for r in range(world) for p in plan[q][r])) for q in range(world)]
    nmax = max(1, max(len(l) for l in out_lists))
    if device_exchange(engine, device):  # voxel data stays in device memory
        recv = _all_gather_rows(export_blocks_device(engine, out_lists[rank], nmax, device), world)
        scratch = make_scratch()
        try:
            v, tri, vp = mesh_rank_device(engine, scratch, recv, out_lists, plan, rank, device)
        finally:
            scratch.close()
        return _merge_meshes(v, tri, vp, world, device)
    recv = _gather_block_data(export_blocks(engine, out_lists[rank]), nmax, world, device)
    hp, ht, hc, hprob = [], [], [], []
    for q in range(world):
        if q == rank or not plan[q][rank]:
            continue
        row_of = {pp: i for i, pp in enumerate(out_lists[q])}
        rows = np.array([row_of[pp] for pp in plan[q][rank]], dtype=np.int64)
        data = recv[q][rows]
        hp.append(np.array(plan[q][rank], dtype=np.int16))
        ht.append(data[:, 0:512].copy().view(np.float32))
        hc.append(data[:, 512:1024].copy().view(RGBW_DTYPE).reshape(-1, 512))
        hprob.append(data[:, 1024:1536].copy().view(np.float32))
    if hp:
        halo = (np.concatenate(hp), np.concatenate(ht), np.concatenate(hc), np.concatenate(hprob))
    else:
        halo = (np.zeros((0, 3), np.int16), np.zeros((0, 512), np.float32), np.zeros((0, 512), RGBW_DTYPE),
                np.zeros((0, 512), np.float32))
    scratch = make_scratch()
    try:
        v, tri, vp = mesh_with_halo(engine, scratch, halo)
    finally:
        scratch.close()
    return _merge_meshes(v, tri, vp, world, device)


def _merge_meshes(v, tri, vp, world, device):
    """every rank's mesh on every rank, vertex indices shifted (one all-gather of the three arrays)"""
    if world == 1:
        return v, tri, vp
    parts = _gather_arrays([np.asarray(v, dtype=np.float32).reshape(-1), np.asarray(tri, dtype=np.int32).reshape(-1),
                            np.asarray(vp, dtype=np.float32).reshape(-1)], world, device)
    parts = [(a.reshape(-1, 3), b.reshape(-1, 3), c_) for a, b, c_ in parts]
    off, vs_, ts_, ps_ = 0, [], [], []
    for (vv, tt, pp) in parts:
        vs_.append(vv)
        ts_.append(tt + off)
        ps_.append(pp)
        off += len(vv)
    return np.concatenate(vs_), np.concatenate(ts_), np.concatenate(ps_)


# ---------------------------------------------------------------------------------------------------------------
# Ray casting across shards (TSDFGrid::RayCast, voxel_tsdf.cu:278-374, 885-902).
#
# A ray marches through the map with a step that depends on what it reads on the way (fine steps once tsdf < 0.5),
# so a rank that renders from its own blocks alone -- other ranks' blocks absent, i.e. "far from any surface" --
# samples different positions than one engine holding the whole map would, and compositing per-rank images by depth
# is not exact.  What IS exact: partition the IMAGE.  Rank r renders rows [r0, r1) and needs every block a ray of
# those rows can read; the replicated directories say which blocks those are and who owns them, so nobody has to ask:
# one all-gather carries the blocks' voxel data, each rank imports what it needs into a scratch engine
# (ratsdf_import_blocks), renders there, and one all-gather assembles the strips.  Retrieve() is a lookup by position
# and an absent block reads as the default voxel in both engines, so the strip equals the same rows of the image one
# engine would render -- bit for bit.
# ---------------------------------------------------------------------------------------------------------------
def strip_rows(height, world):
    """image rows [lo, hi) of every rank: equal strips, the last one takes the remainder"""
    per = (height + world - 1) // world
    return [(min(r * per, height), min((r + 1) * per, height)) for r in range(world)]


def _quat_matrix(q):
    x, y, z, w = [float(v) for v in q]
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], dtype=np.float64)


def frustum_blocks(blocks, intrinsics, width, rows, pose, max_depth, voxel_size, margin_voxels=6.0):
    """Mask over directory records: may a ray through image rows [rows[0], rows[1]) read a voxel of the block?

    Conservative (a superset): the block's bounding sphere, grown by `margin_voxels` (the march reads voxels rounded
    from ray positions, the zero crossing's trilinear taps and gradient taps one voxel further out), against the
    four side planes of the strip's frustum (one pixel of slack), the camera plane, and the ray length
    (max_depth + one coarse step, taken as 1 m)."""
    from ._abi import _as_intr, _as_pose
    k, p = _as_intr(intrinsics), _as_pose(pose)
    if len(blocks) == 0:
        return np.zeros(0, dtype=bool)
    c = (np.stack([blocks["x"], blocks["y"], blocks["z"]], axis=1).astype(np.float64) * 8 + 3.5) * voxel_size
    R = _quat_matrix((p.qx, p.qy, p.qz, p.qw))
    pc = c @ R.T + np.array([p.tx, p.ty, p.tz], dtype=np.float64)          # cam_T_world applied to the centres
    rad = (np.sqrt(3.0) * 4.0 + margin_voxels) * voxel_size
    x, y, z = pc[:, 0], pc[:, 1], pc[:, 2]
    umin, umax, vmin, vmax = -1.0, float(width), float(rows[0]) - 1.0, float(rows[1])

    def side(a, b, cc):  # signed distance to the plane a x + b y + cc z = 0 through the camera centre
        return (a * x + b * y + cc * z) / np.sqrt(a * a + b * b + cc * cc)
    keep = z > -rad
    keep &= side(k.fx, 0.0, k.cx - umin) > -rad
    keep &= side(-k.fx, 0.0, umax - k.cx) > -rad
    keep &= side(0.0, k.fy, k.cy - vmin) > -rad
    keep &= side(0.0, -k.fy, vmax - k.cy) > -rad
    keep &= np.sqrt(x * x + y * y + z * z) < float(max_depth) + 1.0 + rad
    return keep


def raycast_plan(per_rank, intrinsics, height, width, pose, max_depth, voxel_size):
    """plan[q][r] = sorted positions of rank q's blocks that rank r needs to render its strip (q == r included).
    From the replicated directories alone: the same on every rank."""
    world = len(per_rank)
    strips = strip_rows(height, world)
    plan = [[[] for _ in range(world)] for _ in range(world)]
    for q in range(world):
        b = per_rank[q]
        for r in range(world):
            if strips[r][0] >= strips[r][1] or len(b) == 0:
                continue
            m = frustum_blocks(b, intrinsics, width, strips[r], pose, max_depth, voxel_size)
            plan[q][r] = sorted(zip(b["x"][m].tolist(), b["y"][m].tolist(), b["z"][m].tolist()))
    return plan


def _import_chunks(scratch, part):
    for lo in range(0, len(part[0]), 4096):
        scratch.import_blocks(part[0][lo:lo + 4096], part[1][lo:lo + 4096], part[2][lo:lo + 4096],
                              part[3][lo:lo + 4096])


def raycast_strip(scratch, parts, intrinsics, height, width, pose, max_depth, rows):
    """rows [rows[0], rows[1]) of TSDFGrid::RayCast rendered on `scratch` (an empty engine of the map's voxel size)
    after importing `parts` (export_blocks() tuples): (rgba, normal) strips."""
    for part in parts:
        _import_chunks(scratch, part)
    return scratch.raycast_rows(intrinsics, height, width, pose, max_depth, rows[0], rows[1])


def raycast_across_shards(engine, make_scratch, per_rank, intrinsics, height, width, pose, max_depth, voxel_size,
                          device=None):
    """TSDFGrid::RayCast of a map spread over ranks by block ownership (comment above): returns the whole
    (rgba, normal) images, H x W x 4 uint8, on every rank -- equal to one engine's rendering of the same map.
    `per_rank` = the replicated directories; `make_scratch()` builds an empty, UNsharded engine; `device` as
    mesh_across_shards (a cuda device under "nccl")."""
    import torch
    import torch.distributed as dist
    from ._abi import RGBW_DTYPE
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    plan = raycast_plan(per_rank, intrinsics, height, width, pose, max_depth, voxel_size)
    strips = strip_rows(height, world)
    # what I send to the others, once; everybody knows everybody's list, so the buffers have agreed sizes
    out_lists = [sorted(set(p for r in range(world) if r != q for p in plan[q][r])) for q in range(world)]
    nmax = max(1, max(len(l) for l in out_lists))
    if device_exchange(engine, device):  # voxel data stays in device memory
        recv = _all_gather_rows(export_blocks_device(engine, out_lists[rank], nmax, device), world)
        scratch = make_scratch()
        try:
            mine = raycast_rank_device(engine, scratch, recv, out_lists, plan, rank, device, intrinsics, height, width,
                                       pose, max_depth, strips[rank])
        finally:
            scratch.close()
        if world == 1:
            return mine
        got = _gather_arrays([mine[0].reshape(-1), mine[1].reshape(-1)], world, device)
        return (np.concatenate([g[0].reshape(-1, width, 4) for g in got], axis=0),
                np.concatenate([g[1].reshape(-1, width, 4) for g in got], axis=0))
    recv = _gather_block_data(export_blocks(engine, out_lists[rank]), nmax, world, device)
    parts = [export_blocks(engine, plan[rank][rank])]
    for q in range(world):
        if q == rank or not plan[q][rank]:
            continue
        row_of = {pp: i for i, pp in enumerate(out_lists[q])}
        rows = np.array([row_of[pp] for pp in plan[q][rank]], dtype=np.int64)
        data = recv[q][rows]
        parts.append((np.array(plan[q][rank], dtype=np.int16), data[:, 0:512].copy().view(np.float32),
                      data[:, 512:1024].copy().view(RGBW_DTYPE).reshape(-1, 512),
                      data[:, 1024:1536].copy().view(np.float32)))
    scratch = make_scratch()
    try:
        mine = raycast_strip(scratch, parts, intrinsics, height, width, pose, max_depth, strips[rank])
    finally:
        scratch.close()
    if world == 1:
        return mine
    got = _gather_arrays([mine[0].reshape(-1), mine[1].reshape(-1)], world, device)
    return (np.concatenate([g[0].reshape(-1, width, 4) for g in got], axis=0),
            np.concatenate([g[1].reshape(-1, width, 4) for g in got], axis=0))
