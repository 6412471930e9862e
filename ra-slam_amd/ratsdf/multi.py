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
    is new).  Wire format per rank: counts {added, deleted}, then `capacity` 12-byte entries
    (the added ones first, then the deleted ones: position only).  result() returns the replicas in
    DirectoryExchange.result()'s form (entries sorted by position), so multi.query() takes either.

    The engine still exports its whole directory into a device buffer per exchange (one small
    kernel); the set differences are torch ops on that device, only the deltas travel."""

    def __init__(self, capacity=None, device=None, engine=None):
        super().__init__(capacity, device, engine)
        torch = self.torch
        dev = device if device is not None else "cpu"
        self.count = torch.zeros(2, dtype=torch.int32, device=dev)            # {added, deleted}
        self.counts = torch.zeros(self.world * 2, dtype=torch.int32, device=dev)
        self.full = torch.zeros(self.capacity * 3, dtype=torch.int32, device=dev)  # export target
        self.full_count = torch.zeros(1, dtype=torch.int32, device=dev)
        self._prev = torch.zeros((0, 3), dtype=torch.int32, device=dev)       # my directory as last sent
        self._replica = [torch.zeros((0, 3), dtype=torch.int32, device=dev) for _ in range(self.world)]
        self.last_sent = (0, 0)

    def _pos_key(self, rows):
        """int64 key of the block position (x, y, z int16 in the first 6 bytes of an entry)."""
        i64 = self.torch.int64
        return ((rows[:, 1].to(i64) & 0xFFFF) << 32) | (rows[:, 0].to(i64) & 0xFFFFFFFF)

    def fill_from_engine_device(self, engine):
        engine.export_directory_device(self.full.data_ptr(), self.capacity, self.full_count.data_ptr())
        self._engine_stream = self.torch.cuda.ExternalStream(engine.stream(), device=self.device)
        self._pending_full = True

    def fill_from_numpy(self, blocks):
        n = len(blocks)
        if n > self.capacity:
            raise OverflowError(f"block directory has {n} entries, exchange capacity is {self.capacity}")
        raw = np.zeros(self.capacity * 3, dtype=np.int32)
        raw[:n * 3] = np.ascontiguousarray(blocks).view(np.int32).reshape(-1)
        self.full.copy_(self.torch.from_numpy(raw))
        self.full_count.fill_(n)
        self._engine_stream = None
        self._pending_full = True

    def _make_delta(self):
        """send / count from (current export) minus (previous export)."""
        torch = self.torch
        n = int(self.full_count.item())
        if n > self.capacity:
            raise OverflowError(f"block directory has {n} entries, exchange capacity is {self.capacity}")
        now = self.full[:n * 3].reshape(n, 3)
        kn, kp = self._pos_key(now), self._pos_key(self._prev)
        order = torch.argsort(kn)
        now, kn = now[order], kn[order]
        if len(kp):
            at = torch.searchsorted(kp, kn).clamp_(max=len(kp) - 1)
            same_pos = kp[at] == kn
            unchanged = same_pos & (self._prev[at, 2] == now[:, 2]) & (self._prev[at, 1] == now[:, 1])
            added = now[~unchanged]
            at2 = torch.searchsorted(kn, kp).clamp_(max=max(len(kn) - 1, 0))
            gone = self._prev[~(kn[at2] == kp)] if len(kn) else self._prev
        else:
            added, gone = now, self._prev
        na, nd = len(added), len(gone)
        if na + nd > self.capacity:
            raise OverflowError(f"directory delta of {na} + {nd} entries, exchange capacity is {self.capacity}")
        self.send.zero_()
        self.send[:na * 3] = added.reshape(-1)
        self.send[na * 3:(na + nd) * 3] = gone.reshape(-1)
        self.count[0] = na
        self.count[1] = nd
        self._prev = now.clone()
        self.last_sent = (na, nd)

    def all_gather(self):
        torch = self.torch
        es = self._engine_stream
        if es is not None:  # engine stream -> (event) -> torch's current stream
            ev = torch.cuda.Event()
            ev.record(es)
            torch.cuda.current_stream(self.device).wait_event(ev)
        self._make_delta()
        if self.world == 1:
            self.recv.copy_(self.send)
            self.counts.copy_(self.count)
        else:
            self.dist.all_gather_into_tensor(self.recv, self.send)
            self.dist.all_gather_into_tensor(self.counts, self.count)
        if es is not None:  # the engine may not overwrite the export buffer before the delta was taken
            ev2 = torch.cuda.Event()
            ev2.record(torch.cuda.current_stream(self.device))
            es.wait_event(ev2)
        counts = self.counts.reshape(self.world, 2).cpu()
        rows = self.recv.reshape(self.world, self.capacity, 3)
        for r in range(self.world):
            na, nd = int(counts[r, 0]), int(counts[r, 1])
            rep = self._replica[r]
            drop = torch.cat([rows[r, :na], rows[r, na:na + nd]])  # replaced and deleted positions
            if len(rep) and len(drop):
                rep = rep[~torch.isin(self._pos_key(rep), self._pos_key(drop))]
            rep = torch.cat([rep, rows[r, :na]])
            self._replica[r] = rep[torch.argsort(self._pos_key(rep))]

    def result(self):
        """List (one per rank) of structured BLOCK_DTYPE arrays, sorted by block position."""
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
