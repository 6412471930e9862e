"""Multi-GPU host logic: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on
MI355X, "gloo" on CPU for tests).  The reference has no multi-GPU code (single process, single GPU,
modules/tsdf_module.h:152-164); this is the new design of SURVEY 8e:

* block-ownership sharding ("spatial subvolumes"): every rank sees the full frame but inserts only
  the blocks it owns, owner = floormod(block.x >> slab_bits, world).  Integration needs no
  communication.  The per-rank engines are created with shard_rank / shard_count.
* frame-batched streams: every rank integrates its own stream into its own map.
* in both modes the ranks periodically ALL-GATHER their block directories (12-byte entries) so any
  rank can tell which rank holds which block (queries across seams, global statistics).  Messages
  are small (10^3..10^5 entries), so one fixed-capacity all-gather + one count all-gather is used.
"""
import numpy as np

from ._abi import BLOCK_DTYPE


def owner_of(block_x, world, slab_bits=2):
    """Rank owning a block with x block-coordinate `block_x` (numpy-friendly, floor modulo)."""
    return np.mod(np.asarray(block_x, dtype=np.int64) >> slab_bits, world)


class DirectoryExchange:
    """All-gather of block directories with fixed-capacity buffers.

    device=None -> CPU tensors (gloo); otherwise a torch cuda device (nccl/RCCL) and the engine
    writes its directory straight into the send buffer (ratsdf_export_directory_device)."""

    def __init__(self, capacity, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.capacity = int(capacity)
        self.device = device
        kw = dict(dtype=torch.int32, device=device if device is not None else "cpu")
        self.send = torch.zeros(self.capacity * 3, **kw)
        self.count = torch.zeros(1, **kw)
        self.recv = torch.zeros(self.world * self.capacity * 3, **kw)
        self.counts = torch.zeros(self.world, **kw)

    def fill_from_engine_device(self, engine):
        """HIP engine -> send buffer, on the engine's stream (no host round trip)."""
        engine.export_directory_device(self.send.data_ptr(), self.capacity, self.count.data_ptr())

    def fill_from_numpy(self, blocks):
        """blocks: structured array of BLOCK_DTYPE (e.g. Engine.dump_directory()[1])."""
        n = min(len(blocks), self.capacity)
        raw = np.zeros(self.capacity * 3, dtype=np.int32)
        raw[:n * 3] = np.ascontiguousarray(blocks[:n]).view(np.int32).reshape(-1)
        self.send.copy_(self.torch.from_numpy(raw))
        self.count.fill_(n)

    def all_gather(self):
        if self.world == 1:
            self.recv.copy_(self.send)
            self.counts.copy_(self.count)
        else:
            self.dist.all_gather_into_tensor(self.recv, self.send)
            self.dist.all_gather_into_tensor(self.counts, self.count)

    def result(self):
        """List (one per rank) of structured BLOCK_DTYPE arrays."""
        counts = self.counts.cpu().numpy()
        raw = self.recv.cpu().numpy().reshape(self.world, self.capacity * 3)
        out = []
        for r in range(self.world):
            n = int(counts[r])
            out.append(raw[r, :n * 3].copy().view(BLOCK_DTYPE))
        return out


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
