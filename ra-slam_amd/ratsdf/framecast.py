"""One camera stream, N subvolume ranks (BASELINE config 4; SURVEY 8e: "each rank receives the full frame ...
broadcast over xGMI").  The reference has one process and one GPU (modules/tsdf_module.h:152-164); what a rank of
the sharded map needs per frame is the input of TSDFGrid::Integrate (utils/tsdf/voxel_tsdf.cu:416-440): the four
images, the pose, the intrinsics, max_depth.  The rank that owns the camera packs them; a `FrameCaster` moves
them to every rank with ONE collective per chunk of frames, on a side stream, ahead of the integration.

Wire format of a chunk of C frames (a flat uint8 buffer; the same bytes on every rank):

    C x frame_stride bytes   frame i = depth f32[H*W] | ht f32[H*W] | lt f32[H*W] | rgb u8[H*W*3]
                             (the order of the engine's staging slot, include/ratsdf.h ratsdf_integrate_batch;
                             semantic=False: depth | rgb, 7 bytes per pixel), padded to a multiple of 64 bytes
    C x 64 bytes             header of frame i = f32 {qx qy qz qw tx ty tz  fx fy cx cy  max_depth},
                             i32 valid, i32 has_semantics, u32 checksum (byte sum of the frame's images), i32 frame number

15 bytes per pixel (4.6 MB at 640x480, 13.8 MB at 1280x720) is what SURVEY 8d counts as the image's algorithmic
bytes; the header adds 64.

Flow control (device path, backend "nccl" = RCCL): a ring of `ring` chunk buffers per rank.
    post()   side stream: wait until the slot's previous reader is through (event recorded by done() on the
             ENGINE's stream), broadcast into the slot, copy the 64-byte headers to page-locked memory, record
             `ready`.  Nothing on the host waits.
    take()   the oldest posted chunk: the host waits for the HEADERS only (they were sent ring-1 chunks ago), the
             engine's stream is made to wait for `ready`; returns pointers / views and the per-frame camera
             parameters -- the arguments of ratsdf_integrate_device_batch.
    done()   records on the engine's stream that the chunk's images have been read.
So the broadcast of chunk c+1 .. c+ring-1 overlaps the integration of chunk c, exactly like
DirectoryDeltaExchange.all_gather orders its collective against the engine's stream (multi.py).
With device=None (gloo, CPU tensors; the tests' oracle engines) the same calls run synchronously.
"""
import numpy as np

HEADER_BYTES = 64


def frame_stride(npix, semantic=True):
    return ((15 if semantic else 7) * npix + 63) // 64 * 64


def chunk_bytes(height, width, chunk, semantic=True):
    return chunk * (frame_stride(height * width, semantic) + HEADER_BYTES)


def pack_chunk(frames, max_depth, height, width, chunk, first_frame_no=0, semantic=True, out=None):
    """The camera side: `frames` (dicts with rgb, depth, ht, lt, intrinsics, pose: ratsdf.synthetic.frame /
    the dataset readers' frames; at most `chunk`, fewer = the stream's tail) as one wire chunk (numpy uint8)."""
    from ._abi import _as_intr, _as_pose
    npix = height * width
    stride = frame_stride(npix, semantic)
    if len(frames) > chunk:
        raise ValueError("more frames than the chunk holds")
    buf = out if out is not None else np.zeros(chunk_bytes(height, width, chunk, semantic), dtype=np.uint8)
    if buf.dtype != np.uint8 or buf.size != chunk_bytes(height, width, chunk, semantic):
        raise ValueError("out must be a flat uint8 array of chunk_bytes()")
    hdr_f = buf[chunk * stride:].view(np.float32).reshape(chunk, 16)
    hdr_i = buf[chunk * stride:].view(np.int32).reshape(chunk, 16)
    hdr_u = buf[chunk * stride:].view(np.uint32).reshape(chunk, 16)
    hdr_i[:] = 0
    for i, f in enumerate(frames):
        img = buf[i * stride:(i + 1) * stride]
        depth = np.ascontiguousarray(f["depth"], dtype=np.float32)
        if depth.shape != (height, width):
            raise ValueError("frame size differs from the caster's")
        img[:npix * 4] = depth.reshape(-1).view(np.uint8)
        o = npix * 4
        sem = semantic and f.get("ht") is not None and f.get("lt") is not None
        if semantic:
            if sem:
                img[o:o + npix * 4] = np.ascontiguousarray(f["ht"], dtype=np.float32).reshape(-1).view(np.uint8)
                img[o + npix * 4:o + npix * 8] = np.ascontiguousarray(f["lt"], dtype=np.float32).reshape(-1).view(np.uint8)
            else:
                img[o:o + npix * 8] = 0
            o += npix * 8
        img[o:o + npix * 3] = np.ascontiguousarray(f["rgb"], dtype=np.uint8).reshape(-1)
        p, k = _as_pose(f["pose"]), _as_intr(f["intrinsics"])
        hdr_f[i, :12] = (p.qx, p.qy, p.qz, p.qw, p.tx, p.ty, p.tz, k.fx, k.fy, k.cx, k.cy, max_depth)
        hdr_i[i, 12] = 1
        hdr_i[i, 13] = 1 if sem else 0
        hdr_u[i, 14] = np.uint32(int(img[:o + npix * 3].sum(dtype=np.uint64)) & 0xFFFFFFFF)
        hdr_i[i, 15] = first_frame_no + i
    return buf


class Chunk:
    """What take() hands out: the frames of one chunk as the engine's entry points want them."""

    def __init__(self, slot, n, height, width, max_depth, poses, intrinsics, has_sem, frame_no, checksum):
        self.slot, self.n, self.height, self.width = slot, n, height, width
        self.max_depth, self.poses, self.intrinsics = max_depth, poses, intrinsics
        self.has_sem, self.frame_no, self.checksum = has_sem, frame_no, checksum
        self.d_rgb = self.d_depth = self.d_ht = self.d_lt = None   # device path: raw pointers per frame
        self.frames = None                                         # host path: numpy views per frame
        self.pose_arr = self.intr_arr = None                       # [n, 7] / [n, 4] float32 copies of the headers


class FrameCaster:
    def __init__(self, height, width, chunk, ring=3, src=0, device=None, semantic=True, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.H, self.W, self.C, self.src, self.device, self.semantic = height, width, int(chunk), src, device, semantic
        self.ring = max(int(ring), 2)
        self.npix = height * width
        self.stride = frame_stride(self.npix, semantic)
        self.nbytes = chunk_bytes(height, width, self.C, semantic)
        dev = device if device is not None else "cpu"
        # the camera's rank broadcasts straight out of the caller's packed tensor: no slot of its own
        self.slots = [None if self.rank == src else torch.zeros(self.nbytes, dtype=torch.uint8, device=dev)
                      for _ in range(self.ring)]
        self._cur = [None] * self.ring                    # the tensor a slot's chunk lives in (src: the caller's)
        self.posted = self.taken = self.finished = 0
        self.bytes_sent = 0
        if device is not None:
            self.side = torch.cuda.Stream(device=device)
            self.ready = [torch.cuda.Event() for _ in range(self.ring)]
            self.free = [None] * self.ring                # recorded by done() on the engine's stream
            self.hdr = [torch.zeros(self.C * HEADER_BYTES, dtype=torch.uint8).pin_memory() for _ in range(self.ring)]
        else:
            self.hdr = [None] * self.ring

    @property
    def frames_ahead(self):
        """how far the broadcasts may run ahead of the integration, in frames"""
        return (self.ring - 1) * self.C

    def can_post(self):
        return self.posted - self.finished < self.ring

    def post(self, packed=None):
        """Enqueue the broadcast of the next chunk.  The camera's rank passes the packed chunk (a flat uint8 tensor
        of chunk_bytes() on `device`; it must stay untouched until the chunk's done()); the others pass nothing."""
        torch, dist = self.torch, self.dist
        if not self.can_post():
            raise RuntimeError("every slot of the ring holds a chunk that has not been consumed (call done())")
        slot = self.posted % self.ring
        if self.rank == self.src:
            if packed is None or packed.numel() != self.nbytes or packed.dtype != torch.uint8:
                raise ValueError("the camera's rank posts a flat uint8 tensor of chunk_bytes()")
            buf = packed
        else:
            buf = self.slots[slot]
        self._cur[slot] = buf
        if self.device is None:
            if self.world > 1:
                dist.broadcast(buf, src=self.src, group=self.group)
        else:
            with torch.cuda.stream(self.side):
                if self.free[slot] is not None:
                    self.side.wait_event(self.free[slot])   # the slot's previous chunk has been integrated
                if dist.is_initialized():   # (a group of one rank too: the collective still runs through RCCL)
                    dist.broadcast(buf, src=self.src, group=self.group)
                self.hdr[slot].copy_(buf[self.C * self.stride:], non_blocking=True)
                self.ready[slot].record(self.side)
        self.posted += 1
        self.bytes_sent += self.nbytes

    def take(self, engine_stream=None, verify=False):
        """The oldest posted chunk.  Device path: `engine_stream` (torch.cuda.ExternalStream of the engine's
        stream) waits for the chunk's broadcast; the host waits for its headers only.  verify=True recomputes
        every frame's byte checksum on this rank (tests, bench's parity phase: it reads the whole chunk)."""
        torch = self.torch
        if self.taken >= self.posted:
            raise RuntimeError("take() without a posted chunk")
        if self.taken != self.finished:
            raise RuntimeError("the previous chunk has not been released (call done())")
        slot = self.taken % self.ring
        buf = self._cur[slot]
        if self.device is None:
            raw = buf[self.C * self.stride:].numpy()
        else:
            self.ready[slot].synchronize()
            if engine_stream is not None:
                engine_stream.wait_event(self.ready[slot])
            raw = self.hdr[slot].numpy()
        hf = raw.view(np.float32).reshape(self.C, 16)
        hi = raw.view(np.int32).reshape(self.C, 16)
        hu = raw.view(np.uint32).reshape(self.C, 16)
        n = int(hi[:, 12].sum())
        if not np.all(hi[:n, 12] == 1):
            raise RuntimeError("chunk header: valid frames are not a prefix")
        md = float(hf[0, 11]) if n else 0.0
        ch = Chunk(slot, n, self.H, self.W, md, [tuple(r) for r in hf[:n, 0:7].tolist()],
                   [tuple(r) for r in hf[:n, 7:11].tolist()], [bool(v) for v in hi[:n, 13].tolist()],
                   hi[:n, 15].tolist(), hu[:n, 14].tolist())
        ch.pose_arr = np.ascontiguousarray(hf[:n, 0:7])     # (copies: the slot's header buffer is reused)
        ch.intr_arr = np.ascontiguousarray(hf[:n, 7:11])
        npix, st = self.npix, self.stride
        rgb_off = npix * (12 if self.semantic else 4)
        if self.device is None:
            a = buf.numpy()
            ch.frames = []
            for i in range(n):
                img = a[i * st:(i + 1) * st]
                f = dict(depth=img[:npix * 4].view(np.float32).reshape(self.H, self.W),
                         rgb=img[rgb_off:rgb_off + npix * 3].reshape(self.H, self.W, 3), ht=None, lt=None,
                         pose=ch.poses[i], intrinsics=ch.intrinsics[i])
                if self.semantic and ch.has_sem[i]:
                    f["ht"] = img[npix * 4:npix * 8].view(np.float32).reshape(self.H, self.W)
                    f["lt"] = img[npix * 8:npix * 12].view(np.float32).reshape(self.H, self.W)
                ch.frames.append(f)
        else:
            off = buf.data_ptr() + np.arange(n, dtype=np.int64) * st
            ch.d_depth = off.tolist()
            ch.d_rgb = (off + rgb_off).tolist()
            sem = self.semantic and all(ch.has_sem)
            ch.d_ht = (off + npix * 4).tolist() if sem else None
            ch.d_lt = (off + npix * 8).tolist() if sem else None
        if verify:
            for i in range(n):
                img = buf[i * st:i * st + rgb_off + npix * 3]
                got = int(img.to(torch.int64).sum().item()) & 0xFFFFFFFF
                if got != ch.checksum[i]:
                    raise RuntimeError(f"rank {self.rank}: frame {ch.frame_no[i]} arrived with byte sum {got}, "
                                       f"the camera's rank sent {ch.checksum[i]}")
        self.taken += 1
        return ch

    def chunk_tensor(self, chunk):
        """the flat uint8 tensor (wire format) a taken, not yet released chunk lives in"""
        return self._cur[chunk.slot]

    def done(self, chunk, engine_stream=None):
        """The frames of `chunk` have been enqueued on the engine's stream (device path) / integrated (host path):
        once that work has run, the slot may receive another chunk."""
        if self.device is not None:
            ev = self.torch.cuda.Event()
            ev.record(engine_stream if engine_stream is not None else self.torch.cuda.current_stream(self.device))
            self.free[chunk.slot] = ev
        self._cur[chunk.slot] = None
        self.finished += 1


def integrate_chunk(engine, chunk, batch_cache=None):
    """One chunk through the engine: ratsdf_integrate_device_batch on the device path (one HIP-graph replay for the
    whole chunk), TSDFGrid::Integrate frame by frame on the host path (the tests' oracle engines)."""
    if chunk.n == 0:
        return
    if chunk.frames is not None:
        for f in chunk.frames:
            engine.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], chunk.max_depth, f["intrinsics"], f["pose"])
        return
    # (the pointer tables of ratsdf_integrate_device_batch straight from the header arrays: per-frame Python objects
    # cost more host time than a 15-frame chunk takes on the GPU)
    import ctypes as C
    from ._abi import Intrinsics, Pose
    n = chunk.n
    arr = lambda ptrs: (C.c_void_p * n)(*ptrs) if ptrs is not None else None
    ks = (Intrinsics * n).from_buffer_copy(chunk.intr_arr.tobytes())
    ps = (Pose * n).from_buffer_copy(chunk.pose_arr.tobytes())
    engine.integrate_device_batch((n, arr(chunk.d_rgb), arr(chunk.d_depth), arr(chunk.d_ht), arr(chunk.d_lt),
                                   int(chunk.height), int(chunk.width), float(chunk.max_depth), ks, ps))
