"""ctypes view of the C ABI declared in include/ratsdf.h.

The same ABI is exported by the HIP engine (prefix ``ratsdf_``) and, for tests only, by the CPU
oracle (prefix ``ratsdf_oracle_``).  This module only knows the ABI's shape; which shared library
and prefix to bind is the caller's choice (the product binds libratsdf.so, see ``__init__``).
"""
import ctypes as C

import numpy as np

BLOCK_VOLUME = 512

STATUS = {
    0: "ok",
    1: "bad argument",
    2: "device error",
    3: "voxel block pool exhausted",
    4: "internal work list overflow",
    5: "no device",
    6: "not implemented",
    7: "in-launch wait timed out",
}


class RatsdfError(RuntimeError):
    def __init__(self, status, what):
        super().__init__(f"{what}: status {status} ({STATUS.get(status, 'unknown')})")
        self.status = status


class Intrinsics(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float)]


class Pose(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("qx", "qy", "qz", "qw", "tx", "ty", "tz")]


class Bounds(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("xmin", "xmax", "ymin", "ymax", "zmin", "zmax")]


class FrameStats(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("visible_blocks", "updated_voxels", "allocated_blocks",
                                         "deleted_blocks", "active_blocks", "slow_requests")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class Config(C.Structure):
    _fields_ = [("voxel_size", C.c_float), ("truncation", C.c_float), ("device", C.c_int32),
                ("block_bits", C.c_int32), ("bucket_bits", C.c_int32), ("shard_rank", C.c_int32),
                ("shard_count", C.c_int32), ("shard_slab_bits", C.c_int32), ("threads", C.c_int32),
                ("reserved", C.c_int32 * 7)]


# numpy views of the POD records (layouts fixed by the reference, see ratsdf.h)
BLOCK_DTYPE = np.dtype([("x", "<i2"), ("y", "<i2"), ("z", "<i2"), ("offset", "<i2"), ("idx", "<i4")])
RGBW_DTYPE = np.dtype([("r", "u1"), ("g", "u1"), ("b", "u1"), ("weight", "u1")])
VOXEL_TSDF_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("tsdf", "<f4")])
VOXEL_SEGM_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("tsdf", "<f4"),
                             ("prob", "<f4")])
assert BLOCK_DTYPE.itemsize == 12 and VOXEL_TSDF_DTYPE.itemsize == 16
assert VOXEL_SEGM_DTYPE.itemsize == 20 and RGBW_DTYPE.itemsize == 4

# every symbol include/ratsdf.h declares (without prefix)
SYMBOLS = [
    "create", "create_ex", "destroy", "integrate", "integrate_device", "integrate_device_batch",
    "prepare_device_batch", "integrate_batch", "host_alloc", "host_free", "synchronize", "recover", "stream",
    "profile_enable", "profile_read", "profile_read_frames", "totals", "pipeline_counters",
    "num_active_blocks", "last_frame_stats", "query", "gather_valid", "gather_valid_semantic",
    "download_all", "free_buffer", "raycast", "raycast_rows", "raycast_device", "gather_valid_mesh", "download_all_mesh",
    "export_directory_device", "export_directory_delta_device", "import_blocks", "export_blocks_device",
    "import_blocks_device", "group_create", "group_destroy", "group_size",
    "group_integrate_device_batch", "group_synchronize", "group_profile_enable", "group_profile_read",
    "test_allocate", "test_delete",
    "test_retrieve", "test_assign_rgbw", "dump_directory", "dump_voxels", "dump_heap",
    "status_string", "backend",
]


class _OwnedBuffer:
    """a result buffer handed over by the library (malloc'ed there): exposes it through the array interface and gives
    it back with ratsdf_free_buffer when the last array built on it is collected"""

    def __init__(self, lib, address, nbytes):
        self._lib, self._address = lib, address
        self.__array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (address, False), "version": 3}

    def __del__(self):
        try:
            if self._address:
                self._lib.fn["free_buffer"](C.c_void_p(self._address))
                self._address = 0
        except Exception:   # (interpreter shutdown: the library object may be gone already)
            pass


def _ptr(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype)) if a is not None else None


class Library:
    """A loaded shared library exporting the ABI under ``prefix``."""

    def __init__(self, path, prefix="ratsdf_"):
        self.path = str(path)
        self.prefix = prefix
        self.dll = C.CDLL(self.path)
        self.fn = {}
        import os
        for s in SYMBOLS:
            try:
                self.fn[s] = getattr(self.dll, prefix + s)  # AttributeError = missing export
            except AttributeError:
                # (same-box A/B against engine builds of earlier commits, tools/build_variant.sh: entry points
                # added since are simply absent there; never set for the product library)
                if os.environ.get("RATSDF_LIB_VARIANT") != "1":
                    raise
                self.fn[s] = C.CFUNCTYPE(C.c_int)(lambda *a: 6)
        for s in SYMBOLS:
            self.fn[s].restype = C.c_int
        self.fn["status_string"].restype = C.c_char_p
        self.fn["backend"].restype = C.c_char_p
        vp = C.c_void_p
        self.fn["create"].argtypes = [C.c_float, C.c_float, C.c_int, C.POINTER(vp)]
        self.fn["create_ex"].argtypes = [C.POINTER(Config), C.POINTER(vp)]
        self.fn["destroy"].argtypes = [vp]
        self.fn["integrate"].argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_float,
                                         C.POINTER(Intrinsics), C.POINTER(Pose)]
        self.fn["integrate_device"].argtypes = self.fn["integrate"].argtypes
        self.fn["integrate_device_batch"].argtypes = [vp, C.c_int, vp, vp, vp, vp, C.c_int, C.c_int,
                                                      C.c_float, vp, vp]
        self.fn["prepare_device_batch"].argtypes = [vp, C.c_int, C.c_int, C.c_int]
        self.fn["integrate_batch"].argtypes = [vp, C.c_int, vp, vp, vp, vp, C.c_int, C.c_int, C.c_float,
                                               vp, vp, C.c_int]
        self.fn["host_alloc"].argtypes = [C.c_size_t, C.POINTER(vp)]
        self.fn["host_free"].argtypes = [vp]
        self.fn["synchronize"].argtypes = [vp]
        self.fn["recover"].argtypes = [vp]
        self.fn["stream"].argtypes = [vp, C.POINTER(vp)]
        self.fn["profile_enable"].argtypes = [vp, C.c_int]
        self.fn["profile_read"].argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        self.fn["profile_read_frames"].argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int,
                                                    C.POINTER(C.c_int)]
        self.fn["totals"].argtypes = [vp, C.POINTER(C.c_int64), C.c_int]
        self.fn["pipeline_counters"].argtypes = [vp, C.POINTER(C.c_int64), C.c_int]
        self.fn["num_active_blocks"].argtypes = [vp, C.POINTER(C.c_int32)]
        self.fn["last_frame_stats"].argtypes = [vp, C.POINTER(FrameStats)]
        self.fn["query"].argtypes = [vp, C.POINTER(Bounds), C.POINTER(vp), C.POINTER(C.c_size_t)]
        self.fn["gather_valid"].argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
        self.fn["gather_valid_semantic"].argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
        self.fn["download_all"].argtypes = [vp, C.c_char_p]
        self.fn["free_buffer"].argtypes = [vp]
        self.fn["raycast"].argtypes = [vp, C.POINTER(Intrinsics), C.c_int, C.c_int, C.POINTER(Pose),
                                       C.c_float, vp, vp]
        self.fn["raycast_device"].argtypes = self.fn["raycast"].argtypes
        self.fn["raycast_rows"].argtypes = [vp, C.POINTER(Intrinsics), C.c_int, C.c_int, C.POINTER(Pose),
                                            C.c_float, C.c_int, C.c_int, vp, vp]
        self.fn["gather_valid_mesh"].argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t),
                                                 C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(vp)]
        self.fn["download_all_mesh"].argtypes = [vp, C.c_char_p, C.c_char_p, C.c_char_p]
        self.fn["export_directory_device"].argtypes = [vp, vp, C.c_int32, vp]
        self.fn["export_directory_delta_device"].argtypes = [vp, vp, C.c_int32, vp]
        self.fn["import_blocks"].argtypes = [vp, C.c_int32, vp, vp, vp, vp]
        self.fn["export_blocks_device"].argtypes = [vp, C.c_int32, vp, vp, vp]
        self.fn["import_blocks_device"].argtypes = [vp, C.c_int32, vp, vp]
        self.fn["group_create"].argtypes = [vp, C.c_int, C.POINTER(vp)]
        self.fn["group_destroy"].argtypes = [vp]
        self.fn["group_size"].argtypes = [vp, C.POINTER(C.c_int32)]
        self.fn["group_integrate_device_batch"].argtypes = [vp, C.c_int, vp, vp, vp, vp, C.c_int,
                                                            C.c_int, C.c_float, vp, vp]
        self.fn["group_synchronize"].argtypes = [vp]
        self.fn["group_profile_enable"].argtypes = [vp, C.c_int]
        self.fn["group_profile_read"].argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        self.fn["test_allocate"].argtypes = [vp, vp, C.c_int32]
        self.fn["test_delete"].argtypes = [vp, vp, C.c_int32]
        self.fn["test_retrieve"].argtypes = [vp, vp, C.c_int32, vp, vp, vp, vp]
        self.fn["test_assign_rgbw"].argtypes = [vp, vp, vp, C.c_int32]
        self.fn["dump_directory"].argtypes = [vp, C.POINTER(vp), C.POINTER(vp),
                                              C.POINTER(C.c_size_t)]
        self.fn["dump_voxels"].argtypes = [vp, vp, C.c_int32, vp, vp, vp]
        self.fn["dump_heap"].argtypes = [vp, C.POINTER(C.c_int32), vp]
        self.fn["status_string"].argtypes = [C.c_int]
        self.fn["backend"].argtypes = []

    def backend(self):
        return self.fn["backend"]().decode()


def _check(st, what):
    if st != 0:
        raise RatsdfError(st, what)


def _as_pose(pose):
    if isinstance(pose, Pose):
        return pose
    q = [float(v) for v in pose]
    if len(q) != 7:
        raise ValueError("pose must be (qx, qy, qz, qw, tx, ty, tz)")
    return Pose(*q)


def _as_intr(k):
    if isinstance(k, Intrinsics):
        return k
    return Intrinsics(*[float(v) for v in k])


class Engine:
    """One TSDF map (the reference's ``TSDFGrid``, utils/tsdf/voxel_tsdf.cuh:39-145)."""

    def __init__(self, lib, voxel_size, truncation, device=0, block_bits=0, bucket_bits=0,
                 shard_rank=0, shard_count=1, shard_slab_bits=0, threads=0):
        self.lib = lib
        self.voxel_size = float(voxel_size)
        self.truncation = float(truncation)
        cfg = Config()
        cfg.voxel_size = voxel_size
        cfg.truncation = truncation
        cfg.device = device
        cfg.block_bits = block_bits
        cfg.bucket_bits = bucket_bits
        cfg.shard_rank = shard_rank
        cfg.shard_count = shard_count
        cfg.shard_slab_bits = shard_slab_bits
        cfg.threads = threads
        self.block_bits = block_bits or 18
        self.bucket_bits = bucket_bits or 21
        h = C.c_void_p()
        _check(lib.fn["create_ex"](C.byref(cfg), C.byref(h)), "create")
        self._h = h

    # -- lifetime --------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self.lib.fn["destroy"](self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- hot path --------------------------------------------------------------------------
    def integrate(self, rgb, depth, ht, lt, max_depth, intrinsics, pose):
        """TSDFGrid::Integrate (voxel_tsdf.cu:416-452) on host numpy images."""
        depth = np.ascontiguousarray(depth, dtype=np.float32)
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        h, w = depth.shape
        if rgb.shape != (h, w, 3):
            raise ValueError("rgb must be HxWx3 uint8 matching depth")
        if ht is not None:
            ht = np.ascontiguousarray(ht, dtype=np.float32)
        if lt is not None:
            lt = np.ascontiguousarray(lt, dtype=np.float32)
        for a in (ht, lt):
            if a is not None and a.shape != (h, w):
                raise ValueError("ht/lt must match depth")
        k, p = _as_intr(intrinsics), _as_pose(pose)
        st = self.lib.fn["integrate"](self._h, rgb.ctypes.data, depth.ctypes.data,
                                      ht.ctypes.data if ht is not None else None,
                                      lt.ctypes.data if lt is not None else None, h, w,
                                      float(max_depth), C.byref(k), C.byref(p))
        _check(st, "integrate")

    def integrate_device(self, d_rgb, d_depth, d_ht, d_lt, height, width, max_depth, intrinsics,
                         pose):
        """Frame already in HBM (raw device pointers as ints); asynchronous."""
        k, p = _as_intr(intrinsics), _as_pose(pose)
        st = self.lib.fn["integrate_device"](self._h, d_rgb, d_depth, d_ht or None, d_lt or None,
                                             height, width, float(max_depth), C.byref(k),
                                             C.byref(p))
        _check(st, "integrate_device")

    def make_batch(self, d_rgb, d_depth, d_ht, d_lt, height, width, max_depth, intrinsics, poses):
        """Pre-marshals n frames (lists of raw device pointers, intrinsics, poses) for
        integrate_device_batch; returns an opaque tuple that can be replayed many times."""
        n = len(d_rgb)
        arr = lambda ptrs: (C.c_void_p * n)(*ptrs) if ptrs is not None else None
        ks = (Intrinsics * n)(*[_as_intr(k) for k in intrinsics])
        ps = (Pose * n)(*[_as_pose(p) for p in poses])
        return (n, arr(d_rgb), arr(d_depth), arr(d_ht), arr(d_lt), int(height), int(width),
                float(max_depth), ks, ps)

    def integrate_device_batch(self, batch):
        n, rgb, depth, ht, lt, h, w, md, ks, ps = batch
        st = self.lib.fn["integrate_device_batch"](self._h, n, rgb, depth, ht, lt, h, w, md, ks, ps)
        _check(st, "integrate_device_batch")

    def prepare_device_batch(self, n, height, width):
        """scratch + HIP graph of an n-frame batch ahead of time (ratsdf_prepare_device_batch); launches nothing"""
        _check(self.lib.fn["prepare_device_batch"](self._h, int(n), int(height), int(width)), "prepare_device_batch")

    def host_alloc(self, shape, dtype):
        """numpy array in page-locked host memory (ratsdf_host_alloc); release with host_free()."""
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        p = C.c_void_p()
        _check(self.lib.fn["host_alloc"](n, C.byref(p)), "host_alloc")
        buf = (C.c_char * n).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = p
        return arr

    def host_free(self, arr):
        p = getattr(self, "_pinned", {}).pop(arr.ctypes.data, None)
        if p is not None:
            _check(self.lib.fn["host_free"](p), "host_free")

    def integrate_batch(self, frames, max_depth, pinned=False):
        """n frames from host memory in one call (ratsdf_integrate_batch).  frames: dicts with rgb,
        depth, ht, lt (numpy; ht / lt may be None), intrinsics, pose.  pinned=True: every image is a
        host_alloc() array (uploaded without a staging copy)."""
        self.integrate_host_batch(self.make_host_batch(frames, max_depth, pinned))

    def make_host_batch(self, frames, max_depth, pinned=False):
        """Pre-marshals n host frames for integrate_host_batch (the pointer tables of ratsdf_integrate_batch take
        ~0.2 ms of Python per 32 frames: a caller that replays the same buffers builds them once).  The arrays the
        pointers refer to are kept alive by the returned object; it reads them again at every replay."""
        n = len(frames)
        keep = []
        def col(key, dtype):
            ptrs = []
            for f in frames:
                a = f.get(key)
                if a is None:
                    ptrs.append(None)
                else:
                    a = np.ascontiguousarray(a, dtype=dtype)
                    keep.append(a)
                    ptrs.append(a.ctypes.data)
            return (C.c_void_p * n)(*ptrs)
        sem = all(f.get("ht") is not None and f.get("lt") is not None for f in frames)
        h, w = frames[0]["depth"].shape
        ks = (Intrinsics * n)(*[_as_intr(f["intrinsics"]) for f in frames])
        ps = (Pose * n)(*[_as_pose(f["pose"]) for f in frames])
        return (n, col("rgb", np.uint8), col("depth", np.float32), col("ht", np.float32) if sem else None,
                col("lt", np.float32) if sem else None, h, w, float(max_depth), ks, ps, 1 if pinned else 0, keep)

    def integrate_host_batch(self, batch):
        n, rgb, depth, ht, lt, h, w, md, ks, ps, pinned, _keep = batch
        _check(self.lib.fn["integrate_batch"](self._h, n, rgb, depth, ht, lt, h, w, md, ks, ps, pinned), "integrate_batch")

    def synchronize(self):
        _check(self.lib.fn["synchronize"](self._h), "synchronize")

    def recover(self):
        """after a sticky error: rebuild everything derived from the block directory, clear the error (ratsdf_recover)"""
        _check(self.lib.fn["recover"](self._h), "recover")

    def stream(self):
        s = C.c_void_p()
        _check(self.lib.fn["stream"](self._h, C.byref(s)), "stream")
        return s.value or 0

    def profile_enable(self, on=True, every_frame=False):
        """HIP events on k_integrate: every 4th frame (sums), or every frame with per-frame records"""
        _check(self.lib.fn["profile_enable"](self._h, (2 if every_frame else 1) if on else 0), "profile_enable")

    def profile_read_frames(self, capacity=60000):
        """(k_integrate us per frame, start-to-start period us per frame) as float32 arrays"""
        k = np.zeros(capacity, dtype=np.float32)
        p = np.zeros(capacity, dtype=np.float32)
        n = C.c_int()
        _check(self.lib.fn["profile_read_frames"](self._h, _ptr(k, C.c_float), _ptr(p, C.c_float), capacity,
                                                  C.byref(n)), "profile_read_frames")
        m = min(n.value, capacity)
        return k[:m], p[:m]

    def profile_read(self):
        """(summed k_integrate milliseconds, launches) since the last read."""
        ms, n = C.c_double(), C.c_int64()
        _check(self.lib.fn["profile_read"](self._h, C.byref(ms), C.byref(n)), "profile_read")
        return ms.value, n.value

    def totals(self, reset=False):
        """dict(frames, visible_blocks, updated_voxels, allocated_blocks, deleted_blocks) sums."""
        t = (C.c_int64 * 5)()
        _check(self.lib.fn["totals"](self._h, t, 1 if reset else 0), "totals")
        return dict(zip(("frames", "visible_blocks", "updated_voxels", "allocated_blocks",
                         "deleted_blocks"), [int(v) for v in t]))

    def pipeline_counters(self, reset=False):
        """dict(front_tail, in_launch, in_launch_resolver, in_launch_general): frames by where their serial
        allocation-order pass ran (HIP engine only)."""
        t = (C.c_int64 * 4)()
        _check(self.lib.fn["pipeline_counters"](self._h, t, 1 if reset else 0), "pipeline_counters")
        return dict(zip(("front_tail", "in_launch", "in_launch_resolver", "in_launch_general"),
                        [int(v) for v in t]))

    def num_active_blocks(self):
        n = C.c_int32()
        _check(self.lib.fn["num_active_blocks"](self._h, C.byref(n)), "num_active_blocks")
        return n.value

    def last_frame_stats(self):
        s = FrameStats()
        _check(self.lib.fn["last_frame_stats"](self._h, C.byref(s)), "last_frame_stats")
        return s.as_dict()

    # -- query side ------------------------------------------------------------------------
    def _take(self, ptr, n, dtype):
        """the library's result buffer as a numpy array that OWNS it (freed with ratsdf_free_buffer when the array and
        its views are gone): no copy -- a GatherValid of the bench map is 41 MB"""
        if n == 0:
            if ptr.value:
                self.lib.fn["free_buffer"](ptr)
            return np.empty(0, dtype=dtype)
        return np.asarray(_OwnedBuffer(self.lib, ptr.value, n * dtype.itemsize)).view(dtype)

    def query(self, bounds):
        """TSDFSystem::Query / TSDFGrid::GatherVoxels (voxel_tsdf.cu:532-559)."""
        b = bounds if isinstance(bounds, Bounds) else Bounds(*[float(v) for v in bounds])
        p, n = C.c_void_p(), C.c_size_t()
        _check(self.lib.fn["query"](self._h, C.byref(b), C.byref(p), C.byref(n)), "query")
        return self._take(p, n.value, VOXEL_TSDF_DTYPE)

    def gather_valid(self):
        p, n = C.c_void_p(), C.c_size_t()
        _check(self.lib.fn["gather_valid"](self._h, C.byref(p), C.byref(n)), "gather_valid")
        return self._take(p, n.value, VOXEL_TSDF_DTYPE)

    def gather_valid_semantic(self):
        p, n = C.c_void_p(), C.c_size_t()
        _check(self.lib.fn["gather_valid_semantic"](self._h, C.byref(p), C.byref(n)),
               "gather_valid_semantic")
        return self._take(p, n.value, VOXEL_SEGM_DTYPE)

    def download_all(self, path):
        _check(self.lib.fn["download_all"](self._h, str(path).encode()), "download_all")

    def raycast(self, intrinsics, height, width, pose, max_depth):
        """TSDFGrid::RayCast (voxel_tsdf.cu:885-902): returns (rgba, normal) HxWx4 uint8 images."""
        k, p = _as_intr(intrinsics), _as_pose(pose)
        rgba = np.zeros((height, width, 4), dtype=np.uint8)
        normal = np.zeros((height, width, 4), dtype=np.uint8)
        _check(self.lib.fn["raycast"](self._h, C.byref(k), height, width, C.byref(p),
                                      float(max_depth), rgba.ctypes.data, normal.ctypes.data),
               "raycast")
        return rgba, normal

    def raycast_device(self, intrinsics, height, width, pose, max_depth, d_rgba, d_normal):
        """raycast() into DEVICE buffers (H x W x 4 uint8 each; either may be 0), asynchronous on the engine's stream"""
        k, p = _as_intr(intrinsics), _as_pose(pose)
        _check(self.lib.fn["raycast_device"](self._h, C.byref(k), height, width, C.byref(p), float(max_depth),
                                             d_rgba or None, d_normal or None), "raycast_device")

    def raycast_rows(self, intrinsics, height, width, pose, max_depth, row0, row1):
        """rows [row0, row1) of raycast(): (rgba, normal), (row1 - row0) x W x 4 uint8 each"""
        k, p = _as_intr(intrinsics), _as_pose(pose)
        n = max(int(row1) - int(row0), 0)
        rgba = np.zeros((n, width, 4), dtype=np.uint8)
        normal = np.zeros((n, width, 4), dtype=np.uint8)
        _check(self.lib.fn["raycast_rows"](self._h, C.byref(k), height, width, C.byref(p), float(max_depth),
                                           int(row0), int(row1), rgba.ctypes.data, normal.ctypes.data), "raycast_rows")
        return rgba, normal

    def gather_valid_mesh(self):
        """TSDFGrid::GatherValidMesh (voxel_tsdf.cu:736-845): (vertices [n,3] f32 metres,
        triangles [m,3] i32, per-vertex probability [n] f32)."""
        pv, pi, pp = C.c_void_p(), C.c_void_p(), C.c_void_p()
        nv, nt = C.c_size_t(), C.c_size_t()
        _check(self.lib.fn["gather_valid_mesh"](self._h, C.byref(pv), C.byref(nv), C.byref(pi),
                                                C.byref(nt), C.byref(pp)), "gather_valid_mesh")
        v = self._take(pv, nv.value * 3, np.dtype("<f4")).reshape(-1, 3)
        i = self._take(pi, nt.value * 3, np.dtype("<i4")).reshape(-1, 3)
        p = self._take(pp, nv.value, np.dtype("<f4"))
        return v, i, p

    def download_all_mesh(self, vertices_path, indices_path, prob_path):
        _check(self.lib.fn["download_all_mesh"](self._h, str(vertices_path).encode(),
                                                str(indices_path).encode(),
                                                str(prob_path).encode()), "download_all_mesh")

    def export_directory_device(self, d_blocks, capacity, d_count):
        _check(self.lib.fn["export_directory_device"](self._h, d_blocks, capacity, d_count),
               "export_directory_device")

    def export_directory_delta_device(self, d_payload, capacity, d_counts):
        """added / changed entries, then deleted positions, since the previous call; d_payload = 0: forget them"""
        _check(self.lib.fn["export_directory_delta_device"](self._h, d_payload or None, capacity, d_counts or None),
               "export_directory_delta_device")

    def import_blocks(self, block_pos, tsdf, rgbw, prob):
        """blocks copied in from another map (a neighbour rank's subvolume): positions [n, 3] int16 and the three
        voxel arrays [n, 512] in dump_voxels()'s layout; inserted whatever the shard filter says"""
        a, n = self._s3(block_pos)
        t = np.ascontiguousarray(tsdf, dtype=np.float32).reshape(n, BLOCK_VOLUME)
        c = np.ascontiguousarray(rgbw, dtype=RGBW_DTYPE).reshape(n, BLOCK_VOLUME)
        p = np.ascontiguousarray(prob, dtype=np.float32).reshape(n, BLOCK_VOLUME)
        _check(self.lib.fn["import_blocks"](self._h, n, a.ctypes.data, t.ctypes.data, c.ctypes.data, p.ctypes.data),
               "import_blocks")

    def export_blocks_device(self, n, d_block_pos, d_voxels, d_missing):
        """record i of d_voxels (1536 words: tsdf | rgbw | prob) = the voxels of the block at position i of d_block_pos
        (n x 3 int16); all three are device pointers; *d_missing (int32) counts listed blocks the map does not hold.
        Asynchronous on the engine's stream."""
        _check(self.lib.fn["export_blocks_device"](self._h, int(n), d_block_pos or None, d_voxels or None, d_missing),
               "export_blocks_device")

    def import_blocks_device(self, n, d_block_pos, d_voxels):
        """import_blocks() from device buffers in export_blocks_device()'s layout (read on the engine's stream)"""
        _check(self.lib.fn["import_blocks_device"](self._h, int(n), d_block_pos or None, d_voxels or None),
               "import_blocks_device")

    # -- test hooks ------------------------------------------------------------------------
    @staticmethod
    def _s3(a):
        a = np.ascontiguousarray(a, dtype=np.int16).reshape(-1, 3)
        return a, a.shape[0]

    def test_allocate(self, block_pos):
        a, n = self._s3(block_pos)
        _check(self.lib.fn["test_allocate"](self._h, a.ctypes.data, n), "test_allocate")

    def test_delete(self, block_pos):
        a, n = self._s3(block_pos)
        _check(self.lib.fn["test_delete"](self._h, a.ctypes.data, n), "test_delete")

    def test_retrieve(self, points):
        a, n = self._s3(points)
        rgbw = np.zeros(n, dtype=RGBW_DTYPE)
        tsdf = np.zeros(n, dtype=np.float32)
        prob = np.zeros(n, dtype=np.float32)
        blocks = np.zeros(n, dtype=BLOCK_DTYPE)
        _check(self.lib.fn["test_retrieve"](self._h, a.ctypes.data, n, rgbw.ctypes.data,
                                            tsdf.ctypes.data, prob.ctypes.data,
                                            blocks.ctypes.data), "test_retrieve")
        return rgbw, tsdf, prob, blocks

    def test_assign_rgbw(self, points, values):
        a, n = self._s3(points)
        v = np.ascontiguousarray(values, dtype=RGBW_DTYPE).reshape(-1)
        if v.shape[0] != n:
            raise ValueError("values must match points")
        _check(self.lib.fn["test_assign_rgbw"](self._h, a.ctypes.data, v.ctypes.data, n),
               "test_assign_rgbw")

    def dump_directory(self):
        pe, pb, n = C.c_void_p(), C.c_void_p(), C.c_size_t()
        _check(self.lib.fn["dump_directory"](self._h, C.byref(pe), C.byref(pb), C.byref(n)),
               "dump_directory")
        return (self._take(pe, n.value, np.dtype("<i4")), self._take(pb, n.value, BLOCK_DTYPE))

    def dump_voxels(self, pool_idx):
        idx = np.ascontiguousarray(pool_idx, dtype=np.int32).reshape(-1)
        n = idx.shape[0]
        tsdf = np.zeros((n, BLOCK_VOLUME), dtype=np.float32)
        rgbw = np.zeros((n, BLOCK_VOLUME), dtype=RGBW_DTYPE)
        prob = np.zeros((n, BLOCK_VOLUME), dtype=np.float32)
        _check(self.lib.fn["dump_voxels"](self._h, idx.ctypes.data, n, tsdf.ctypes.data,
                                          rgbw.ctypes.data, prob.ctypes.data), "dump_voxels")
        return tsdf, rgbw, prob

    def dump_heap(self):
        nf = C.c_int32()
        heap = np.zeros(1 << self.block_bits, dtype=np.int32)
        _check(self.lib.fn["dump_heap"](self._h, C.byref(nf), heap.ctypes.data), "dump_heap")
        return nf.value, heap


class Group:
    """Several engines of one GPU stepped together (ratsdf_group_*): frame f of every member stream
    goes through one launch triple.  Members stay usable on their own between group calls."""

    def __init__(self, engines):
        self.engines = list(engines)
        self.lib = self.engines[0].lib
        arr = (C.c_void_p * len(self.engines))(*[e._h for e in self.engines])
        h = C.c_void_p()
        _check(self.lib.fn["group_create"](arr, len(self.engines), C.byref(h)), "group_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self.lib.fn["group_destroy"](self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def make_batch(self, d_rgb, d_depth, d_ht, d_lt, height, width, max_depth, intrinsics, poses):
        """Arguments are lists over frames of lists over members (raw device pointers, intrinsics,
        poses); returns an opaque tuple for integrate_device_batch."""
        n, s = len(d_rgb), len(self.engines)
        flat = lambda rows: [v for row in rows for v in row]
        arr = lambda rows: (C.c_void_p * (n * s))(*flat(rows)) if rows is not None else None
        ks = (Intrinsics * (n * s))(*[_as_intr(k) for k in flat(intrinsics)])
        ps = (Pose * (n * s))(*[_as_pose(p) for p in flat(poses)])
        return (n, arr(d_rgb), arr(d_depth), arr(d_ht), arr(d_lt), int(height), int(width),
                float(max_depth), ks, ps)

    def integrate_device_batch(self, batch):
        n, rgb, depth, ht, lt, h, w, md, ks, ps = batch
        _check(self.lib.fn["group_integrate_device_batch"](self._h, n, rgb, depth, ht, lt, h, w, md,
                                                           ks, ps), "group_integrate_device_batch")

    def synchronize(self):
        _check(self.lib.fn["group_synchronize"](self._h), "group_synchronize")

    def profile_enable(self, on=True):
        _check(self.lib.fn["group_profile_enable"](self._h, 1 if on else 0), "group_profile_enable")

    def profile_read(self):
        ms, n = C.c_double(), C.c_int64()
        _check(self.lib.fn["group_profile_read"](self._h, C.byref(ms), C.byref(n)),
               "group_profile_read")
        return ms.value, n.value
