/*
 * ratsdf.h -- C ABI of the MI355X-native voxel-hashed TSDF + semantic integration engine.
 *
 * This is the drop-in boundary for RA-SLAM's TSDF hot path.  The reference exposes the path as an
 * in-process C++ class API (no FFI of its own); every entry point below names the reference
 * interface it replaces (paths relative to the reference tree).  Types are PODs with the
 * reference's exact layouts so buffers can be handed to reference-side consumers unchanged.
 *
 * All functions return a ratsdf_status (0 = OK) and never throw across the boundary.
 * One engine handle = one GPU context + one HIP stream; calls on one handle must be serialised by
 * the caller (the reference's TSDFSystem does so with mtx_read_, modules/tsdf_module.h:152-164);
 * distinct handles are independent.
 *
 * The same ABI, with the prefix ratsdf_oracle_ instead of ratsdf_, is implemented by the CPU
 * restatement under oracle/ (test infrastructure only).
 */
#ifndef RATSDF_H_
#define RATSDF_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- constants of the reference (defaults; create_ex can override the two *_BITS) ---------- */
#define RATSDF_BLOCK_LEN 8            /* utils/tsdf/voxel_mem.cuh:17-22  BLOCK_LEN            */
#define RATSDF_BLOCK_VOLUME 512       /* utils/tsdf/voxel_mem.cuh:22     BLOCK_VOLUME         */
#define RATSDF_DEFAULT_BLOCK_BITS 18  /* utils/tsdf/voxel_mem.cuh:11-13  NUM_BLOCK = 2^18     */
#define RATSDF_DEFAULT_BUCKET_BITS 21 /* utils/tsdf/voxel_hash.cuh:13-15 NUM_BUCKET = 2^21    */
#define RATSDF_ENTRIES_PER_BUCKET 2   /* utils/tsdf/voxel_hash.cuh:18-20                      */

typedef enum ratsdf_status {
  RATSDF_OK = 0,
  RATSDF_ERR_BAD_ARGUMENT = 1,   /* reference: assert()s in TSDFGrid::Integrate, voxel_tsdf.cu:419-428;
                                    also: NaN / inf in a pose, intrinsics or max_depth (the reference would
                                    integrate garbage: every voxel picks pixel (0, 0))                     */
  RATSDF_ERR_DEVICE = 2,         /* reference: CUDA_SAFE_CALL prints in debug builds, errors.cuh:13-20  */
  RATSDF_ERR_POOL_EXHAUSTED = 3, /* reference: device assert(idx >= 1), voxel_mem.cu:39                 */
  RATSDF_ERR_CAPACITY = 4,       /* an internal work list overflowed (no reference counterpart)        */
  RATSDF_ERR_NO_DEVICE = 5,      /* HIP engine only: no gfx950 device / runtime available              */
  RATSDF_ERR_NOT_IMPLEMENTED = 6,
  RATSDF_ERR_TIMEOUT = 7         /* an in-launch wait between workgroups expired (sticky until
                                    ratsdf_recover; a workgroup skipped its share of a frame)         */
} ratsdf_status;

/* CameraIntrinsics<float>, utils/cuda/camera.cuh:13-52 */
typedef struct ratsdf_intrinsics {
  float fx, fy, cx, cy;
} ratsdf_intrinsics;

/* SE3<float> = Eigen quaternion + translation, utils/cuda/lie_group.cuh:8-45.  The reference turns
 * the 4x4 pose matrix into a quaternion on the HOST (lie_group.cuh:15-19) before any kernel runs,
 * so the pose crosses the ABI as (q, t).  Maps world -> camera (cam_T_world). */
typedef struct ratsdf_pose {
  float qx, qy, qz, qw;
  float tx, ty, tz;
} ratsdf_pose;

/* BoundingCube<float>, utils/tsdf/voxel_tsdf.cuh:19-34 (member order preserved) */
typedef struct ratsdf_bounds {
  float xmin, xmax, ymin, ymax, zmin, zmax;
} ratsdf_bounds;

/* VoxelSpatialTSDF (16 B), utils/tsdf/voxel_types.cuh:46-56 */
typedef struct ratsdf_voxel_tsdf {
  float x, y, z;
  float tsdf;
} ratsdf_voxel_tsdf;

/* VoxelSpatialTSDFSEGM (20 B), utils/tsdf/voxel_types.cuh:61-70; also the DownloadAll file record,
 * modules/tsdf_module.cc:57-64 */
typedef struct ratsdf_voxel_segm {
  float x, y, z;
  float tsdf;
  float prob;
} ratsdf_voxel_segm;

/* VoxelBlock (12 B), utils/tsdf/voxel_mem.cuh:75-95: block position on the integer block grid,
 * chain offset (0 = tail / normal entry, >0 = distance in entries to next chain element,
 * <0 only in lookup results = "known absent"), pool index (-1 = empty). */
typedef struct ratsdf_block {
  int16_t x, y, z;
  int16_t offset;
  int32_t idx;
} ratsdf_block;

/* VoxelRGBW (4 B), utils/tsdf/voxel_types.cuh:10-19 */
typedef struct ratsdf_rgbw {
  uint8_t r, g, b, weight;
} ratsdf_rgbw;

/* Per-frame counters (the reference only logs active-block counts, voxel_tsdf.cu:442,451,865). */
typedef struct ratsdf_frame_stats {
  int32_t visible_blocks;   /* V: length of the visible list (GatherVisible, voxel_tsdf.cu:465-474) */
  int32_t updated_voxels;   /* U: voxels that passed every test in tsdf_integrate_kernel            */
  int32_t allocated_blocks; /* blocks inserted by the allocation pass                               */
  int32_t deleted_blocks;   /* blocks removed by space carving                                      */
  int32_t active_blocks;    /* NumActiveBlock() after the frame, voxel_hash.cu:225                  */
  int32_t slow_requests;    /* allocation requests that went through the chained-bucket resolver    */
} ratsdf_frame_stats;

typedef struct ratsdf_config {
  float voxel_size;     /* TSDFGrid ctor arg, voxel_tsdf.cuh:47 */
  float truncation;     /* TSDFGrid ctor arg, voxel_tsdf.cuh:47 */
  int32_t device;       /* HIP device ordinal (ignored by the oracle) */
  int32_t block_bits;   /* 0 -> RATSDF_DEFAULT_BLOCK_BITS  */
  int32_t bucket_bits;  /* 0 -> RATSDF_DEFAULT_BUCKET_BITS */
  /* Block-ownership sharding across GPUs (no reference counterpart): a candidate block
   * is inserted only if owner(block) == shard_rank, owner = floormod(x >> shard_slab_bits, count).
   * shard_count <= 1 disables the filter. */
  int32_t shard_rank;
  int32_t shard_count;
  int32_t shard_slab_bits;
  int32_t threads;      /* oracle only: worker threads for the multithreaded CPU baseline (0/1 = serial) */
  int32_t reserved[7];
} ratsdf_config;

typedef struct ratsdf_engine ratsdf_engine;

/* ---- lifetime ------------------------------------------------------------------------------ */
/* TSDFGrid::TSDFGrid(voxel_size, truncation) + VoxelHashTable/VoxelMemPool ctors,
 * voxel_tsdf.cu:376-395, voxel_hash.cu:25-33, voxel_mem.cu:13-27 */
int ratsdf_create(float voxel_size, float truncation, int device, ratsdf_engine** out);
int ratsdf_create_ex(const ratsdf_config* cfg, ratsdf_engine** out);
/* TSDFGrid::~TSDFGrid, voxel_tsdf.cu:397-414 */
int ratsdf_destroy(ratsdf_engine* e);

/* ---- the hot path -------------------------------------------------------------------------- */
/* TSDFGrid::Integrate(img_rgb, img_depth, img_ht, img_lt, max_depth, intrinsics, cam_T_world),
 * voxel_tsdf.cu:416-452.  Host buffers: rgb = H*W*3 u8 (RGB order), depth/ht/lt = H*W f32
 * (metres / probabilities).  ht == NULL or lt == NULL means all-ones images, as
 * TSDFSystem::Integrate substitutes (modules/tsdf_module.cc:27-31).  Caller buffers are not
 * retained after return (they are copied into the engine's page-locked staging ring before the call
 * returns).  The HIP engine does NOT wait for the frame: upload and kernels are enqueued, and a device
 * error of this frame is reported by the next entry point that synchronises (ratsdf_synchronize, every
 * query / statistics call) -- the reference ends Integrate with a stream synchronisation
 * (voxel_tsdf.cu:450) but returns nothing, so its callers (examples/tsdf/offline.cc:169,
 * examples/scannet_evaluation/eval_one.cc:75) cannot tell the difference.  RATSDF_SYNC_INTEGRATE=1 in the
 * environment restores the wait; the oracle integrates synchronously. */
int ratsdf_integrate(ratsdf_engine* e, const uint8_t* rgb, const float* depth, const float* ht,
                     const float* lt, int height, int width, float max_depth,
                     const ratsdf_intrinsics* intrinsics, const ratsdf_pose* cam_T_world);

/* Same frame, inputs already resident in device memory (HBM); enqueues on the engine's stream and
 * returns without a host synchronisation.  This removes the reference's PCIe copies and its three
 * mid-frame host syncs (voxel_tsdf.cu:433-451, 862-864).  HIP engine only. */
int ratsdf_integrate_device(ratsdf_engine* e, const void* d_rgb, const void* d_depth,
                            const void* d_ht, const void* d_lt, int height, int width,
                            float max_depth, const ratsdf_intrinsics* intrinsics,
                            const ratsdf_pose* cam_T_world);
/* n consecutive frames of one stream, all inputs resident in HBM: d_rgb/d_depth/d_ht/d_lt are host
 * arrays of n device pointers (d_ht / d_lt may be NULL = no semantics), intrinsics and cam_T_world
 * are host arrays of n elements.  Equivalent to n calls of ratsdf_integrate_device in order (what the
 * reference's TSDFSystem worker does with its queue, modules/tsdf_module.cc:88-115), without the
 * per-call cost of crossing a language boundary.  HIP engine only. */
int ratsdf_integrate_device_batch(ratsdf_engine* e, int n, const void* const* d_rgb,
                                  const void* const* d_depth, const void* const* d_ht,
                                  const void* const* d_lt, int height, int width, float max_depth,
                                  const ratsdf_intrinsics* intrinsics,
                                  const ratsdf_pose* cam_T_world);
/* Optional: builds ahead of time what the first ratsdf_integrate_device_batch(n, height, width) would build on
 * the spot (image-sized scratch, the HIP graph of an n-frame batch), so that the first batch of a caller with a
 * deadline costs what every later one costs.  Launches nothing.  The reference has no counterpart (its
 * TSDFGrid allocates for 1920x1080 in the constructor, voxel_tsdf.cu:11-13,385-391).  HIP engine only. */
int ratsdf_prepare_device_batch(ratsdf_engine* e, int n, int height, int width);
/* n consecutive frames from HOST memory: the loop of the reference's TSDFSystem worker over its
 * queued inputs (modules/tsdf_module.cc:88-115), as one call.  rgb/depth/ht/lt are host arrays of n
 * host pointers (ht / lt may be NULL = all-ones images); uploads are enqueued ahead of the frames
 * that use them, frame i+1's map-independent part runs while frame i is integrated.  Like
 * ratsdf_integrate the call returns when the caller's buffers are no longer in use -- pageable images have been
 * copied into the engine's staging ring, page-locked ones have been uploaded -- not when the frames have been
 * integrated: the next call's uploads overlap this call's last kernels, and a device error of these frames is
 * reported by the next entry point that synchronises (RATSDF_SYNC_INTEGRATE=1 restores the wait).  `pinned` != 0 says
 * that every image buffer comes from ratsdf_host_alloc (uploaded without a staging copy); a frame
 * whose four images lie side by side in one such block in the order depth | ht | lt | rgb goes up as
 * one copy instead of four (~10 us of copy-engine overhead each), and consecutive frames whose blocks
 * lie side by side at a stride of 16 bytes per pixel (ratsdf::TSDFSystem's queue hands them out so) go
 * up up to four at a time. */
int ratsdf_integrate_batch(ratsdf_engine* e, int n, const uint8_t* const* rgb,
                           const float* const* depth, const float* const* ht,
                           const float* const* lt, int height, int width, float max_depth,
                           const ratsdf_intrinsics* intrinsics, const ratsdf_pose* cam_T_world,
                           int pinned);
/* Page-locked host memory for image buffers (what the reference gets implicitly from
 * cudaMemcpy of cv::Mat data, voxel_tsdf.cu:433-440, only faster): hipHostMalloc / hipHostFree. */
int ratsdf_host_alloc(size_t bytes, void** out);
int ratsdf_host_free(void* p);
/* cudaStreamSynchronize(stream_), voxel_tsdf.cu:450; also surfaces sticky device-side errors
 * (pool exhausted, work-list overflow).  A stream synchronisation and a read of a page-locked flag the kernels raise
 * on error: no device-to-host copy, no launch. */
int ratsdf_synchronize(ratsdf_engine* e);
/* After a sticky error -- RATSDF_ERR_TIMEOUT above all, where a workgroup gave up waiting and skipped its share of a
 * frame -- the engine refuses nothing but reports the error for ever, and the structures derived from the block
 * directory (occupancy bits, the list of live blocks, the free list, claims, frame counters) may no longer agree with
 * it.  ratsdf_recover waits for the GPU, rebuilds all of them from the directory (which is whole: the workgroups that
 * edit it are the ones that were waited for), empties the work lists, tells a consumer of directory deltas to take a
 * whole directory next, and clears the error.  The map keeps every frame before the failed one and whatever part of
 * that one reached the voxels; integration can go on.  HIP engine only (the oracle has no such errors and returns
 * RATSDF_OK); no reference counterpart -- the reference asserts (voxel_mem.cu:39) or hangs. */
int ratsdf_recover(ratsdf_engine* e);
/* Native handle of the engine's stream (hipStream_t) so callers can order their own work / events. */
int ratsdf_stream(ratsdf_engine* e, void** out_stream);

/* Optional in-stream timing of the dominant kernel (k_integrate) with HIP events on the engine's
 * stream: enable, run frames, read back the summed kernel time and the number of timed launches
 * (read synchronises and resets the accumulators).  The reference's counterpart is the wall-clock
 * log line around Integrate (modules/tsdf_module.cc:108-112).  HIP engine only. */
int ratsdf_profile_enable(ratsdf_engine* e, int enable);
int ratsdf_profile_read(ratsdf_engine* e, double* integrate_ms, int64_t* launches);
/* enable = 2 times EVERY frame and keeps per-frame records (for latency distributions: a frame whose
 * view is new costs several times the steady-state frame): k_us[i] = k_integrate time of timed frame
 * i, period_us[i] = start of frame i's k_integrate to the start of frame i+1's (0 for the last frame of
 * a drained run of frames; at most 60 000 frames between reads).  *n = frames recorded; at most
 * `capacity` are copied; the records are cleared. */
int ratsdf_profile_read_frames(ratsdf_engine* e, float* k_us, float* period_us, int capacity, int* n);

/* VoxelHashTable::NumActiveBlock, voxel_hash.cu:225 */
int ratsdf_num_active_blocks(ratsdf_engine* e, int32_t* out);
/* Counters of the most recently completed frame (synchronises). */
int ratsdf_last_frame_stats(ratsdf_engine* e, ratsdf_frame_stats* out);

/* Running sums since creation / the last reset: {frames, sum V, sum U, sum allocated blocks, sum
 * deleted blocks}; used to turn a timed run into algorithmic bytes (15 W H + 12 V + 24 U per frame). */
int ratsdf_totals(ratsdf_engine* e, int64_t* out5, int reset);

/* Which form of a frame's serial allocation-order pass (TSDFGrid::Allocate's AquireBlock order,
 * voxel_tsdf.cu:454-463 + voxel_hash.cu:46-108) the frames took since creation / the last reset -- a
 * property of the MI355X engine's launch layout, no reference counterpart (the oracle reports
 * RATSDF_ERR_NOT_IMPLEMENTED): out4 = {frames whose pass ran at the tail of the frame's first launch,
 * frames whose pass ran beside the voxel update (ordinary path), ... with the chained-bucket resolver,
 * ... through the general path}. */
int ratsdf_pipeline_counters(ratsdf_engine* e, int64_t* out4, int reset);

/* ---- query side ---------------------------------------------------------------------------- */
/* TSDFGrid::GatherVoxels(BoundingCube<float>) == TSDFSystem::Query, voxel_tsdf.cu:532-559,
 * modules/tsdf_module.cc:39-43.  Output: every voxel of every allocated block that lies wholly
 * inside the bounds, ordered by ascending hash-entry index then x + 8y + 64z.  *out is owned by
 * the library until ratsdf_free_buffer. */
int ratsdf_query(ratsdf_engine* e, const ratsdf_bounds* bounds, ratsdf_voxel_tsdf** out, size_t* n);
/* TSDFGrid::GatherValid, voxel_tsdf.cu:476-502 */
int ratsdf_gather_valid(ratsdf_engine* e, ratsdf_voxel_tsdf** out, size_t* n);
/* TSDFGrid::GatherValidSemantic, voxel_tsdf.cu:504-530 (the DownloadAll payload) */
int ratsdf_gather_valid_semantic(ratsdf_engine* e, ratsdf_voxel_segm** out, size_t* n);
/* TSDFSystem::DownloadAll(file_path): raw 20-byte records, modules/tsdf_module.cc:57-64 */
int ratsdf_download_all(ratsdf_engine* e, const char* file_path);
int ratsdf_free_buffer(void* p);

/* ---- rendering ------------------------------------------------------------------ */
/* TSDFGrid::RayCast(max_depth, virtual_cam, cam_T_world, tsdf_rgba, tsdf_normal),
 * utils/tsdf/voxel_tsdf.cu:278-374,885-902 == TSDFSystem::Render (modules/tsdf_module.cc:45-55, which
 * passes 2 * max_depth).  The reference writes two uchar4 images into OpenGL textures through
 * CUDA-GL interop (utils/gl/image.cc:108-119); here they are written to host buffers of
 * height*width*4 bytes (either may be NULL).  Step size is truncation / 2 (voxel_tsdf.cu:892). */
int ratsdf_raycast(ratsdf_engine* e, const ratsdf_intrinsics* virtual_cam, int height, int width,
                   const ratsdf_pose* cam_T_world, float max_depth, uint8_t* rgba, uint8_t* normal);
/* Rows [row0, row1) of the same rendering (buffers of (row1 - row0) * width * 4 bytes): what one rank of a map
 * that is spread over GPUs renders when the IMAGE is partitioned (ratsdf.multi.raycast_across_shards) -- a strip
 * cannot be had by shifting cy, that changes the float arithmetic of the rays.  No reference counterpart. */
int ratsdf_raycast_rows(ratsdf_engine* e, const ratsdf_intrinsics* virtual_cam, int height, int width,
                        const ratsdf_pose* cam_T_world, float max_depth, int row0, int row1, uint8_t* rgba,
                        uint8_t* normal);
/* Same, output to device buffers, asynchronous on the engine's stream.  HIP engine only. */
int ratsdf_raycast_device(ratsdf_engine* e, const ratsdf_intrinsics* virtual_cam, int height,
                          int width, const ratsdf_pose* cam_T_world, float max_depth, void* d_rgba,
                          void* d_normal);

/* ---- mesh export ---------------------------------------------------------------- */
/* TSDFGrid::GatherValidMesh(vertex_buffer, index_buffer, vertex_prob_buffer),
 * utils/tsdf/voxel_tsdf.cu:736-845 with marching_cube_kernel :561-715: marching cubes over every
 * allocated block (voxels observed with weight > 10 only), vertices in metres (3 floats each),
 * triangles as 3 vertex indices, one probability per vertex.  Buffers are owned by the library until
 * ratsdf_free_buffer. */
int ratsdf_gather_valid_mesh(ratsdf_engine* e, float** vertices, size_t* n_vertices,
                             int32_t** indices, size_t* n_triangles, float** vertex_prob);
/* TSDFSystem::DownloadAllMesh(vertices_path, indices_path, prob_path), modules/tsdf_module.cc:66-86:
 * three raw files (float3 per vertex, int3 per triangle, float per vertex), the format read by
 * python_utils/mesh_processor.py:10-15. */
int ratsdf_download_all_mesh(ratsdf_engine* e, const char* vertices_path, const char* indices_path,
                             const char* prob_path);

/* ---- several streams on one GPU --------------------------------------------------------------- */
/* Frame-batched integration of concurrent streams (BASELINE configs[4] on one device; no reference
 * counterpart: the reference runs one TSDFGrid per process, modules/tsdf_module.h:152-164).  A group
 * steps its member engines together: frame f of every member goes through ONE launch triple whose
 * grids have a slice per member, which fills the chip that a single 640x480 frame leaves mostly
 * waiting on memory round trips.  Every member's map is exactly what ratsdf_integrate_device_batch on
 * that member alone would have produced.  Members must live on one device and share voxel size,
 * truncation and table sizes; they stay usable on their own between group calls (the group orders
 * its work after / before theirs with events).  HIP engine only. */
typedef struct ratsdf_group ratsdf_group;
int ratsdf_group_create(ratsdf_engine* const* engines, int n_engines, ratsdf_group** out);
int ratsdf_group_destroy(ratsdf_group* g);   /* the member engines are not destroyed */
int ratsdf_group_size(ratsdf_group* g, int32_t* out);
/* n_frames consecutive frames of every member stream, inputs resident in HBM.  Element
 * [f * n_engines + s] of each array belongs to frame f of member s; otherwise as
 * ratsdf_integrate_device_batch (d_ht / d_lt may be NULL).  Asynchronous. */
int ratsdf_group_integrate_device_batch(ratsdf_group* g, int n_frames, const void* const* d_rgb,
                                        const void* const* d_depth, const void* const* d_ht,
                                        const void* const* d_lt, int height, int width,
                                        float max_depth, const ratsdf_intrinsics* intrinsics,
                                        const ratsdf_pose* cam_T_world);
/* waits for the group's work and runs ratsdf_synchronize on every member (first error wins) */
int ratsdf_group_synchronize(ratsdf_group* g);
/* as ratsdf_profile_enable / _read, for the group's k_integrate launches (one launch = all members) */
int ratsdf_group_profile_enable(ratsdf_group* g, int enable);
int ratsdf_group_profile_read(ratsdf_group* g, double* integrate_ms, int64_t* launches);

/* ---- multi-GPU support --------------------------------------------------------------------- */
/* Writes the compact block directory (allocated entries in ascending entry order, 12 B each) into a
 * caller-provided DEVICE buffer so it can be all-gathered with RCCL without touching the host.
 * d_count (device int32) receives the number of ALLOCATED entries, which may exceed `capacity`: only
 * the first `capacity` of them are written, and the engine's sticky status becomes
 * RATSDF_ERR_CAPACITY (reported by the next ratsdf_synchronize), so a truncated directory is never
 * mistaken for a complete one.  Enqueued on the engine's stream.  HIP engine only. */
int ratsdf_export_directory_device(ratsdf_engine* e, void* d_blocks, int32_t capacity,
                                   void* d_count);

/* The directory DELTA since the previous call ("RCCL ncclAllGather ... of the block directory delta"):
 * entries added or changed first, then {position, offset 0, idx -1} per deleted position, into d_payload
 * (`capacity` 12-byte entries); d_counts = int32[2] {added, deleted}, the TRUE numbers -- their sum exceeding
 * `capacity`, or deleted == 0x7FFFFFFF (the engine's log of deleted positions overflowed), means the delta is
 * unusable and the caller exports the whole directory instead.  A position deleted and inserted again since the
 * previous call is in both lists: apply "drop, then add".  d_payload == NULL forgets the changes so far (call it
 * after a whole-directory export).  The engine starts keeping this record with the FIRST call (an engine nobody
 * asks for deltas pays nothing for them): that call delivers no delta -- with a payload it reports deleted ==
 * 0x7FFFFFFF.  Asynchronous on the engine's stream, touches no sticky error.  No reference
 * counterpart; the oracle reports RATSDF_ERR_NOT_IMPLEMENTED. */
int ratsdf_export_directory_delta_device(ratsdf_engine* e, void* d_payload, int32_t capacity, void* d_counts);

/* Blocks copied in from ANOTHER map (a neighbour rank's subvolume, SURVEY 8e: "so any rank can answer
 * Query / ray-cast / mesh across subvolume seams"): inserts the n blocks (8^3 voxels each, x + 8y + 64z order)
 * whatever the engine's shard filter says -- an existing block is overwritten -- so that consumers which read
 * NEIGHBOUR blocks (marching cubes: voxel_tsdf.cu:582-620 reads the 2x2x2 block neighbourhood) find them.
 * ratsdf_gather_valid_mesh of a sharded engine emits the cells of the blocks the engine OWNS only, so imported
 * blocks are read, never meshed.  Meant for a scratch engine built for one export (ratsdf.multi.mesh_across_shards);
 * frames integrated afterwards would update imported blocks like any other.  No reference counterpart. */
int ratsdf_import_blocks(ratsdf_engine* e, int32_t n, const int16_t* block_pos, const float* tsdf,
                         const ratsdf_rgbw* rgbw, const float* prob);
/* The same exchange with the voxel data staying in device memory (what the across-shard exports use under RCCL: the
 * all-gather's device buffer is filled by the owner's engine and read by the receiver's scratch engine, no host copy
 * of voxel data).  d_block_pos = n x 3 int16, d_voxels = n records of 1536 32-bit words {tsdf[512] | rgbw[512] |
 * prob[512]}, voxel order x + 8y + 64z -- both DEVICE pointers of the engine's device.
 *   export: record i = the voxels of the block at position i; a block the map does not hold leaves a record of zeros
 *           and is counted in *d_missing (int32, device).  Asynchronous on the engine's stream.
 *   import: as ratsdf_import_blocks.  Reads the buffers on the engine's stream (order them before it, e.g. by
 *           synchronising the producer) and returns when the blocks are in.
 * HIP engine only; the oracle reports RATSDF_ERR_NOT_IMPLEMENTED.  No reference counterpart. */
int ratsdf_export_blocks_device(ratsdf_engine* e, int32_t n, const void* d_block_pos, void* d_voxels,
                                void* d_missing);
int ratsdf_import_blocks_device(ratsdf_engine* e, int32_t n, const void* d_block_pos, const void* d_voxels);

/* ---- test / inspection hooks (mirror the reference's gtest kernels) ------------------------- */
/* One allocation pass over an explicit list of block positions (3 x int16 each), request i having
 * raster rank i, followed by ResetLocks: the Allocate<<<>>> kernel + ResetLocks of
 * utils/tests/voxel_hash_test.cu:36-39,98-99,140-141. */
int ratsdf_test_allocate(ratsdf_engine* e, const int16_t* block_pos, int32_t n);
/* One carve pass that requests deletion of the listed blocks: like space carving, the blocks are
 * looked up first and then deleted in ascending hash-entry order (the order of the reference's
 * visible list, voxel_tsdf.cu:847-867), followed by ResetLocks: VoxelHashTable::Delete,
 * voxel_hash.cu:110-159. */
int ratsdf_test_delete(ratsdf_engine* e, const int16_t* block_pos, int32_t n);
/* VoxelHashTable::Retrieve<Voxel>(point, cache) with a fresh cache per point,
 * voxel_hash.cuh:104-143; voxel_hash_test.cu:41-45.  points = 3 x int16 voxel coordinates.
 * Any output array may be NULL.  Misses return the default voxel (weight 0 / tsdf -10 / prob 0,
 * voxel_types.cu:3,8,11) and block {pos, offset -1, idx -1} (voxel_hash.cu:214-217). */
int ratsdf_test_retrieve(ratsdf_engine* e, const int16_t* points, int32_t n, ratsdf_rgbw* rgbw,
                         float* tsdf, float* prob, ratsdf_block* blocks);
/* *RetrieveMutable<VoxelRGBW>(point) = value, voxel_hash_test.cu:47-54.  Points whose block is
 * not allocated are skipped (the reference asserts). */
int ratsdf_test_assign_rgbw(ratsdf_engine* e, const int16_t* points, const ratsdf_rgbw* values,
                            int32_t n);
/* Compact dump of the hash directory: allocated entries in ascending entry order. */
int ratsdf_dump_directory(ratsdf_engine* e, int32_t** entry_index, ratsdf_block** blocks,
                          size_t* n);
/* Raw voxel storage of the given pool blocks: 512 voxels each, index x + 8y + 64z. */
int ratsdf_dump_voxels(ratsdf_engine* e, const int32_t* pool_idx, int32_t n, float* tsdf,
                       ratsdf_rgbw* rgbw, float* prob);
/* Free-list state: num_free and the heap array (2^block_bits ints), voxel_mem.cu:15-24. */
int ratsdf_dump_heap(ratsdf_engine* e, int32_t* num_free, int32_t* heap);

const char* ratsdf_status_string(int status);
/* "hip-gfx950" for the engine, "cpu-oracle" for the oracle. */
const char* ratsdf_backend(void);

#ifdef __cplusplus
}
#endif
#endif /* RATSDF_H_ */
