// device_math.h -- fp32 camera / pose arithmetic of the TSDF path, in the reference's evaluation
// order.  Compiled with -ffp-contract=off: every multiply and add below is a separate rounding, as
// in the reference's nvcc device code for these expressions is NOT guaranteed (nvcc contracts), so
// the canonical order is the one fixed by the oracle and restated here operation by operation.
//
// Follows utils/cuda/camera.cuh:35-51 (K, K^-1), utils/cuda/lie_group.cuh:25-40 (SE3) and Eigen's
// quaternion * vector (uv = 2 q.vec x v; v + w uv + q.vec x uv), hnormalized (per-component
// division) and norm().
#pragma once
#include "device_types.h"

namespace ratsdf {

__host__ __device__ inline V3 cross3(const V3& a, const V3& b) {
  return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

__host__ __device__ inline V3 quat_rotate(const Quat& q, const V3& v) {
  const V3 qv{q.x, q.y, q.z};
  V3 uv = cross3(qv, v);
  uv.x += uv.x;
  uv.y += uv.y;
  uv.z += uv.z;
  const V3 c = cross3(qv, uv);
  return V3{(v.x + q.w * uv.x) + c.x, (v.y + q.w * uv.y) + c.y, (v.z + q.w * uv.z) + c.z};
}

__host__ __device__ inline V3 se3_apply(const Se3& T, const V3& v) {
  const V3 r = quat_rotate(T.q, v);
  return V3{r.x + T.t.x, r.y + T.t.y, r.z + T.t.z};
}

// SE3::Inverse (lie_group.cuh:25-27), host side like the reference (voxel_tsdf.cu:459)
inline Se3 se3_inverse(const Se3& T) {
  const Quat& q = T.q;
  const float n2 = (q.x * q.x + q.y * q.y) + (q.z * q.z + q.w * q.w);
  Quat qi;
  if (n2 > 0.f) {
    qi = Quat{(-q.x) / n2, (-q.y) / n2, (-q.z) / n2, q.w / n2};
  } else {
    qi = Quat{0.f, 0.f, 0.f, 0.f};
  }
  const V3 nt{-T.t.x, -T.t.y, -T.t.z};
  return Se3{qi, quat_rotate(qi, nt)};
}

__host__ __device__ inline V3 intr_mul(const Intr& K, const V3& v) {
  return V3{K.fx * v.x + K.cx * v.z, K.fy * v.y + K.cy * v.z, v.z};
}

inline Intr intr_inverse(const Intr& K) {
  const float fxi = 1 / K.fx;
  const float fyi = 1 / K.fy;
  return Intr{fxi, fyi, -K.cx * fxi, -K.cy * fyi};
}

// float -> int with CUDA cvt.rzi.s32.f32 semantics (round toward zero, saturating, NaN -> 0).
// v_cvt_i32_f32 has exactly these semantics in hardware; the asm keeps the compiler from treating
// out-of-range inputs as undefined behaviour.
__device__ inline int f2i(float f) {
  int r;
  asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(f));
  return r;
}

// |(int)roundf(x)| in one instruction: v_cvt_rpi_i32_f32 of |x| = floor(|x| + 0.5) evaluated exactly = roundf(|x|)
// for every float (tools/probes/round_probe.hip), saturating at INT_MAX, NaN -> 0.
__device__ inline uint32_t rpi_abs(float x) {
  int r;
  asm("v_cvt_rpi_i32_f32_e64 %0, |%1|" : "=v"(r) : "v"(x));
  return (uint32_t)r;
}
// (int)roundf(x) for |x| < 2^31 (roundf is symmetric about zero: the magnitude above, then the sign): four
// instructions instead of the eight of roundf + conversion.  NaN -> 0 like the conversion of a NaN.
__device__ inline int round_to_int(float x) {
  const int a = (int)rpi_abs(x);
  return x < 0.f ? -a : a;
}

// IEEE-exact division with a shared divisor.  hipcc expands `a / b` (correctly rounded, no fast-math)
// into: r0 = rcp(b); e = fma(-b, r0, 1); r1 = fma(e, r0, r0); q0 = a * r1; e2 = fma(-b, q0, a);
// q1 = fma(e2, r1, q0); e3 = fma(-b, q1, a); q = fma(e3, r1, q1), wrapped in v_div_scale /
// v_div_fixup for operands near the ends of the exponent range and for inf / NaN / 0.  When several
// numerators share one divisor the refined reciprocal r1 is computed once and every quotient costs
// five instructions instead of ten -- the same FMA sequence, hence the same correctly rounded result,
// for finite operands away from the exponent limits (|b| in [2^-60, 2^60], quotient not subnormal),
// which is what the callers guarantee or guard.
struct Recip {
  float d, r1;
};
__device__ inline Recip make_recip(float d) {
  const float r0 = __builtin_amdgcn_rcpf(d);
  const float e = fmaf(-d, r0, 1.f);
  return Recip{d, fmaf(e, r0, r0)};
}
__device__ inline float div_shared(float a, const Recip& b) {
  const float q0 = a * b.r1;
  const float e2 = fmaf(-b.d, q0, a);
  const float q1 = fmaf(e2, b.r1, q0);
  const float e3 = fmaf(-b.d, q1, a);
  return fmaf(e3, b.r1, q1);
}
__device__ inline bool recip_safe(float d) {  // divisor range in which div_shared equals IEEE `/`
  const float ad = fabsf(d);
  return ad > 1e-18f && ad < 1e18f;
}

// Hash(), utils/tsdf/voxel_hash.cu:19-23
__host__ __device__ inline uint32_t block_hash(int x, int y, int z, uint32_t mask) {
  return (((uint32_t)x * 73856093u) ^ ((uint32_t)y * 19349669u) ^ ((uint32_t)z * 83492791u)) & mask;
}

// is_voxel_visible, utils/tsdf/voxel_tsdf.cu:64-73
__device__ inline bool voxel_visible(int gx, int gy, int gz, const FrameParams& P) {
  const V3 pw{(float)gx * P.vs, (float)gy * P.vs, (float)gz * P.vs};
  const V3 pc = se3_apply(P.T, pw);
  const V3 ph = intr_mul(P.K, pc);
  const float u = ph.x / ph.z;
  const float v = ph.y / ph.z;
  return (u >= 0 && u <= (float)(P.W - 1) && v >= 0 && v <= (float)(P.H - 1) && ph.z >= 0);
}

// is_block_visible<Full>, utils/tsdf/voxel_tsdf.cu:75-96 (corner coordinates in short arithmetic).
// The reference ANDs / ORs all 8 corner tests; the result is the same when the loop stops at the
// first corner that decides it, which matters here: blocks straddling the image border are requested
// again every frame and fail on an early corner, blocks in view pass the "any corner" test at once.
template <bool Full>
__device__ inline bool block_visible(int bx, int by, int bz, const FrameParams& P) {
  const int x = (int16_t)(bx << 3), y = (int16_t)(by << 3), z = (int16_t)(bz << 3);
#pragma unroll 1
  for (int i = 0; i < 8; ++i) {
    const int cx = (int16_t)(x + ((i >> 0) & 1) * 7);
    const int cy = (int16_t)(y + ((i >> 1) & 1) * 7);
    const int cz = (int16_t)(z + ((i >> 2) & 1) * 7);
    const bool v = voxel_visible(cx, cy, cz, P);
    if (Full && !v) return false;
    if (!Full && v) return true;
  }
  return Full;
}

// Which of the 8 per-XCD work lists a block belongs to: the image is cut into 8x8 tiles and tiles
// are dealt to the lists so that every list covers 8 scattered tiles (load balance).  Workgroup b of
// k_integrate serves list b & 7 and workgroups b, b+8, ... share an XCD (observed round-robin
// placement), so the per-pixel texels of a tile are pulled into ONE XCD's L2 instead of all eight.
// Placement only affects speed: any list is a correct home for any block.
__device__ inline int block_list_of(int bx, int by, int bz, const FrameParams& P) {
  const V3 pw{(float)(bx * 8 + 4) * P.vs, (float)(by * 8 + 4) * P.vs, (float)(bz * 8 + 4) * P.vs};
  const V3 pc = se3_apply(P.T, pw);
  const V3 ph = intr_mul(P.K, pc);
  float u = ph.x / ph.z, v = ph.y / ph.z;
  u = fminf(fmaxf(u, 0.f), (float)(P.W - 1));  // NaN -> 0
  v = fminf(fmaxf(v, 0.f), (float)(P.H - 1));
  const int tx = (int)(u * 8.f / (float)P.W), ty = (int)(v * 8.f / (float)P.H);
  return (tx + 3 * ty) & 7;
}

// owner of a block = floormod(bx >> slab_bits, shard_count).  bx is a short: shifted by shard_bias (a
// multiple of shard_count >= 32768) the slab number is non-negative, and its remainder comes from one multiply-high
// with shard_magic = floor(2^32 / shard_count) + 1 (exact while (slab + bias) * shard_count < 2^32) -- the `%` of a
// run-time divisor was 17 vector instructions per ray sample, evaluated whether the map is sharded or not.
__device__ inline bool shard_owned(int bx, const FrameParams& P) {
  if (P.shard_count <= 1) return true;
  const uint32_t u = (uint32_t)((bx >> P.shard_slab_bits) + P.shard_bias);
  const uint32_t m = u - __umulhi(u, P.shard_magic) * (uint32_t)P.shard_count;
  return m == (uint32_t)P.shard_rank;
}

}  // namespace ratsdf
