// round_probe.hip -- is v_cvt_rpi_i32_f32(|x|) == (int)roundf(|x|) for every float?  (GPU box)
// roundf = round half away from zero; the ISA describes v_cvt_rpi_i32_f32 as floor(x + 0.5).  If the
// hardware evaluates that exactly (not as an fp32 addition), the two agree for every x >= 0.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>

__device__ inline int f2i(float f) { int r; asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(f)); return r; }
__device__ inline int rpi(float f) { int r; asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(f)); return r; }

__global__ void k_check(unsigned long long* bad, unsigned* first_bad) {
  // all non-negative float bit patterns (0 .. 0x7FFFFFFF), grid-strided
  const unsigned long long n = 0x80000000ull;
  unsigned long long local = 0;
  for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n;
       i += (unsigned long long)gridDim.x * blockDim.x) {
    const float x = __uint_as_float((unsigned)i);
    const int a = f2i(roundf(x));
    const int b = rpi(x);
    if (a != b) {
      ++local;
      atomicMin(first_bad, (unsigned)i);
    }
  }
  if (local) atomicAdd(bad, local);
}

// Pixel pick of the voxel update (kernels_integrate.h): u = (int)roundf(x), "0 <= u < W" as one unsigned
// compare -- against the short form  ua = v_cvt_rpi_i32_f32(|x|),  in = ua < W && !(x <= -0.5)
// (a negative x lands in the image only if it rounds to 0).  Every one of the 2^32 bit patterns, NaNs,
// infinities and -0.5 itself included, for several widths.
__device__ inline int rpi_abs(float f) { int r; asm("v_cvt_rpi_i32_f32_e64 %0, |%1|" : "=v"(r) : "v"(f)); return r; }
__global__ void k_pick(unsigned long long* bad, unsigned* first_bad) {
  const unsigned widths[7] = {1u, 2u, 3u, 640u, 1280u, 1920u, 65535u};
  unsigned long long local = 0;
  for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < 0x100000000ull;
       i += (unsigned long long)gridDim.x * blockDim.x) {
    const float x = __uint_as_float((unsigned)i);
    const int u = f2i(roundf(x));
    const int ua = rpi_abs(x);
    const bool neg_out = !(x <= -0.5f);
    bool ok = true;
    for (int k = 0; k < 7; ++k) {
      const bool in_old = (unsigned)u < widths[k];
      const bool in_new = (unsigned)ua < widths[k] && neg_out;
      ok = ok && in_old == in_new && (!in_old || u == ua);
    }
    if (!ok) {
      ++local;
      atomicMin(first_bad, (unsigned)i);
    }
  }
  if (local) atomicAdd(bad, local);
}

int main() {
  unsigned long long* d_bad;
  unsigned* d_first;
  hipMalloc(&d_bad, 8);
  hipMalloc(&d_first, 4);
  hipMemset(d_bad, 0, 8);
  hipMemset(d_first, 0xFF, 4);
  hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, 0, d_bad, d_first);
  hipDeviceSynchronize();
  unsigned long long bad = 0;
  unsigned first = 0;
  hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost);
  hipMemcpy(&first, d_first, 4, hipMemcpyDeviceToHost);
  float f;
  memcpy(&f, &first, 4);
  printf("mismatches over all non-negative floats: %llu (first at bits 0x%08x = %g)\n", bad, first, f);
  hipMemset(d_bad, 0, 8);
  hipMemset(d_first, 0xFF, 4);
  hipLaunchKernelGGL(k_pick, dim3(4096), dim3(256), 0, 0, d_bad, d_first);
  hipDeviceSynchronize();
  hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost);
  hipMemcpy(&first, d_first, 4, hipMemcpyDeviceToHost);
  memcpy(&f, &first, 4);
  printf("pixel pick: mismatches over all 2^32 floats x 7 widths: %llu (first at bits 0x%08x = %g)\n", bad, first, f);
  return bad != 0;
}
