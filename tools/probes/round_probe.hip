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
  return 0;
}
