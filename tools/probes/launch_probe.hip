// launch_probe.hip -- how fast does the chip START the waves of one launch?  (GPU box)
// Every wave stamps the 100 MHz wall clock at its first instruction, then idles for `hold_us`
// (so nothing has to wait for a slot unless the grid exceeds the resident capacity), and the host
// prints the spread of the start stamps for several launch shapes.
//   hipcc --offload-arch=gfx950 -O3 -o launch_probe launch_probe.hip && ./launch_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <int LDS_BYTES, int VGPRS>
__global__ __launch_bounds__(256) void k_probe(unsigned long long* stamps, int hold_ticks, float* sink) {
  const unsigned long long t0 = wall_clock64();
  __shared__ char lds[LDS_BYTES > 0 ? LDS_BYTES : 1];
  float acc[VGPRS];
#pragma unroll
  for (int i = 0; i < VGPRS; ++i) acc[i] = (float)(threadIdx.x + i);
  if (LDS_BYTES > 0) lds[threadIdx.x % (LDS_BYTES > 0 ? LDS_BYTES : 1)] = (char)threadIdx.x;
  while ((long long)(wall_clock64() - t0) < hold_ticks) {
#pragma unroll
    for (int i = 0; i < VGPRS; ++i) acc[i] = acc[i] * 1.0001f + 0.5f;
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < VGPRS; ++i) s += acc[i];
  if (s == 12345.678f) sink[0] = s + (LDS_BYTES > 0 ? lds[0] : 0);
  if ((threadIdx.x & 63) == 0) stamps[(size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t0;
}

template <int LDS_BYTES, int VGPRS>
void run(const char* name, int grid, int threads, int hold_us) {
  const size_t nw = (size_t)grid * (threads / 64);
  unsigned long long* d;
  float* sink;
  hipMalloc(&d, nw * 8);
  hipMalloc(&sink, 4);
  std::vector<unsigned long long> h(nw);
  double best = 1e9, p50 = 0;
  for (int rep = 0; rep < 5; ++rep) {
    hipLaunchKernelGGL((k_probe<LDS_BYTES, VGPRS>), dim3(grid), dim3(threads), 0, 0, d, hold_us * 100, sink);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, nw * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double spread = (h.back() - h.front()) * 0.01;
    if (spread < best) { best = spread; p50 = (h[nw / 2] - h.front()) * 0.01; }
  }
  printf("%-34s grid %5d x %3d thr (%6zu waves), hold %d us: start spread %.2f us (p50 %.2f) -> %.2f ns/wave\n",
         name, grid, threads, nw, hold_us, best, p50, best * 1e3 / nw);
  hipFree(d);
  hipFree(sink);
}

int main() {
  run<0, 4>("tiny (no LDS, few VGPRs)", 1024, 256, 20);
  run<0, 4>("tiny (no LDS, few VGPRs)", 2048, 256, 20);
  run<0, 4>("tiny (no LDS, few VGPRs)", 1920, 256, 20);
  run<6144, 4>("6 KB LDS", 1920, 256, 20);
  run<16384, 4>("16 KB LDS", 1920, 256, 20);
  run<0, 48>("~56 VGPRs", 1920, 256, 20);
  run<6144, 48>("6 KB LDS + ~56 VGPRs", 1920, 256, 20);
  run<6144, 48>("6 KB LDS + ~56 VGPRs, 512 thr", 960, 512, 20);
  run<6144, 48>("6 KB LDS + ~56 VGPRs, 1024 thr", 480, 1024, 20);
  run<0, 4>("tiny, short-lived waves", 4096, 256, 2);
  run<6144, 48>("6 KB LDS + ~56 VGPRs, short-lived", 4096, 256, 2);
  return 0;
}
