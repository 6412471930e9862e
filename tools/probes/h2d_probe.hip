// h2d_probe.hip -- what the host link gives page-locked -> device copies of frame size (GPU box).
// One 640x480 frame is 4.6 MB (15 B/pixel); ratsdf_integrate_batch(pinned) sends one copy per frame on two
// alternating streams and reaches ~41 GB/s.  This prints GB/s for copies of 1, 2, 4, 8 frames on 1 and 2 streams,
// so that "what is left to gain from larger copies" is a measured number.
//   hipcc --offload-arch=gfx950 -O2 -o h2d_probe h2d_probe.hip && ./h2d_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>

int main() {
  const size_t frame = (size_t)640 * 480 * 16;  // the staging slot's stride
  const int frames = 64;
  void *h = nullptr, *d = nullptr;
  if (hipHostMalloc(&h, frame * frames, hipHostMallocDefault) != hipSuccess || hipMalloc(&d, frame * frames) != hipSuccess) {
    fprintf(stderr, "allocation failed\n");
    return 1;
  }
  memset(h, 1, frame * frames);
  hipStream_t s[4];
  for (auto& x : s) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
  for (int ns : {1, 2, 4})
    for (int run : {1, 2, 4, 8, 16}) {
      double best = 0;
      for (int rep = 0; rep < 4; ++rep) {
        hipDeviceSynchronize();
        const auto t0 = std::chrono::steady_clock::now();
        const int rounds = 8;
        for (int r = 0; r < rounds; ++r)
          for (int i = 0, k = 0; i + run <= frames; i += run, ++k)
            hipMemcpyAsync((char*)d + (size_t)i * frame, (char*)h + (size_t)i * frame, frame * run,
                           hipMemcpyHostToDevice, s[k % ns]);
        hipDeviceSynchronize();
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const double gbps = (double)rounds * (frames / run) * run * frame / dt / 1e9;
        if (gbps > best) best = gbps;
      }
      printf("streams %d  frames/copy %2d  (%.1f MB)  %.1f GB/s\n", ns, run, run * frame / 1e6, best);
    }
  return 0;
}
