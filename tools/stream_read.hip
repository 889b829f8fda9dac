// stream_read.hip -- calibration: what read bandwidth does THIS box sustain for a linear sweep
// with 16-byte-per-lane loads?  (The scan kernel's roofline is quoted against 8.0 TB/s spec; this
// gives the practical ceiling beside it.)   hipcc --offload-arch=gfx950 -O3 -o stream_read stream_read.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int UNROLL>
__global__ __launch_bounds__(256) void sweep(const uint4* __restrict__ p, size_t n16, unsigned* out) {
  size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
  unsigned acc = 0;
  for (; i + (UNROLL - 1) * 256 < n16; i += stride) {
    uint4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = p[i + u * 256];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <int UNROLL>
float run(const uint4* p, size_t n16, unsigned* out, int grid, int iters) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(sweep<UNROLL>, dim3(grid), dim3(256), 0, 0, p, n16, out);
  hipEventRecord(a, 0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(sweep<UNROLL>, dim3(grid), dim3(256), 0, 0, p, n16, out);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / iters;
}

int main(int argc, char** argv) {
  const size_t bytes = argc > 1 ? strtoull(argv[1], 0, 10) : (size_t)960000000;
  const size_t n16 = bytes / 16;
  uint4* p; unsigned* out;
  hipMalloc(&p, n16 * 16); hipMalloc(&out, 4);
  hipMemset(p, 1, n16 * 16);
  const int grids[] = {256, 512, 1024, 2048, 4096};
  for (int g : grids) {
    float m4 = run<4>(p, n16, out, g, 20), m8 = run<8>(p, n16, out, g, 20), m16 = run<16>(p, n16, out, g, 20);
    printf("bytes %zu grid %5d  unroll4 %.1f GB/s  unroll8 %.1f GB/s  unroll16 %.1f GB/s\n", bytes, g,
           bytes / m4 / 1e6, bytes / m8 / 1e6, bytes / m16 / 1e6);
  }
  return 0;
}
