// scan_i8_probe.hip -- timing / per-wave cycle accounting of scan_i8.hip's tile-best modes (diagnostic)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DCRS_STAMPS] -o scan_i8_probe scan_i8_probe.hip
//   ./scan_i8_probe <rows> <dim> <nq> <k>
#include "../compressed-rag-suite_amd/csrc/scan_i8.hip"

#include <algorithm>
#include <stdio.h>
#include <vector>

int main(int argc, char** argv) {
  const int rows = argc > 1 ? atoi(argv[1]) : 1250000;
  const int dim = argc > 2 ? atoi(argv[2]) : 768;
  const int nq = argc > 3 ? atoi(argv[3]) : 64;
  const int k = argc > 4 ? atoi(argv[4]) : 10;
  const int wgpc = argc > 5 ? atoi(argv[5]) : 2;
  const int tr = crs::scan_i8_tile_rows();
  const int n_tiles = (rows + tr - 1) / tr;
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int nwg = std::min(n_tiles, prop.multiProcessorCount * wgpc);
  char* slab; _Float16* q; float *ps, *sc; int* pr;
  hipMalloc(&slab, (size_t)rows * dim); hipMalloc(&q, (size_t)nq * dim * 2); hipMalloc(&sc, (size_t)rows * 4);
  hipMemset(slab, 0x11, (size_t)rows * dim); hipMemset(q, 0x11, (size_t)nq * dim * 2); hipMemset(sc, 0x11, (size_t)rows * 4);
  hipMalloc(&ps, (size_t)nwg * nq * 64 * 4); hipMalloc(&pr, (size_t)nwg * nq * 64 * 4);
  crs::ScanArgs a{};
  a.q = q; a.slab = slab; a.scales = sc; a.part_scores = ps; a.part_rows = pr;
  a.n_rows = rows; a.n_tiles = n_tiles; a.nq = nq; a.k = k; a.nwg = nwg; a.nqb = 1;
  const int tps = (n_tiles + nwg - 1) / nwg;
  const int slots = tps <= 20 ? 0 : (k <= 4 ? 4 : k <= 10 ? 10 : 16);
  a.kp = slots ? slots : tps;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) crs::scan_launch_i8(a, dim, slots, 0);
  hipEventRecord(e0, 0);
  const int it = 50;
  for (int i = 0; i < it; ++i) crs::scan_launch_i8(a, dim, slots, 0);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / it;
#ifdef CRS_STAMPS
  {
    const size_t nst = (size_t)nwg * 4 * 12;
    unsigned long long* st; hipMalloc(&st, nst * 8); hipMemset(st, 0, nst * 8);
    a.stamps = st;
    crs::scan_launch_i8(a, dim, slots, 0);
    hipDeviceSynchronize();
    std::vector<unsigned long long> hs(nst);
    hipMemcpy(hs.data(), st, nst * 8, hipMemcpyDeviceToHost);
    const char* names[12] = {"prologue", "look-ahead issue", "", "fragment reads + MFMA", "scores + selection", "wait for loads",
                             "LDS store", "barrier", "", "", "exit", "TOTAL"};
    for (int i = 0; i < 12; ++i) {
      if (!names[i][0]) continue;
      std::vector<double> v;
      for (size_t w = 0; w < nst / 12; ++w) v.push_back((double)hs[w * 12 + i]);
      std::sort(v.begin(), v.end());
      double sum = 0; for (double x : v) sum += x;
      printf("  %-24s mean %9.0f  median %9.0f  max %9.0f\n", names[i], sum / v.size(), v[v.size() / 2], v.back());
    }
  }
#endif
  printf("int8 rows %d dim %d nq %d  streams %d (%d/CU) tiles/stream %.1f slots %d : %.1f us  %.0f GB/s\n", rows, dim, nq, nwg, wgpc,
         (double)n_tiles / nwg, slots, us, ((double)rows * dim + rows * 4.0) / us / 1e3);
  return 0;
}
