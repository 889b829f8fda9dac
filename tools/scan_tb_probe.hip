// scan_tb_probe.hip -- timing-only builds of scan_tb.hip (what does the staging pipeline alone sustain?)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DCRS_TB_EXPERIMENT=1|2] -o scan_tb_probe scan_tb_probe.hip
//   ./scan_tb_probe <rows> <dim> <nq> <k> <wg_per_cu>
#include "../compressed-rag-suite_amd/csrc/scan_tb.hip"

#include <algorithm>
#include <stdio.h>
#include <vector>

int main(int argc, char** argv) {
  const int rows = argc > 1 ? atoi(argv[1]) : 1250000;
  const int dim = argc > 2 ? atoi(argv[2]) : 384;
  const int nq = argc > 3 ? atoi(argv[3]) : 64;
  const int k = argc > 4 ? atoi(argv[4]) : 10;
  const int wgpc = argc > 5 ? atoi(argv[5]) : crs::scan_tb_wg_per_cu(dim, 4);
  const int tr = dim <= 512 ? 32 : 16;
  const int n_tiles = (rows + tr - 1) / tr;
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int nwg = std::min(n_tiles, prop.multiProcessorCount * wgpc);
  _Float16 *slab, *q; float* ps; int* pr;
  hipMalloc(&slab, (size_t)rows * dim * 2); hipMalloc(&q, (size_t)nq * dim * 2);
  hipMemset(slab, 0x11, (size_t)rows * dim * 2); hipMemset(q, 0x11, (size_t)nq * dim * 2);
  hipMalloc(&ps, (size_t)nwg * nq * 64 * 4); hipMalloc(&pr, (size_t)nwg * nq * 64 * 4);
  crs::ScanArgs a{};
  a.q = q; a.slab = slab; a.part_scores = ps; a.part_rows = pr;
  a.n_rows = rows; a.n_tiles = n_tiles; a.nq = nq; a.k = k; a.nwg = nwg; a.nqb = 1;
  const int tps = (n_tiles + nwg - 1) / nwg;
  const int slots = tps <= 20 ? 0 : 10;
  a.kp = slots ? slots : tps;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) crs::scan_launch_tb(a, dim, 4, slots, 0);
  hipEventRecord(e0, 0);
  const int it = 50;
  for (int i = 0; i < it; ++i) crs::scan_launch_tb(a, dim, 4, slots, 0);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / it;
#ifdef CRS_STAMPS
  {   // per-wave cycle accumulators (s_memtime ticks: 100 MHz)
    const size_t nst = (size_t)nwg * 4 * 12;
    unsigned long long* st; hipMalloc(&st, nst * 8); hipMemset(st, 0, nst * 8);
    a.stamps = st;
    crs::scan_launch_tb(a, dim, 4, slots, 0);
    hipDeviceSynchronize();
    std::vector<unsigned long long> hs(nst);
    hipMemcpy(hs.data(), st, nst * 8, hipMemcpyDeviceToHost);
    const char* names[12] = {"prologue", "look-ahead issue", "", "fragment reads + MFMA", "selection", "wait for loads",
                             "LDS store", "barrier", "", "", "exit", "TOTAL"};
    for (int half = 0; half < 2; ++half) {
      printf(" waves %d-%d\n", half * 2, half * 2 + 1);
      for (int i = 0; i < 12; ++i) {
        if (!names[i][0]) continue;
        std::vector<double> v;
        for (size_t w = 0; w < nst / 12; ++w) if ((int)(w & 3) / 2 == half) v.push_back((double)hs[w * 12 + i]);
        std::sort(v.begin(), v.end());
        double sum = 0; for (double x : v) sum += x;
        printf("  %-24s mean %9.0f  median %9.0f  max %9.0f\n", names[i], sum / v.size(), v[v.size() / 2], v.back());
      }
    }
  }
#endif
  printf("rows %d dim %d nq %d  streams %d (%d/CU) tiles/stream %.1f slots %d : %.1f us  %.0f GB/s\n", rows, dim, nq, nwg, wgpc,
         (double)n_tiles / nwg, slots, us, (double)rows * dim * 2 / us / 1e3);
  return 0;
}
