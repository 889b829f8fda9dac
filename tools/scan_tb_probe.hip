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
  printf("rows %d dim %d nq %d  streams %d (%d/CU) tiles/stream %.1f slots %d : %.1f us  %.0f GB/s\n", rows, dim, nq, nwg, wgpc,
         (double)n_tiles / nwg, slots, us, (double)rows * dim * 2 / us / 1e3);
  return 0;
}
