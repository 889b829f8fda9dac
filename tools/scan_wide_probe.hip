// scan_wide_probe.hip -- diagnostic build of scan_wide.hip with per-wave s_memtime accumulators.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scan_wide_probe scan_wide_probe.hip
//   ./scan_wide_probe <rows> <dim> <nq> <k>
#define CRS_STAMPS 1
#include "../compressed-rag-suite_amd/csrc/scan_wide.hip"

#include <algorithm>
#include <math.h>
#include <stdio.h>
#include <vector>

int main(int argc, char** argv) {
  const int rows = argc > 1 ? atoi(argv[1]) : 1250000;
  const int dim = argc > 2 ? atoi(argv[2]) : 384;
  const int nq = argc > 3 ? atoi(argv[3]) : 256;
  const int k = argc > 4 ? atoi(argv[4]) : 10;
  const int nw = crs::scan_wide_waves(nq, k, dim);
  if (!nw) { printf("wide kernel not applicable\n"); return 1; }
  const int tile_rows = crs::scan_wide_tile_rows(nw, dim);
  const int n_tiles = (rows + tile_rows - 1) / tile_rows;
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int nqb = (nq + 32 * nw - 1) / (32 * nw);
  int nwg = prop.multiProcessorCount * crs::scan_wide_wg_per_cu(nw, dim) / nqb;
  if (nqb > 1) nwg &= ~7;
  nwg = std::min(nwg, n_tiles);
  std::vector<_Float16> h((size_t)rows * dim), hq((size_t)nq * dim);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; float u = 0; for (int i = 0; i < 4; ++i) { s = s * 1664525u + 1013904223u; u += ((s >> 8) & 0xffff) / 65536.0f - 0.5f; } return u; };
  const float sc = 1.0f / sqrtf((float)dim / 3.0f);
  for (auto& x : h) x = (_Float16)(rnd() * sc);
  for (auto& x : hq) x = (_Float16)(rnd() * sc);
  _Float16 *slab, *q; float* ps; int* pr; unsigned long long* st;
  hipMalloc(&slab, h.size() * 2); hipMalloc(&q, hq.size() * 2);
  hipMalloc(&ps, (size_t)nwg * nq * 32 * 4); hipMalloc(&pr, (size_t)nwg * nq * 32 * 4);
  const size_t nst = (size_t)nwg * nqb * nw * 12;
  hipMalloc(&st, nst * 8);
  hipMemcpy(slab, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(q, hq.data(), hq.size() * 2, hipMemcpyHostToDevice);
  crs::ScanArgs a{};
  a.q = q; a.slab = slab; a.part_scores = ps; a.part_rows = pr; a.stamps = st;
  a.n_rows = rows; a.n_tiles = n_tiles; a.nq = nq; a.k = k; a.kp = 2 * crs::scan_wide_slots(k); a.nwg = nwg; a.nqb = nqb; a.sched = 2;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 4; ++rep) {
    hipMemset(st, 0, nst * 8);
    hipEventRecord(e0, 0);
    int e = crs::scan_launch_wide(a, dim, nw, 0);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    if (e) { printf("launch error %d\n", e); return 1; }
    hipEventElapsedTime(&ms, e0, e1);
  }
  std::vector<unsigned long long> hs(nst);
  hipMemcpy(hs.data(), st, nst * 8, hipMemcpyDeviceToHost);
  printf("rows %d dim %d nq %d k %d | waves/wg %d nqb %d streams %d tiles/stream %.1f | kernel %.1f us (with stamps)\n", rows, dim, nq, k, nw, nqb, nwg,
         (double)n_tiles / nwg, ms * 1e3);
  // slots 2 / 4: the selection of waves 4..7 (deferred one tile) / of waves 0..3; 5, 8, 9: unused since the tile-best rewrite
  const char* names[12] = {"prologue", "tile-load issue", "selection (waves 4-7, deferred)", "MFMA sweep", "selection (waves 0-3)", "(unused)",
                           "wait next tile + LDS store", "barrier", "(unused)", "(unused)", "final flush", "TOTAL"};
  const size_t nwaves = nst / 12;
  for (int i = 0; i < 12; ++i) {
    std::vector<double> v; for (size_t w = 0; w < nwaves; ++w) v.push_back((double)hs[w * 12 + i]);
    std::sort(v.begin(), v.end());
    double sum = 0; for (double x : v) sum += x;
    printf("  %-28s mean %9.0f  median %9.0f  max %9.0f\n", names[i], sum / nwaves, v[nwaves / 2], v.back());
  }
  return 0;
}
