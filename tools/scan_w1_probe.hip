// scan_w1_probe.hip -- diagnostic build of scan_w1.hip with per-wave s_memtime accumulators.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scan_w1_probe scan_w1_probe.hip ; ./scan_w1_probe <rows> <dim> <nq>
#define CRS_STAMPS 1
#include "../compressed-rag-suite_amd/csrc/scan_w1.hip"

#include <algorithm>
#include <math.h>
#include <stdio.h>
#include <vector>

int main(int argc, char** argv) {
  const int rows = argc > 1 ? atoi(argv[1]) : 1000000;
  const int dim = argc > 2 ? atoi(argv[2]) : 768;
  const int nq = argc > 3 ? atoi(argv[3]) : 256;
  const int qpw = crs::scan_w1_queries_per_wg(nq, 16, dim);
  if (!qpw) { printf("w1 kernel not applicable\n"); return 1; }
  const int nwaves_wg = (getenv("CRS_SCAN_W1") && getenv("CRS_SCAN_W1")[0] == '1') ? 4 : 8;
  const int qg = qpw;
  const int n_tiles = (rows + 31) / 32;
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int nqb = (nq + qpw - 1) / qpw;
  int nwg = prop.multiProcessorCount / nqb;
  if (nqb > 1) nwg &= ~7;
  nwg = std::min(nwg, n_tiles);
  const int kp = (n_tiles + nwg - 1) / nwg;
  std::vector<_Float16> h((size_t)rows * dim), hq((size_t)nq * dim);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; float u = 0; for (int i = 0; i < 4; ++i) { s = s * 1664525u + 1013904223u; u += ((s >> 8) & 0xffff) / 65536.0f - 0.5f; } return u; };
  const float sc = 1.0f / sqrtf((float)dim / 3.0f);
  const bool zero = getenv("PROBE_ZERO") != nullptr;   // zero-filled operands: what the same instruction stream does at the un-throttled clock
  for (auto& x : h) x = zero ? (_Float16)0 : (_Float16)(rnd() * sc);
  for (auto& x : hq) x = zero ? (_Float16)0 : (_Float16)(rnd() * sc);
  _Float16 *slab, *q; float* ps; int* pr; unsigned long long* st;
  hipMalloc(&slab, h.size() * 2); hipMalloc(&q, hq.size() * 2);
  hipMalloc(&ps, (size_t)nwg * nq * kp * 4); hipMalloc(&pr, (size_t)nwg * nq * kp * 4);
  const size_t nst = (size_t)nwg * nqb * nwaves_wg * 12;
  hipMalloc(&st, nst * 8);
  hipMemcpy(slab, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(q, hq.data(), hq.size() * 2, hipMemcpyHostToDevice);
  crs::ScanArgs a{};
  a.q = q; a.slab = slab; a.part_scores = ps; a.part_rows = pr; a.stamps = st;
  a.n_rows = rows; a.n_tiles = n_tiles; a.nq = nq; a.k = 16; a.kp = kp; a.nwg = nwg; a.nqb = nqb; a.sched = getenv("PROBE_MODE") ? atoi(getenv("PROBE_MODE")) : 0;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 4; ++rep) {
    hipMemset(st, 0, nst * 8);
    hipEventRecord(e0, 0);
    int e = crs::scan_launch_w1(a, dim, 0);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    if (e) { printf("launch error %d\n", e); return 1; }
    hipEventElapsedTime(&ms, e0, e1);
  }
  std::vector<unsigned long long> hs(nst);
  hipMemcpy(hs.data(), st, nst * 8, hipMemcpyDeviceToHost);
  printf("rows %d dim %d nq %d | queries/wg %d nqb %d streams %d tiles/stream %d | kernel %.1f us (with stamps)\n", rows, dim, nq, qg, nqb, nwg, kp, ms * 1e3);
  const char* names[12] = {"prologue", "stores + transfer issue", "k-loop (MFMA)", "fold", "wait next tile (vmcnt)", "barrier", "epilogue", "", "", "", "", "TOTAL"};
  const size_t nwaves = nst / 12;
  for (int i = 0; i < 12; ++i) {
    if (!names[i][0]) continue;
    std::vector<double> v; for (size_t w = 0; w < nwaves; ++w) v.push_back((double)hs[w * 12 + i]);
    std::sort(v.begin(), v.end());
    double sum = 0; for (double x : v) sum += x;
    printf("  %-28s mean %9.0f  median %9.0f  max %9.0f   per tile %7.0f\n", names[i], sum / nwaves, v[nwaves / 2], v.back(), sum / nwaves / kp);
  }
  return 0;
}
