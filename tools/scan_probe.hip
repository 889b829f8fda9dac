// scan_probe.hip -- diagnostic build of the scan kernel with in-kernel s_memtime stamps
// (cdna_hip_programming.md section 7, "In-kernel stamps").  Never part of libcrs_hip.so.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCRS_STAMPS -I../compressed-rag-suite_amd/csrc -o scan_probe scan_probe.hip
//   ./scan_probe <rows> <dim 384> <nq> <k> <variant 0|3>
#define CRS_STAMPS 1
#include "../compressed-rag-suite_amd/csrc/scan.hip"
#include "../compressed-rag-suite_amd/csrc/scan_i8.hip"

#include <algorithm>
#include <stdio.h>
#include <vector>

int main(int argc, char** argv) {
  const int rows = argc > 1 ? atoi(argv[1]) : 100000;
  const int dim = argc > 2 ? atoi(argv[2]) : 384;
  const int nq = argc > 3 ? atoi(argv[3]) : 64;
  const int k = argc > 4 ? atoi(argv[4]) : 10;
  if (argc > 5) setenv("CRS_SCAN_VARIANT", argv[5], 1);
  const int tr = crs::scan_tile_rows(dim);
  const int n_tiles = (rows + tr - 1) / tr;
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int nwg = std::min(n_tiles, prop.multiProcessorCount * crs::scan_wg_per_cu());
  std::vector<_Float16> h((size_t)rows * dim), hq((size_t)nq * dim);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (auto& x : h) x = (_Float16)(rnd() * 0.1f);
  for (auto& x : hq) x = (_Float16)(rnd() * 0.1f);
  _Float16 *slab, *q; float* ps; int* pr; unsigned long long* st;
  hipMalloc(&slab, h.size() * 2); hipMalloc(&q, hq.size() * 2);
  hipMalloc(&ps, (size_t)nwg * nq * 64 * 4); hipMalloc(&pr, (size_t)nwg * nq * 64 * 4);
  hipMalloc(&st, (size_t)nwg * 4 * 64 * 8);
  hipMemcpy(slab, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(q, hq.data(), hq.size() * 2, hipMemcpyHostToDevice);
  unsigned* tau; hipMalloc(&tau, nq * 4);
  crs::ScanArgs a{};
  a.q = q; a.slab = slab; a.scales = nullptr; a.part_scores = ps; a.part_rows = pr; a.tau_shared = crs::scan_share_tau() ? tau : nullptr;
  a.stamps = st; a.n_rows = rows; a.n_tiles = n_tiles; a.nq = nq; a.k = k; a.nqb = (nq + 63) / 64; a.nwg = nwg; a.kp = k; a.sched = getenv("CRS_SCAN_SCHED") ? atoi(getenv("CRS_SCAN_SCHED")) : 1; a.boot = getenv("CRS_SCAN_BOOT") && getenv("CRS_SCAN_BOOT")[0] == '0' ? 0 : 1;
  for (int rep = 0; rep < 3; ++rep) {
    hipMemset(st, 0, (size_t)nwg * 4 * 64 * 8);
    hipMemset(tau, 0, nq * 4);
    int e = crs::scan_launch_f16(a, dim, nwg, 0);
    hipDeviceSynchronize();
    if (e) { printf("launch error %d\n", e); return 1; }
  }
  std::vector<unsigned long long> hs((size_t)nwg * 4 * 64);
  hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
  // wave 0 of every workgroup
  auto col = [&](int slot) { std::vector<double> v; for (int b = 0; b < nwg; ++b) v.push_back((double)hs[((size_t)b * 4) * 64 + slot]); return v; };
  auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  auto mx = [](std::vector<double> v) { return *std::max_element(v.begin(), v.end()); };
  auto mn = [](std::vector<double> v) { return *std::min_element(v.begin(), v.end()); };
  auto diff = [&](int a_, int b_) { auto x = col(a_), y = col(b_); std::vector<double> d; for (size_t i = 0; i < x.size(); ++i) if (x[i] && y[i]) d.push_back(y[i] - x[i]); return d; };
  printf("rows %d dim %d nq %d k %d  nwg %d tiles/wg %.2f variant %d\n", rows, dim, nq, k, nwg, (double)n_tiles / nwg, crs::scan_variant());
  auto r0 = col(62), r1 = col(63);
  const double t0 = mn(r0);
  std::vector<double> starts, ends; for (int b = 0; b < nwg; ++b) { starts.push_back((r0[b] - t0) * 10.0); ends.push_back((r1[b] - t0) * 10.0); }
  printf("wg start (ns after first): median %.0f max %.0f | wg end: median %.0f max %.0f  (memrealtime 100 MHz)\n", med(starts), mx(starts), med(ends), mx(ends));
  printf("cycles (s_memtime), median over workgroups [max]:\n");
  auto pr2 = [&](const char* name, int a_, int b_) { auto d = diff(a_, b_); if (!d.empty()) printf("  %-28s %8.0f [%8.0f]  n=%zu\n", name, med(d), mx(d), d.size()); };
  pr2("query fragments (0->1)", 0, 1);
  pr2("first tile load+park (1->2)", 1, 2);
  for (int it = 0; it < 14; ++it) {
    char nm[64];
    snprintf(nm, 64, "tile %d issue+math", it); pr2(nm, it == 0 ? 2 : 5 + 3 * (it - 1), 3 + 3 * it);
    snprintf(nm, 64, "tile %d wait+park", it); pr2(nm, 3 + 3 * it, 4 + 3 * it);
    snprintf(nm, 64, "tile %d barrier", it); pr2(nm, 4 + 3 * it, 5 + 3 * it);
  }
  pr2("final compaction+store", 58, 59);
  pr2("whole kernel (0->59)", 0, 59);
  {
    // the slowest workgroups (wave 0 of each): what did they do differently?
    std::vector<std::pair<double,int>> order;
    for (int b = 0; b < nwg; ++b) order.push_back({(double)(hs[((size_t)b * 4) * 64 + 59] - hs[((size_t)b * 4) * 64 + 0]), b});
    std::sort(order.begin(), order.end());
    auto show = [&](int b) {
      const unsigned long long* r = &hs[((size_t)b * 4) * 64];
      int tiles = 0; for (int it = 0; it < 14; ++it) if (r[3 + 3 * it]) ++tiles;
      printf("    wg %4d  total %7.0f  start+%5.0f ns  qfrag %6.0f  tiles(stamped) %2d  compactions %llu (%6llu cyc)  final %6.0f  wait/park sum %6.0f  barrier sum %6.0f\n", b,
             (double)(r[59] - r[0]), (double)(r[62] - (unsigned long long)t0) * 10.0, (double)(r[1] - r[0]), tiles, r[60], r[61], (double)(r[59] - r[58]),
             [&]{ double s_ = 0; for (int it = 0; it < 14; ++it) if (r[4 + 3 * it]) s_ += (double)(r[4 + 3 * it] - r[3 + 3 * it]); return s_; }(),
             [&]{ double s_ = 0; for (int it = 0; it < 14; ++it) if (r[5 + 3 * it]) s_ += (double)(r[5 + 3 * it] - r[4 + 3 * it]); return s_; }());
    };
    printf("  fastest 4 / median 2 / slowest 8 workgroups:\n");
    for (int i = 0; i < 4; ++i) show(order[i].second);
    show(order[nwg / 2].second); show(order[nwg / 2 + 1].second);
    for (int i = nwg - 8; i < nwg; ++i) show(order[i].second);
  }
  printf("  compactions per wave: median %.0f max %.0f ; cycles in compaction: median %.0f max %.0f\n", med(col(60)), mx(col(60)), med(col(61)), mx(col(61)));
  return 0;
}
