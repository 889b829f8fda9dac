// scan_refine.hip -- second stage of the "tile best" scan variants (scan.hip dump mode, scan_wide.hip).
//
// Those kernels keep, per query and tile, only the tile's BEST SCORE, filed under the tile's first row,
// instead of filtering every score against a running threshold.  That is exact, not approximate:
//   order representatives by (best score desc, tile asc) -- merge.hip's (score desc, id asc) with
//   id = first row.  Let x be a row of the final top-k (order: score desc, row asc) and X its tile.  If X
//   were not among the k best tiles, k tiles Y_1..Y_k would precede it; Y_i holds a row y_i with
//   score(y_i) = best(Y_i) > best(X) >= score(x), or with score(y_i) = best(X) >= score(x) and Y_i < X,
//   i.e. row(y_i) < row(x) because tiles are contiguous row ranges -- k distinct rows beat x, a
//   contradiction.  Hence the k best tiles contain every top-k row, and no arg-max is ever needed.
// This kernel re-opens those k tiles per query (16, 32 or 64 rows each), re-scores their rows with the scan's
// own arithmetic (v_mfma_f32_16x16x32_f16, same k-chunk order) and ranks the (<= 1024, typically ~k) candidates by
// counting: one 16-wave workgroup per query, the final sorted (score desc, row asc) top-k comes out directly.

#include "scan.h"

namespace crs {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr float kNegInfR = -__builtin_huge_valf();

// 16 waves per query: wave u re-scores 16-row block u (and u + 16) of the k winning tiles, so the whole
// re-score is at most two dependent global round trips; every load is unconditional (rows past the
// end are clamped and masked afterwards) so that the D/32 loads of a block are all in flight at once.
constexpr int kRefThreads = 1024;

template <int D>
__global__ __launch_bounds__(kRefThreads) void refine_kernel(const _Float16* __restrict__ q16, const _Float16* __restrict__ slab,
                                                            int n_rows, const float* __restrict__ win_s,
                                                            const int64_t* __restrict__ win, int k, int tile_rows, int64_t id_base, float* __restrict__ out_s,
                                                            int64_t* __restrict__ out_i) {
  constexpr int kKs = D / 32;
  __shared__ float cs[1024];   // k * tile_rows <= 16 * 64
  __shared__ int ci[1024];
  __shared__ int cnt;
  const int q = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, kq = lane >> 4;
  if (tid == 0) cnt = 0;
  for (int r = tid; r < k; r += kRefThreads) { out_s[(size_t)q * k + r] = kNegInfR; out_i[(size_t)q * k + r] = -1; }

  const int halves = tile_rows / 16;          // 16-row MFMA blocks per tile: 1, 2 or 4
  const int units = k * halves;               // <= 64
  // Only rows scoring >= the k-th best tile representative can reach the final top-k (that score is
  // attained by k distinct rows already), so the ranking below sees ~k candidates, not k * tile_rows.
  // (a hair below it: scan_wide.hip forms the same dot products with the 32x32x16 MFMA shape, and nothing
  // promises the two shapes round the last bit alike; a lower bar only admits a few more candidates)
  const float t1 = win_s[(size_t)q * k + k - 1];
  const float tau = (win[(size_t)q * k + k - 1] >= 0) ? t1 - (1e-5f * fabsf(t1) + 1e-30f) : kNegInfR;
  // B operand: every column carries query q (lane (n, kq) holds Q[32 ks + 8 kq .. + 8])
  f16x8 qf[kKs];
  const _Float16* qrow = q16 + (size_t)q * D + kq * 8;
#pragma unroll
  for (int ks = 0; ks < kKs; ++ks) qf[ks] = *reinterpret_cast<const f16x8*>(qrow + ks * 32);
  __syncthreads();

  for (int u = wave; u < units; u += kRefThreads / 64) {
    const int j = u / halves, hb = u % halves;
    const int64_t w = win[(size_t)q * k + j];
    const int first = (w < 0) ? 0 : ((int)w / tile_rows) * tile_rows + hb * 16;
    const int row = first + lr;
    const _Float16* arow = slab + (size_t)(row < n_rows ? row : n_rows - 1) * D + kq * 8;
    f16x8 af[kKs];
#pragma unroll
    for (int ks = 0; ks < kKs; ++ks) af[ks] = *reinterpret_cast<const f16x8*>(arow + ks * 32);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < kKs; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[ks], qf[ks], acc, 0, 0, 0);
    // lane (n = lr, kq) holds rows 4 kq + i of column n; column 0 is as good as any
    if (lr == 0 && w >= 0) {
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) {
        const int rr = first + 4 * kq + ii;
        if (rr < n_rows && acc[ii] >= tau) {
          const int p = atomicAdd(&cnt, 1);   // p < units * 16 <= 1024 by construction
          cs[p] = acc[ii];
          ci[p] = rr;
        }
      }
    }
  }
  __syncthreads();
  const int m = cnt;
  for (int c = tid; c < m; c += kRefThreads) {   // rank by counting = output slot
    const float s = cs[c];
    const int id = ci[c];
    int rank = 0;
    for (int o = 0; o < m; ++o) {
      const float so = cs[o];
      const int io = ci[o];
      rank += (so > s || (so == s && io < id)) ? 1 : 0;
    }
    if (rank < k) { out_s[(size_t)q * k + rank] = s; out_i[(size_t)q * k + rank] = (int64_t)id + id_base; }
  }
}

}  // namespace

int refine_launch(const _Float16* q16, int nq, int pdim, const _Float16* slab, int n_rows, const float* win_s,
                  const int64_t* win, int k, int tile_rows, int64_t id_base, float* out_s, int64_t* out_i, hipStream_t stream) {
  if (k > 16 || (tile_rows != 16 && tile_rows != 32 && tile_rows != 64)) return -1;
#define CRS_REFINE(DD) hipLaunchKernelGGL((refine_kernel<DD>), dim3(nq), dim3(kRefThreads), 0, stream, q16, slab, n_rows, win_s, win, k, tile_rows, \
                                         id_base, out_s, out_i)
  switch (pdim) {
    case 128: CRS_REFINE(128); break;
    case 256: CRS_REFINE(256); break;
    case 384: CRS_REFINE(384); break;
    case 512: CRS_REFINE(512); break;
    case 640: CRS_REFINE(640); break;
    case 768: CRS_REFINE(768); break;
    case 896: CRS_REFINE(896); break;
    case 1024: CRS_REFINE(1024); break;
    default: return -1;
  }
#undef CRS_REFINE
  return (int)hipGetLastError();
}

}  // namespace crs
