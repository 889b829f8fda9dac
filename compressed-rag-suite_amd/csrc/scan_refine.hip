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

// TH threads per workgroup: 1024 (16 waves x <= 128 registers) up to 768-element rows; 896 / 1024-element rows keep their
// query fragments (D / 8 registers) in 8 waves x <= 256 registers instead of spilling
template <int D, int TH>
__global__ __launch_bounds__(TH) void refine_kernel(const _Float16* __restrict__ q16, const _Float16* __restrict__ slab,
                                                            int n_rows, const float* __restrict__ win_s,
                                                            const int64_t* __restrict__ win, int k, int tile_rows, int64_t id_base, float* __restrict__ out_s,
                                                            int64_t* __restrict__ out_i) {
  constexpr int kKs = D / 32;
  __shared__ float cs[2048];   // k * tile_rows <= 16 * 64 (chain kernels) or 64 * 32 (dump mode, k <= 64)
  __shared__ int ci[2048];
  __shared__ int cnt;
  const int q = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, kq = lane >> 4;
  if (tid == 0) cnt = 0;
  for (int r = tid; r < k; r += TH) { out_s[(size_t)q * k + r] = kNegInfR; out_i[(size_t)q * k + r] = -1; }

  const int halves = tile_rows / 16;          // 16-row MFMA blocks per tile: 1, 2 or 4
  const int units = k * halves;               // <= 128
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

  for (int u = wave; u < units; u += TH / 64) {
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
          const int p = atomicAdd(&cnt, 1);   // p < units * 16 <= 2048 by construction
          cs[p] = acc[ii];
          ci[p] = rr;
        }
      }
    }
  }
  __syncthreads();
  const int m = cnt;
  for (int c = tid; c < m; c += TH) {   // rank by counting = output slot
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

// ---- int8 slabs (scan_i8.hip, tile-best modes).  The scan's score is exact integer arithmetic up to the final
// scaling: q16 = rint(q / sq), sq = max|q| / 32512, q16 = 256 hi + lo, score = (256 S_hi + S_lo) * sq * s_row
// with S_x = sum_k row[k] x[k] in int32.  This kernel forms the same integers with plain VALU dot products (one
// wave per row, D/64 bytes per lane) and applies the same float operations in the same order, so a re-scored
// row carries bit for bit the score the scan gave it.
template <int D>
__global__ __launch_bounds__(kRefThreads) void refine_i8_kernel(const _Float16* __restrict__ q16, const signed char* __restrict__ slab,
                                                               const float* __restrict__ scales, int n_rows,
                                                               const float* __restrict__ win_s, const int64_t* __restrict__ win,
                                                               int k, int tile_rows, int64_t id_base,
                                                               float* __restrict__ out_s, int64_t* __restrict__ out_i) {
  constexpr int BPL = D / 64;                  // bytes of a row per lane (4, 8, 12 or 16)
  __shared__ float cs[2048];
  __shared__ int ci[2048];
  __shared__ int cnt;
  __shared__ float red[16];
  __shared__ short qhi[D], qlo[D];
  const int q = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) cnt = 0;
  for (int r = tid; r < k; r += kRefThreads) { out_s[(size_t)q * k + r] = kNegInfR; out_i[(size_t)q * k + r] = -1; }
  // query scale: max |q| over the row (a max is order-independent, so this equals the scan's value)
  const _Float16* qrow = q16 + (size_t)q * D;
  float amax = 0.f;
  for (int c = tid; c < D; c += kRefThreads) amax = fmaxf(amax, fabsf((float)qrow[c]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
  if (lane == 0) red[wave] = amax;
  __syncthreads();
  amax = red[0];
#pragma unroll
  for (int w = 1; w < 16; ++w) amax = fmaxf(amax, red[w]);
  const float qscale = amax > 0.f ? amax / 32512.0f : 1.0f;
  for (int c = tid; c < D; c += kRefThreads) {
    const int v = (int)rintf((float)qrow[c] / qscale);
    const int lo = ((v + 128) & 255) - 128;
    qlo[c] = (short)lo;
    qhi[c] = (short)((v - lo) >> 8);
  }
  const float t1 = win_s[(size_t)q * k + k - 1];
  const float tau = (win[(size_t)q * k + k - 1] >= 0) ? t1 : kNegInfR;   // identical arithmetic: no margin needed
  __syncthreads();

  int myhi[BPL], mylo[BPL];                    // this lane's columns lane * BPL .. + BPL - 1, the same for every row
#pragma unroll
  for (int j = 0; j < BPL; ++j) { myhi[j] = qhi[lane * BPL + j]; mylo[j] = qlo[lane * BPL + j]; }
  const int nrows = k * tile_rows;             // <= 64 * 32
  for (int u = wave; u < nrows; u += kRefThreads / 64) {
    const int64_t w = win[(size_t)q * k + u / tile_rows];
    if (w < 0) continue;                        // wave-uniform
    const int row = (int)w + (u % tile_rows);
    if (row >= n_rows) continue;
    const signed char* src = slab + (size_t)row * D + lane * BPL;
    int shi = 0, slo = 0;
#pragma unroll
    for (int wd = 0; wd < BPL / 4; ++wd) {
      const int bits = *reinterpret_cast<const int*>(src + wd * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int x = (bits << (24 - 8 * e)) >> 24;     // sign-extended byte e
        shi += x * myhi[wd * 4 + e];
        slo += x * mylo[wd * 4 + e];
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { shi += __shfl_xor(shi, o); slo += __shfl_xor(slo, o); }
    if (lane == 0) {
      const float sc = ((float)shi * 256.0f + (float)slo) * qscale * scales[row];
      if (sc >= tau) {
        const int p = atomicAdd(&cnt, 1);
        cs[p] = sc;
        ci[p] = row;
      }
    }
  }
  __syncthreads();
  const int m = cnt;
  for (int c = tid; c < m; c += kRefThreads) {
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

int refine_i8_launch(const _Float16* q16, int nq, int pdim, const void* slab, const float* scales, int n_rows, const float* win_s,
                     const int64_t* win, int k, int tile_rows, int64_t id_base, float* out_s, int64_t* out_i, hipStream_t stream) {
  if (k > 64 || tile_rows != 32) return -1;
#define CRS_REFINE8(DD) hipLaunchKernelGGL((refine_i8_kernel<DD>), dim3(nq), dim3(kRefThreads), 0, stream, q16, \
                                          reinterpret_cast<const signed char*>(slab), scales, n_rows, win_s, win, k, tile_rows, id_base, out_s, out_i)
  switch (pdim) {
    case 256: CRS_REFINE8(256); break;
    case 512: CRS_REFINE8(512); break;
    case 768: CRS_REFINE8(768); break;
    case 1024: CRS_REFINE8(1024); break;
    default: return -1;
  }
#undef CRS_REFINE8
  return (int)hipGetLastError();
}

int refine_launch(const _Float16* q16, int nq, int pdim, const _Float16* slab, int n_rows, const float* win_s,
                  const int64_t* win, int k, int tile_rows, int64_t id_base, float* out_s, int64_t* out_i, hipStream_t stream) {
  if (k > 64 || (tile_rows != 16 && tile_rows != 32 && tile_rows != 64) || k * tile_rows > 2048) return -1;
#define CRS_REFINE(DD) hipLaunchKernelGGL((refine_kernel<DD, (DD >= 896 ? 512 : 1024)>), dim3(nq), dim3(DD >= 896 ? 512 : 1024), 0, stream, q16, slab, \
                                         n_rows, win_s, win, k, tile_rows, id_base, out_s, out_i)
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
