// exact.hip -- the exactness certificate behind "Recall@10 = 1.0 vs the fp32 ranking", and the escalation that
// restores it when the certificate fails (the reference stores and ranks fp32: rag/indexing.py:114-119,171-176).
//
// The store over-fetches k' candidates per query from the fp16 / int8 slab (crs_cosine_topk) and re-ranks them by
// their fp32 scores against the fp32 shadow.  That result IS the fp32 top-k of ALL rows whenever no un-fetched row
// can reach it:
//     every un-fetched row j has   slab_score(j) <= t            (t = the k'-th slab score: the scan is exact on its
//                                                                 own scores; + 2e-5 |t| for the last-bit difference
//                                                                 between the scan's and the tile refine's MFMA shape)
//     and                          |slab_score(j) - s32(j)| <= eps_q
// so s32(j) <= t + eps_q =: bound, and the re-ranked list is exact iff its k-th fp32 score is > bound.
// eps_q is a worst-case (Cauchy-Schwarz) bound, per query, from quantities that are MEASURED, not assumed:
//     slab_score - s32 = <q16 - q, c^_j> + <q, c^_j - c_j>  (+ accumulation error)      c^_j = the row as the slab holds it
//     |.| <= dq (1 + E) + |q| E + arith
//     dq    = |q16 - q|_2 computed here from the very two query blocks the scan and the re-rank read
//             (+ sqrt(pdim) max|q16| / 65024 for int8 slabs: the scan moves the query to 16-bit fixed point, scan_i8.hip)
//     E     = max over the shard's rows of |c^_j - c_j|_2, tracked by slab_append (convert.hip) in a device scalar
//     arith = (1.5 pdim + 8) 2^-23: pdim exact products summed in fp32 in any order with truncation (<= pdim 2^-23
//             sum |a_i b_i| <= pdim 2^-23) plus the fp32 FMA chain of the re-rank (<= dim 2^-24 + the butterfly)
// Uncertified queries (status 1) are escalated INSIDE the same stream with no host round trip: collect_above sweeps
// the slab once more for them and lists every row whose slab score is >= (k-th fp32 score so far) - eps_q -- a row of
// the true fp32 top-k cannot score lower -- and refine_list re-ranks that list in fp32.  Both kernels return at once
// when every query of the batch is certified, so they sit in the hipGraph of a batch at the price of two empty launches.
// A list longer than `cap` (more near-identical rows than that) sets status 2: the caller repeats with a larger cap.

#include "scan.h"

namespace crs {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr float kNegInfE = -__builtin_huge_valf();

__device__ __forceinline__ float wsum(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
  return x;
}
__device__ __forceinline__ float wmax(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o));
  return x;
}

// fp32 score of one candidate row, the arithmetic of crs_refine_f32 (convert.hip): lane-strided FMA chain + butterfly
__device__ __forceinline__ float dot_f32(const float* __restrict__ a, const float* __restrict__ b, int dim, int lane) {
  float acc = 0.f;
  for (int e = lane; e < dim; e += 64) acc = fmaf(a[e], b[e], acc);
  return wsum(acc);
}

// the same for up to four rows at once (loads of all rows in flight together: one HBM round trip instead of four);
// per row the identical FMA order as dot_f32, so a candidate's score does not depend on which path scored it
__device__ __forceinline__ void dot4_f32(const float* __restrict__ a, const float* const* __restrict__ rows, int n, int dim, int lane,
                                         float* __restrict__ out) {
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int e = lane; e < dim; e += 64) {
    const float x = a[e];
    float y[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) y[u] = (u < n) ? rows[u][e] : 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = fmaf(x, y[u], acc[u]);
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) out[u] = wsum(acc[u]);
}

// ---- certificate -----------------------------------------------------------------------------------------------
// One 256-thread workgroup per query.  ws_thr / ws_cnt: the escalation workspace's per-query threshold and counter.
__global__ __launch_bounds__(256) void refine_cert_kernel(const float* __restrict__ q32, const _Float16* __restrict__ q16, int dim,
                                                         int pdim, int is_i8, const float* __restrict__ shadow, int64_t n_rows,
                                                         int64_t id_base, const int64_t* __restrict__ cand,
                                                         const float* __restrict__ cand_s, int k_in, int k_out, float err_rows,
                                                         float err_arith, float* __restrict__ out_s, int64_t* __restrict__ out_i,
                                                         int* __restrict__ status, float* __restrict__ ws_thr, int* __restrict__ ws_cnt, int* __restrict__ ws_done) {
  __shared__ float sh_s[64];
  __shared__ int64_t sh_i[64];
  __shared__ float red[4][3];
  __shared__ float kth_s;
  const int qi = blockIdx.x;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float* a = q32 + (size_t)qi * dim;
  const _Float16* a16 = q16 + (size_t)qi * pdim;
  // |q16 - q|^2, |q|^2, max |q16| over the padded row (elements past dim: q = 0)
  float d2 = 0.f, n2 = 0.f, am = 0.f;
  for (int e = t; e < pdim; e += 256) {
    const float x = e < dim ? a[e] : 0.f, h = (float)a16[e];
    d2 = fmaf(h - x, h - x, d2);
    n2 = fmaf(x, x, n2);
    am = fmaxf(am, fabsf(h));
  }
  d2 = wsum(d2); n2 = wsum(n2); am = wmax(am);
  if (lane == 0) { red[wave][0] = d2; red[wave][1] = n2; red[wave][2] = am; }
  // candidate c is scored by wave c & 3, four candidates of a wave at a time (k_in = 16: ONE round trip per wave)
  for (int c0 = wave; c0 < k_in; c0 += 16) {
    const float* rows[4];
    int64_t ids[4];
    bool oks[4];
    int n = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = c0 + 4 * u;
      ids[u] = -1; oks[u] = false; rows[u] = shadow;
      if (c < k_in) {
        n = u + 1;
        ids[u] = cand[(size_t)qi * k_in + c];
        const int64_t row = ids[u] - id_base;
        oks[u] = ids[u] >= 0 && row >= 0 && row < n_rows;
        if (oks[u]) rows[u] = shadow + (size_t)row * dim;
      }
    }
    float sc[4];
    dot4_f32(a, rows, n, dim, lane, sc);
    if (lane == 0) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (c0 + 4 * u < k_in) { sh_s[c0 + 4 * u] = oks[u] ? sc[u] : kNegInfE; sh_i[c0 + 4 * u] = oks[u] ? ids[u] : (int64_t)-1; }
    }
  }
  if (t == 0) kth_s = kNegInfE;
  __syncthreads();
  if (t < k_out) { out_s[(size_t)qi * k_out + t] = kNegInfE; out_i[(size_t)qi * k_out + t] = -1; }
  __syncthreads();
  if (t < k_in) {
    const float s = sh_s[t];
    const int64_t id = sh_i[t];
    if (id >= 0) {
      int rank = 0;
      for (int j = 0; j < k_in; ++j) {
        const float sj = sh_s[j];
        const int64_t ij = sh_i[j];
        rank += (ij >= 0 && (sj > s || (sj == s && (ij < id || (ij == id && j < t))))) ? 1 : 0;
      }
      if (rank < k_out) { out_s[(size_t)qi * k_out + rank] = s; out_i[(size_t)qi * k_out + rank] = id; }
      if (rank == k_out - 1) kth_s = s;
    }
  }
  __syncthreads();
  if (t == 0) {
    const float dd = red[0][0] + red[1][0] + red[2][0] + red[3][0];
    const float nn = red[0][1] + red[1][1] + red[2][1] + red[3][1];
    const float mx = fmaxf(fmaxf(red[0][2], red[1][2]), fmaxf(red[2][2], red[3][2]));
    float dq = sqrtf(dd) * 1.0001f;
    if (is_i8) dq += sqrtf((float)pdim) * mx * (1.0001f / 65024.0f);
    const float eps = dq * (1.0f + err_rows) * 1.0001f + sqrtf(nn) * err_rows * 1.0002f + err_arith;
    // t = the k'-th slab score; fewer than k' valid candidates = every row of the shard was fetched
    int valid = 0;
    float tmin = __builtin_huge_valf();
    for (int c = 0; c < k_in; ++c) {   // (a candidate id outside this shard counts as fetched-but-unusable: never certifies)
      if (cand[(size_t)qi * k_in + c] >= 0) { ++valid; tmin = fminf(tmin, cand_s[(size_t)qi * k_in + c]); }
    }
    const float kth = kth_s;
    int st = 0;
    if (valid == k_in && (int64_t)k_in < n_rows) {
      const float bound = tmin + eps + 2e-5f * fabsf(tmin);
      st = (kth > bound) ? 0 : 1;        // kth == -inf (fewer than k_out candidates, all rows fetched) cannot get here
    }
    status[qi] = st;
    ws_thr[qi] = kth - eps;
    ws_cnt[qi] = 0;
    if (qi == 0) *ws_done = 0;       // the escalation kernel's "blocks through" counter
  }
}

// ---- escalation (one launch): stage 1 lists every row whose slab score reaches the query's threshold, stage 2 re-ranks the lists ---
// Stage 1: plain fp16 MFMA sweep with fragment-shaped global loads (a rare path: no LDS staging, no selection state).
// KS = pdim / 128.  A = 16 slab rows, B = 16 uncertified queries (resident in VGPRs for the sweep), D[row][query].
// Stage 2 (fp32 re-rank of an escalated query's list against the shadow) is done by the LAST block through stage 1 -- release
// fence + counter, acquire fence, no spinning -- one query after the other: escalations are rare, and when NO query needs one
// every block leaves at the top (status is read-only until stage 2), so the common case is a single empty launch instead of
// two.  Dynamic LDS: max(cap * 12, 4 KB) bytes -- stage 1's list of uncertified queries and stage 2's (score, id) list share it.
__device__ void refine_list_one(int qi, char* smem, const float* __restrict__ q32, int dim, const float* __restrict__ shadow,
                                int64_t n_rows, int64_t id_base, int* __restrict__ status, const int* __restrict__ counts,
                                const int64_t* __restrict__ lists, int cap, int k_out, float* __restrict__ out_s,
                                int64_t* __restrict__ out_i) {
  const int n = __hip_atomic_load(&counts[qi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (n > cap) {                       // more rows in the band than the list holds: the caller repeats with a larger cap
    if (t == 0) status[qi] = 2;
    return;
  }
  int64_t* li = reinterpret_cast<int64_t*>(smem);
  float* ls = reinterpret_cast<float*>(smem + (size_t)cap * 8);
  const float* a = q32 + (size_t)qi * dim;
  for (int c0 = wave; c0 < n; c0 += 16) {
    const float* rows[4];
    int64_t ids[4];
    bool oks[4];
    int m = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = c0 + 4 * u;
      ids[u] = -1; oks[u] = false; rows[u] = shadow;
      if (c < n) {
        m = u + 1;
        ids[u] = lists[(size_t)qi * cap + c];
        const int64_t row = ids[u] - id_base;
        oks[u] = row >= 0 && row < n_rows;
        if (oks[u]) rows[u] = shadow + (size_t)row * dim;
      }
    }
    float sc[4];
    dot4_f32(a, rows, m, dim, lane, sc);
    if (lane == 0) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (c0 + 4 * u < n) { ls[c0 + 4 * u] = oks[u] ? sc[u] : kNegInfE; li[c0 + 4 * u] = oks[u] ? ids[u] : (int64_t)-1; }
    }
  }
  if (t < k_out) { out_s[(size_t)qi * k_out + t] = kNegInfE; out_i[(size_t)qi * k_out + t] = -1; }
  __syncthreads();
  for (int c = t; c < n; c += 256) {
    const float s = ls[c];
    const int64_t id = li[c];
    if (id < 0) continue;
    int rank = 0;
    for (int j = 0; j < n && rank < k_out; ++j) {
      const float sj = ls[j];
      const int64_t ij = li[j];
      rank += (ij >= 0 && (sj > s || (sj == s && ij < id))) ? 1 : 0;
    }
    if (rank < k_out) { out_s[(size_t)qi * k_out + rank] = s; out_i[(size_t)qi * k_out + rank] = id; }
  }
  __syncthreads();                     // the list's LDS is reused by the next query
}

template <int KS, bool I8>
__global__ __launch_bounds__(256) void escalate_kernel(const float* __restrict__ q32, const _Float16* __restrict__ q16, int nq, int dim,
                                                      const void* __restrict__ slab_, const float* __restrict__ scales,
                                                      const float* __restrict__ shadow, int n_rows, int64_t id_base,
                                                      int* __restrict__ status, const float* __restrict__ thr, int cap,
                                                      int* __restrict__ counts, int64_t* __restrict__ lists, int* __restrict__ done,
                                                      int k_out, float* __restrict__ out_s, int64_t* __restrict__ out_i) {
  constexpr int D = KS * 128, kSteps = D / 32;
  extern __shared__ __attribute__((aligned(16))) char esc_smem[];
  int* need = reinterpret_cast<int*>(esc_smem);      // stage 1: up to 1024 uncertified queries of a chunk of the batch
  __shared__ int n_need;
  __shared__ int is_last;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lr = lane & 15, kq = lane >> 4;
  bool any = false;
  if (t == 0) n_need = 0;
  __syncthreads();
  for (int q0 = 0; q0 < nq; q0 += 1024) {          // uncertified queries of this chunk of the batch, in any order
    if (q0) { __syncthreads(); if (t == 0) n_need = 0; __syncthreads(); }
    for (int q = q0 + t; q < nq && q < q0 + 1024; q += 256)
      if (status[q] == 1) need[atomicAdd(&n_need, 1)] = q;
    __syncthreads();
    const int nn = n_need;
    any = any || nn > 0;
    for (int g = 0; g < nn; g += 16) {
      const int myq = (g + lr < nn) ? need[g + lr] : -1;
      f16x8 qf[kSteps];
      const _Float16* qrow = q16 + (size_t)(myq < 0 ? 0 : myq) * D + kq * 8;
#pragma unroll
      for (int s = 0; s < kSteps; ++s) qf[s] = *reinterpret_cast<const f16x8*>(qrow + s * 32);
      const float th = myq < 0 ? __builtin_huge_valf() : thr[myq];
      const int n_tiles = (n_rows + 15) / 16;
      for (int tile = blockIdx.x * 4 + wave; tile < n_tiles; tile += gridDim.x * 4) {
        const int row = tile * 16 + lr;
        const int crow = row < n_rows ? row : n_rows - 1;
        f16x8 af[kSteps];
        if (I8) {
          const int8_t* arow = reinterpret_cast<const int8_t*>(slab_) + (size_t)crow * D + kq * 8;
#pragma unroll
          for (int s = 0; s < kSteps; ++s) {
            const int2 raw = *reinterpret_cast<const int2*>(arow + s * 32);
            const int8_t* b8 = reinterpret_cast<const int8_t*>(&raw);
#pragma unroll
            for (int e = 0; e < 8; ++e) af[s][e] = (_Float16)(float)b8[e];
          }
        } else {
          const _Float16* arow = reinterpret_cast<const _Float16*>(slab_) + (size_t)crow * D + kq * 8;
#pragma unroll
          for (int s = 0; s < kSteps; ++s) af[s] = *reinterpret_cast<const f16x8*>(arow + s * 32);
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < kSteps; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[s], qf[s], acc, 0, 0, 0);
        // lane (query column lr, quad kq) holds rows tile * 16 + 4 kq + i
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int rr = tile * 16 + 4 * kq + i;
          float sc = acc[i];
          if (I8) sc *= scales[rr < n_rows ? rr : n_rows - 1];
          if (myq >= 0 && rr < n_rows && sc >= th) {
            const int p = atomicAdd(&counts[myq], 1);
            if (p < cap) lists[(size_t)myq * cap + p] = (int64_t)rr + id_base;
          }
        }
      }
    }
  }
  if (!any) return;      // nothing to escalate: every block sees the same (read-only) status words and leaves here
  // ---- stage 2 by the last block through stage 1 (cdna_hip_programming.md, in-launch hand-off: stores -> vmcnt(0) -> barrier ->
  // lane 0: agent release fence -> vmcnt(0) -> relaxed agent fetch_add; the reader: agent acquire fence -> barrier -> reads)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (t == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    is_last = (__hip_atomic_fetch_add(done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1) ? 1 : 0;
  }
  __syncthreads();
  if (!is_last) return;
  if (t == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    *done = 0;           // (refine_cert_kernel zeroes it as well, ahead of every escalation)
  }
  __syncthreads();
  for (int q = 0; q < nq; ++q)
    if (status[q] == 1)  // (block-uniform: status[q] is written below only for q itself, behind a barrier)
      refine_list_one(q, esc_smem, q32, dim, shadow, (int64_t)n_rows, id_base, status, counts, lists, cap, k_out, out_s, out_i);
}

template <int KS>
int launch_escalate(bool i8, unsigned grid, size_t lds, hipStream_t st, const float* q32, const _Float16* q16, int nq, int dim, const void* slab,
                    const float* scales, const float* shadow, int n_rows, int64_t id_base, int* status, const float* thr, int cap, int* counts,
                    int64_t* lists, int* done, int k_out, float* out_s, int64_t* out_i) {
  const void* kernel = i8 ? reinterpret_cast<const void*>(&escalate_kernel<KS, true>) : reinterpret_cast<const void*>(&escalate_kernel<KS, false>);
  if (lds > 48 * 1024) {     // beyond the default dynamic-LDS limit (up to the CU's 160 KiB: cap <= 13312)
    const hipError_t ae = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (ae != hipSuccess) return (int)ae;
  }
  if (i8)
    hipLaunchKernelGGL((escalate_kernel<KS, true>), dim3(grid), dim3(256), lds, st, q32, q16, nq, dim, slab, scales, shadow, n_rows, id_base,
                       status, thr, cap, counts, lists, done, k_out, out_s, out_i);
  else
    hipLaunchKernelGGL((escalate_kernel<KS, false>), dim3(grid), dim3(256), lds, st, q32, q16, nq, dim, slab, scales, shadow, n_rows, id_base,
                       status, thr, cap, counts, lists, done, k_out, out_s, out_i);
  return (int)hipGetLastError();
}

}  // namespace

float exact_err_arith(int dim, int pdim) { return (1.5f * (float)pdim + 8.0f) * 1.1920929e-7f; }

// Worst-case |stored row - fp32 row|_2 when the store did not track it: fp16 rounds each element to 2^-11 relative
// (2^-25 absolute below the normal range); an int8 row errs by at most scale / 2 = max|x| / 254 <= 1 / 254 per element.
float exact_err_rows_bound(int dim, int slab_type) {
  if (slab_type == 1) return sqrtf((float)dim) / 254.0f * 1.0001f;
  return 4.8828125e-4f * 1.0001f + sqrtf((float)dim) * 2.98023224e-8f;
}

int refine_cert_launch(const float* q32, const _Float16* q16, int nq, int dim, int pdim, int slab_type, const float* shadow,
                       int64_t n_rows, int64_t id_base, const int64_t* cand, const float* cand_s, int k_in, int k_out,
                       float err_rows, float* out_s, int64_t* out_i, int* status, float* ws_thr, int* ws_cnt, int* ws_done, hipStream_t stream) {
  if (nq <= 0) return 0;
  hipLaunchKernelGGL(refine_cert_kernel, dim3(nq), dim3(256), 0, stream, q32, q16, dim, pdim, slab_type == 1 ? 1 : 0, shadow, n_rows,
                     id_base, cand, cand_s, k_in, k_out, err_rows, exact_err_arith(dim, pdim), out_s, out_i, status, ws_thr, ws_cnt, ws_done);
  return (int)hipGetLastError();
}

int escalate_launch(const float* q32, const _Float16* q16, int nq, int dim, int pdim, int slab_type, const void* slab,
                    const float* scales, const float* shadow, int64_t n_rows, int64_t id_base, int k_out, float* out_s,
                    int64_t* out_i, int* status, const float* ws_thr, int* ws_cnt, int* ws_done, int64_t* ws_lists, int cap, int cus,
                    hipStream_t stream) {
  if (nq <= 0) return 0;
  const bool i8 = slab_type == 1;
  const int64_t tiles = (n_rows + 15) / 16;
  int64_t g = (tiles + 3) / 4;
  const int64_t gmax = (int64_t)(cus > 0 ? cus : 256) * 4;
  const unsigned grid = (unsigned)(g < gmax ? (g < 1 ? 1 : g) : gmax);
  const size_t lds = (size_t)cap * 12 > 4096 ? (size_t)cap * 12 : 4096;
#define CRS_ESC(KS_) return launch_escalate<KS_>(i8, grid, lds, stream, q32, q16, nq, dim, slab, scales, shadow, (int)n_rows, id_base, status, \
                                                  ws_thr, cap, ws_cnt, ws_lists, ws_done, k_out, out_s, out_i)
  switch (pdim / 128) {
    case 1: CRS_ESC(1);
    case 2: CRS_ESC(2);
    case 3: CRS_ESC(3);
    case 4: CRS_ESC(4);
    case 5: CRS_ESC(5);
    case 6: CRS_ESC(6);
    case 7: CRS_ESC(7);
    case 8: CRS_ESC(8);
    default: return -1;
  }
#undef CRS_ESC
}

}  // namespace crs
