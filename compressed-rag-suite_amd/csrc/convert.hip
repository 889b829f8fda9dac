// convert.hip -- slab build / query conversion / exact re-score kernels (gfx950, HBM-bound).
//
//   slab_append : fp32 embeddings -> L2-normalised fp16 or int8(+scale) slab rows, optional fp32
//                 shadow.  Replaces collection.add(embeddings=embeddings.tolist(), ...)
//                 (reference rag/indexing.py:114-119): no Python lists, no sqlite, one pass.
//   queries_to_f16 : the same normalise+cast for a query batch (rag/indexing.py:156-168 flatten
//                 + ChromaDB's cosine-space normalisation).
//   rescore     : exact fp32 <q, shadow[id]> for over-fetched candidates + per-row re-sort.
//
// One wave64 per row; rows are <= 1024 elements so a lane owns <= 16 strided elements.

#include "scan.h"

namespace crs {
namespace {

constexpr float kNegInf = -__builtin_huge_valf();

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
  return x;
}
__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o));
  return x;
}

// grid: one wave per row, 4 waves per block
template <bool I8>
__global__ __launch_bounds__(256) void slab_append_kernel(const float* __restrict__ emb, int64_t n,
                                                         int dim, int pdim, void* __restrict__ slab,
                                                         float* __restrict__ scales,
                                                         float* __restrict__ shadow, int64_t row0,
                                                         float* __restrict__ row_err_max) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  const float* src = emb + r * dim;
  float ss = 0.f;
  for (int c = lane; c < dim; c += 64) {
    const float x = src[c];
    ss += x * x;
  }
  ss = wave_sum(ss);
  const float inv_den = fmaxf(sqrtf(ss), 1e-12f);
  const int64_t dr = row0 + r;
  float err2 = 0.f;
  if (I8) {
    float amax = 0.f;
    for (int c = lane; c < dim; c += 64) amax = fmaxf(amax, fabsf(src[c] / inv_den));
    amax = wave_max(amax);
    const float sc = amax / 127.0f;
    const float safe = sc > 0.f ? sc : 1.0f;
    int8_t* dst = reinterpret_cast<int8_t*>(slab) + dr * pdim;
    for (int c = lane; c < pdim; c += 64) {
      float x = 0.f;
      if (c < dim) x = src[c] / inv_den;
      float qv = rintf(x / safe);
      qv = fminf(fmaxf(qv, -127.f), 127.f);
      dst[c] = (int8_t)qv;
      if (shadow && c < dim) shadow[dr * dim + c] = x;
      const float d = x - qv * sc;            // the row as the scan sees it: int8 * scale
      err2 = fmaf(d, d, err2);
    }
    if (lane == 0) scales[dr] = sc;
  } else {
    _Float16* dst = reinterpret_cast<_Float16*>(slab) + dr * pdim;
    for (int c = lane; c < pdim; c += 64) {
      float x = 0.f;
      if (c < dim) x = src[c] / inv_den;
      const _Float16 h = (_Float16)x;
      dst[c] = h;
      if (shadow && c < dim) shadow[dr * dim + c] = x;
      const float d = x - (float)h;
      err2 = fmaf(d, d, err2);
    }
  }
  // |stored row - fp32 row|_2, maximum over the shard's rows: the row term of the exactness certificate (exact.hip).
  // Non-negative floats order like their bit patterns; the plain read first keeps the atomics to the few rows that raise it.
  if (row_err_max) {
    const float err = sqrtf(wave_sum(err2)) * 1.0001f;
    if (lane == 0) {
      int* p = reinterpret_cast<int*>(row_err_max);
      if (__float_as_int(err) > __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(p, __float_as_int(err));
    }
  }
}

// one wave per (query, slot)
__global__ __launch_bounds__(256) void rescore_dot_kernel(const float* __restrict__ q32, int nq,
                                                         int dim, const float* __restrict__ shadow,
                                                         int64_t n_rows, int64_t id_base, int k,
                                                         float* __restrict__ scores,
                                                         const int64_t* __restrict__ ids) {
  const int lane = threadIdx.x & 63;
  const int64_t e = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= (int64_t)nq * k) return;
  const int qi = (int)(e / k);
  const int64_t id = ids[e];
  if (id < 0) {
    if (lane == 0) scores[e] = kNegInf;
    return;
  }
  const int64_t row = id - id_base;
  if (row < 0 || row >= n_rows) return;  // foreign shard: leave untouched (the caller pre-fills)
  const float* a = q32 + (size_t)qi * dim;
  const float* b = shadow + (size_t)row * dim;
  float acc = 0.f;
  for (int c = lane; c < dim; c += 64) acc = fmaf(a[c], b[c], acc);
  acc = wave_sum(acc);
  if (lane == 0) scores[e] = acc;
}

// one wave per query, k <= 64: rank-by-counting sort on (score desc, id asc); empty slots last
__global__ __launch_bounds__(64) void sort_rows_kernel(int nq, int k, float* __restrict__ scores,
                                                      int64_t* __restrict__ ids) {
  const int lane = threadIdx.x;
  const int qi = blockIdx.x;
  const bool in = lane < k;
  float s = in ? scores[(size_t)qi * k + lane] : kNegInf;
  int64_t id = in ? ids[(size_t)qi * k + lane] : -1;
  if (id < 0) s = kNegInf;
  int rank = 0;
  for (int j = 0; j < k; ++j) {
    const float sj = __shfl(s, j);
    const int64_t ij = __shfl(id, j);
    bool before;  // does entry j precede this lane's entry?
    if (ij < 0) before = (id < 0) && (j < lane);
    else if (id < 0) before = true;
    else before = (sj > s) || (sj == s && ij < id);
    rank += before ? 1 : 0;
  }
  if (in) {
    scores[(size_t)qi * k + rank] = s;
    ids[(size_t)qi * k + rank] = id;
  }
}

// Over-fetch re-rank (crs_refine_f32): one 256-thread workgroup per query.  Wave w re-scores candidates
// w, w+4, ... (fp32 FMA over the lane's strided elements, butterfly sum), the scores meet in LDS, and the
// first k_in threads rank themselves by counting (score desc, id asc, empties last); ranks < k_out leave.
__global__ __launch_bounds__(256) void refine_f32_kernel(const float* __restrict__ q32, int dim,
                                                        const float* __restrict__ shadow, int64_t n_rows,
                                                        int64_t id_base, const int64_t* __restrict__ cand,
                                                        int k_in, int k_out, float* __restrict__ out_s,
                                                        int64_t* __restrict__ out_i) {
  __shared__ float sh_s[64];
  __shared__ int64_t sh_i[64];
  const int qi = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* a = q32 + (size_t)qi * dim;
  for (int c = wave; c < k_in; c += 4) {
    int64_t id = cand[(size_t)qi * k_in + c];
    const int64_t row = id - id_base;
    const bool ok = id >= 0 && row >= 0 && row < n_rows;
    float acc = 0.f;
    if (ok) {
      const float* b = shadow + (size_t)row * dim;
      for (int e = lane; e < dim; e += 64) acc = fmaf(a[e], b[e], acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) { sh_s[c] = ok ? acc : kNegInf; sh_i[c] = ok ? id : (int64_t)-1; }
  }
  __syncthreads();
  const int t = threadIdx.x;
  if (t < k_out) { out_s[(size_t)qi * k_out + t] = kNegInf; out_i[(size_t)qi * k_out + t] = -1; }
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
    }
  }
}

}  // namespace

int refine_f32_launch(const float* q32, int nq, int dim, const float* shadow, int64_t n_rows, int64_t id_base,
                      const int64_t* cand, int k_in, int k_out, float* out_s, int64_t* out_i, hipStream_t stream) {
  if (nq <= 0) return 0;
  hipLaunchKernelGGL(refine_f32_kernel, dim3(nq), dim3(256), 0, stream, q32, dim, shadow, n_rows, id_base, cand,
                     k_in, k_out, out_s, out_i);
  return (int)hipGetLastError();
}

int slab_append_launch(const float* emb, int64_t n, int dim, int pdim, int slab_type, void* slab,
                       float* scales, float* shadow, int64_t row0, float* row_err_max, hipStream_t stream) {
  if (n <= 0) return 0;
  const unsigned blocks = (unsigned)((n + 3) / 4);
  if (slab_type == 1)
    hipLaunchKernelGGL((slab_append_kernel<true>), dim3(blocks), dim3(256), 0, stream, emb, n, dim,
                       pdim, slab, scales, shadow, row0, row_err_max);
  else
    hipLaunchKernelGGL((slab_append_kernel<false>), dim3(blocks), dim3(256), 0, stream, emb, n, dim,
                       pdim, slab, scales, shadow, row0, row_err_max);
  return (int)hipGetLastError();
}

int queries_to_f16_launch(const float* q, int nq, int dim, int pdim, _Float16* out,
                          hipStream_t stream) {
  return slab_append_launch(q, nq, dim, pdim, 0, out, nullptr, nullptr, 0, nullptr, stream);
}

// fp32 scores of arbitrary candidate lists (any k): out[e] = <q32[e / k], shadow[ids[e] - id_base]>, -inf for ids < 0
int score_rows_launch(const float* q32, int nq, int dim, const float* shadow, int64_t n_rows, int64_t id_base, int k, const int64_t* ids,
                      float* scores, hipStream_t stream) {
  const int64_t e = (int64_t)nq * k;
  if (e <= 0) return 0;
  hipLaunchKernelGGL(rescore_dot_kernel, dim3((unsigned)((e + 3) / 4)), dim3(256), 0, stream, q32, nq, dim, shadow, n_rows, id_base, k,
                     scores, ids);
  return (int)hipGetLastError();
}

int rescore_launch(const float* q32, int nq, int dim, const float* shadow, int64_t n_rows,
                   int64_t id_base, int k, float* scores, int64_t* ids, hipStream_t stream) {
  const int64_t e = (int64_t)nq * k;
  if (e <= 0) return 0;
  hipLaunchKernelGGL(rescore_dot_kernel, dim3((unsigned)((e + 3) / 4)), dim3(256), 0, stream, q32,
                     nq, dim, shadow, n_rows, id_base, k, scores, ids);
  hipLaunchKernelGGL(sort_rows_kernel, dim3(nq), dim3(64), 0, stream, nq, k, scores, ids);
  return (int)hipGetLastError();
}

}  // namespace crs
