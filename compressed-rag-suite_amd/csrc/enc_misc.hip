// enc_misc.hip -- K1 (embedding gather + LayerNorm), LayerNorm, K7/K8 (pooling + L2 normalise).
// All HBM-bound row kernels: one wave64 per token row (hidden <= 1024 -> <= 16 elements per lane),
// statistics in fp32 with a two-pass (mean, then centred variance) form, eps inside the sqrt
// exactly as torch.nn.LayerNorm.  Each writes the fp32 residual stream AND the fp16 copy the next
// GEMM reads, so no separate cast pass exists.

#include "enc.h"

namespace crs {
namespace {

constexpr int kMaxPerLane = 16;  // hidden <= 1024 (generic instantiation); hot shapes get exact counts

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
  return x;
}

// v[] holds this lane's strided elements (index c = lane + 64*i); normalise and store
template <int PL>
__device__ __forceinline__ void ln_store(float (&v)[PL], int hidden, int lane, const float* g,
                                         const float* b, float eps, float* x32, _Float16* x16) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < PL; ++i) s += (lane + 64 * i < hidden) ? v[i] : 0.f;
  const float mean = wave_sum(s) / hidden;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < PL; ++i) {
    const float d = v[i] - mean;
    q += (lane + 64 * i < hidden) ? d * d : 0.f;
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / hidden + eps);
#pragma unroll
  for (int i = 0; i < PL; ++i) {
    const int c = lane + 64 * i;
    if (c < hidden) {
      const float o = (v[i] - mean) * rstd * g[c] + b[c];
      x32[c] = o;
      x16[c] = (_Float16)o;
    }
  }
}

// ---- float2 forms (hidden a multiple of 128): a lane owns columns 128 i + 2 lane + {0, 1}.  These kernels are
// bound by vector-memory ISSUE, not bytes -- a wave pays ~100 cycles per load/store instruction whatever its
// width, and the 4-byte form needs 48 of them per token at four split-K partials -- so 8 bytes per lane
// halves their time on the retrieve path (16 bytes would leave a third of the lanes without work at 384).
template <int P2>
__device__ __forceinline__ void ln_store2(float (&v)[P2][2], int hidden, int lane, const float* g, const float* b,
                                          float eps, float* x32, _Float16* x16) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < P2; ++i) s += v[i][0] + v[i][1];
  const float mean = wave_sum(s) / hidden;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < P2; ++i) {
    const float d0 = v[i][0] - mean, d1 = v[i][1] - mean;
    q += d0 * d0 + d1 * d1;
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / hidden + eps);
#pragma unroll
  for (int i = 0; i < P2; ++i) {
    const int c = 128 * i + 2 * lane;
    const float2 gg = *reinterpret_cast<const float2*>(g + c), bb = *reinterpret_cast<const float2*>(b + c);
    float2 o;
    o.x = (v[i][0] - mean) * rstd * gg.x + bb.x;
    o.y = (v[i][1] - mean) * rstd * gg.y + bb.y;
    *reinterpret_cast<float2*>(x32 + c) = o;
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    h2 h = {(_Float16)o.x, (_Float16)o.y};
    *reinterpret_cast<h2*>(x16 + c) = h;
  }
}

template <int P2>
__global__ __launch_bounds__(256) void embed_ln2_kernel(const int* __restrict__ ids, const float* __restrict__ word,
                                                       const float* __restrict__ pos, const float* __restrict__ type0,
                                                       const float* __restrict__ g, const float* __restrict__ b,
                                                       float eps, int tokens, int seq, int hidden, int vocab,
                                                       float* __restrict__ x32, _Float16* __restrict__ x16) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= tokens) return;
  int id = ids[t];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  const float* w = word + (size_t)id * hidden;
  const float* p = pos + (size_t)(t % seq) * hidden;
  float v[P2][2];
#pragma unroll
  for (int i = 0; i < P2; ++i) {
    const int c = 128 * i + 2 * lane;
    const float2 a = *reinterpret_cast<const float2*>(w + c), ty = *reinterpret_cast<const float2*>(type0 + c),
                 pp = *reinterpret_cast<const float2*>(p + c);
    v[i][0] = (a.x + ty.x) + pp.x;   // (word + token_type) + position, as modeling_bert
    v[i][1] = (a.y + ty.y) + pp.y;
  }
  ln_store2<P2>(v, hidden, lane, g, b, eps, x32 + (size_t)t * hidden, x16 + (size_t)t * hidden);
}

template <int P2, int NS>
__global__ __launch_bounds__(256) void layernorm2_kernel(const float* __restrict__ y, const float* __restrict__ bias,
                                                        const float* residual,   // may alias x32 (in place)
                                                        const float* __restrict__ g, const float* __restrict__ b,
                                                        float eps, int tokens, int hidden, float* x32,
                                                        _Float16* __restrict__ x16) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= tokens) return;
  const float* src = y + (size_t)t * hidden;
  const size_t split_stride = (size_t)tokens * hidden;
  float2 part[NS][P2], extra[2][P2];
#pragma unroll
  for (int i = 0; i < P2; ++i) {
    const int c = 128 * i + 2 * lane;
#pragma unroll
    for (int sidx = 0; sidx < NS; ++sidx) part[sidx][i] = *reinterpret_cast<const float2*>(src + sidx * split_stride + c);
    extra[0][i] = bias ? *reinterpret_cast<const float2*>(bias + c) : float2{0.f, 0.f};
    extra[1][i] = residual ? *reinterpret_cast<const float2*>(residual + (size_t)t * hidden + c) : float2{0.f, 0.f};
  }
  float v[P2][2];
#pragma unroll
  for (int i = 0; i < P2; ++i) {
    float a0 = part[0][i].x, a1 = part[0][i].y;
#pragma unroll
    for (int sidx = 1; sidx < NS; ++sidx) { a0 += part[sidx][i].x; a1 += part[sidx][i].y; }
    v[i][0] = (a0 + extra[0][i].x) + extra[1][i].x;
    v[i][1] = (a1 + extra[0][i].y) + extra[1][i].y;
  }
  ln_store2<P2>(v, hidden, lane, g, b, eps, x32 + (size_t)t * hidden, x16 + (size_t)t * hidden);
}

template <int PL>
__global__ __launch_bounds__(256) void embed_ln_kernel(const int* __restrict__ ids, const float* __restrict__ word,
                                                      const float* __restrict__ pos, const float* __restrict__ type0,
                                                      const float* __restrict__ g, const float* __restrict__ b,
                                                      float eps, int tokens, int seq, int hidden, int vocab,
                                                      float* __restrict__ x32, _Float16* __restrict__ x16) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= tokens) return;
  int id = ids[t];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  const float* w = word + (size_t)id * hidden;
  const float* p = pos + (size_t)(t % seq) * hidden;
  float v[PL];
#pragma unroll
  for (int i = 0; i < PL; ++i) {
    const int c = lane + 64 * i;
    v[i] = (c < hidden) ? (w[c] + type0[c]) + p[c] : 0.f;   // (word + token_type) + position, as modeling_bert
  }
  ln_store<PL>(v, hidden, lane, g, b, eps, x32 + (size_t)t * hidden, x16 + (size_t)t * hidden);
}

// y: [NS][tokens][hidden] fp32 split-K partial sums of the preceding GEMM (NS = 1: the full product);
// bias / residual are added here when given, so the GEMM needs no epilogue pass.  NS is a template
// parameter so that every partial-sum load of a row is in flight at once (a runtime loop made hipcc
// wait for each split before issuing the next: one memory round trip per split).
template <int PL, int NS>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ y,
                                                       const float* __restrict__ bias,
                                                       const float* residual,   // may alias x32 (in place)
                                                       const float* __restrict__ g,
                                                       const float* __restrict__ b, float eps, int tokens,
                                                       int hidden, float* x32, _Float16* __restrict__ x16) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= tokens) return;
  const float* src = y + (size_t)t * hidden;
  const size_t split_stride = (size_t)tokens * hidden;
  float part[NS][PL], extra[2][PL];
#pragma unroll
  for (int i = 0; i < PL; ++i) {
    const int c = lane + 64 * i;
    const bool in = c < hidden;
#pragma unroll
    for (int sidx = 0; sidx < NS; ++sidx) part[sidx][i] = in ? src[sidx * split_stride + c] : 0.f;
    extra[0][i] = (in && bias) ? bias[c] : 0.f;
    extra[1][i] = (in && residual) ? residual[(size_t)t * hidden + c] : 0.f;
  }
  float v[PL];
#pragma unroll
  for (int i = 0; i < PL; ++i) {
    float a = part[0][i];
#pragma unroll
    for (int sidx = 1; sidx < NS; ++sidx) a += part[sidx][i];
    v[i] = (a + extra[0][i]) + extra[1][i];
  }
  ln_store<PL>(v, hidden, lane, g, b, eps, x32 + (size_t)t * hidden, x16 + (size_t)t * hidden);
}

// one 256-thread block per sentence: mean over the real tokens (sum / clamp(count, 1e-9)) or the
// [CLS] row, then x / max(||x||, 1e-12).  Thread t owns columns t, t+256, ...; the token loop is
// unrolled so several rows are in flight per thread.
__global__ __launch_bounds__(256) void pool_kernel(const float* __restrict__ x32, const int* __restrict__ lens,
                                                  int batch, int seq, int hidden, int pooling, int normalize,
                                                  float* __restrict__ out, _Float16* __restrict__ out16, int pdim16) {
  __shared__ float red[4];
  __shared__ float part[4][1024];        // mean pooling: one partial row per wave (hidden <= 1024)
  const int bi = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* base = x32 + (size_t)bi * seq * hidden;
  int len = lens[bi];
  len = len < 0 ? 0 : (len > seq ? seq : len);
  float v[4] = {0.f, 0.f, 0.f, 0.f};   // hidden <= 1024
  if (pooling == 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + 256 * i;
      if (c < hidden) v[i] = base[c];
    }
  } else {
    // Mean over the real tokens.  Wave w sums tokens w, w + 4, ...; a lane covers column pairs 2 lane + 128 p (8-byte
    // loads), four tokens unrolled: up to 32 loads of 8 bytes in flight per lane.  (A thread per column walking the
    // tokens four 4-byte loads at a time kept 4 KB in flight per workgroup: 50 us for 256 x 256 x 384 at index build.)
    const int npair = (hidden + 127) >> 7;        // passes of 128 columns: 3 (hidden 384), 6 (768), 8 (1024)
    for (int p = 0; p < npair; ++p) {
      const int c = 2 * lane + 128 * p;
      float2 a[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
      if (c < hidden) {                            // hidden is even (head_dim multiples)
        int t = wave;
        for (; t + 12 < len; t += 16) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float2 x = *reinterpret_cast<const float2*>(base + (size_t)(t + 4 * u) * hidden + c);
            a[u].x += x.x;
            a[u].y += x.y;
          }
        }
        for (; t < len; t += 4) {
          const float2 x = *reinterpret_cast<const float2*>(base + (size_t)t * hidden + c);
          a[0].x += x.x;
          a[0].y += x.y;
        }
        part[wave][c] = (a[0].x + a[1].x) + (a[2].x + a[3].x);
        part[wave][c + 1] = (a[0].y + a[1].y) + (a[2].y + a[3].y);
      }
    }
    __syncthreads();
    const float inv = 1.0f / fmaxf((float)len, 1e-9f);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + 256 * i;
      if (c < hidden) v[i] = ((part[0][c] + part[1][c]) + (part[2][c] + part[3][c])) * inv;
    }
  }
  float scale = 1.f;
  if (normalize) {
    float q = v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    q = wave_sum(q);
    if (lane == 0) red[wave] = q;
    __syncthreads();
    q = red[0] + red[1] + red[2] + red[3];
    scale = 1.0f / fmaxf(sqrtf(q), 1e-12f);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i;
    if (c < hidden) out[(size_t)bi * hidden + c] = v[i] * scale;
    // optional fp16 copy in the scan's query layout (row zero-padded to pdim16): saves the separate
    // normalise + cast launch between the encoder and the scan
    if (out16 && c < pdim16) out16[(size_t)bi * pdim16 + c] = (c < hidden) ? (_Float16)(v[i] * scale) : (_Float16)0.f;
  }
}

}  // namespace

int embed_ln_launch(const int* ids, const float* word, const float* pos, const float* type0, const float* g,
                    const float* b, float eps, int tokens, int seq, int hidden, int vocab, float* x32,
                    _Float16* x16, hipStream_t stream) {
#define CRS_EMB(PL) hipLaunchKernelGGL((embed_ln_kernel<PL>), dim3((tokens + 3) / 4), dim3(256), 0, stream, ids, word, \
                                      pos, type0, g, b, eps, tokens, seq, hidden, vocab, x32, x16)
#define CRS_EMB2(P2) hipLaunchKernelGGL((embed_ln2_kernel<P2>), dim3((tokens + 3) / 4), dim3(256), 0, stream, ids, word, \
                                       pos, type0, g, b, eps, tokens, seq, hidden, vocab, x32, x16)
  if (hidden == 384) CRS_EMB2(3); else if (hidden == 768) CRS_EMB2(6); else if (hidden <= 64) CRS_EMB(1); else CRS_EMB(16);
#undef CRS_EMB2
#undef CRS_EMB
  return (int)hipGetLastError();
}

int layernorm_launch(const float* y, int nsplit, const float* bias, const float* residual, const float* g,
                     const float* b, float eps, int tokens, int hidden, float* x32, _Float16* x16,
                     hipStream_t stream) {
#define CRS_LN2(PL, NS) hipLaunchKernelGGL((layernorm_kernel<PL, NS>), dim3((tokens + 3) / 4), dim3(256), 0, stream, y, \
                                         bias, residual, g, b, eps, tokens, hidden, x32, x16)
#define CRS_LN(PL)                                                                    \
  switch (nsplit) {                                                                   \
    case 1: CRS_LN2(PL, 1); break;                                                    \
    case 2: CRS_LN2(PL, 2); break;                                                    \
    case 3: CRS_LN2(PL, 3); break;                                                    \
    case 4: CRS_LN2(PL, 4); break;                                                    \
    case 6: CRS_LN2(PL, 6); break;                                                    \
    case 8: CRS_LN2(PL, 8); break;                                                    \
    case 16: CRS_LN2(PL, 16); break;                                                  \
    default: return -1;                                                               \
  }
#define CRS_LNV2(P2, NS) hipLaunchKernelGGL((layernorm2_kernel<P2, NS>), dim3((tokens + 3) / 4), dim3(256), 0, stream, y, \
                                          bias, residual, g, b, eps, tokens, hidden, x32, x16)
#define CRS_LNV(P2)                                                                   \
  switch (nsplit) {                                                                   \
    case 1: CRS_LNV2(P2, 1); break;                                                   \
    case 2: CRS_LNV2(P2, 2); break;                                                   \
    case 3: CRS_LNV2(P2, 3); break;                                                   \
    case 4: CRS_LNV2(P2, 4); break;                                                   \
    case 6: CRS_LNV2(P2, 6); break;                                                   \
    case 8: CRS_LNV2(P2, 8); break;                                                   \
    case 16: CRS_LNV2(P2, 16); break;                                                 \
    default: return -1;                                                               \
  }
  if (hidden == 384) { CRS_LNV(3); } else if (hidden == 768) { CRS_LNV(6); } else if (hidden <= 64) { CRS_LN(1); } else { CRS_LN(16); }
#undef CRS_LNV
#undef CRS_LNV2
#undef CRS_LN
#undef CRS_LN2
  return (int)hipGetLastError();
}

int pool_launch(const float* x32, const int* lens, int batch, int seq, int hidden, int pooling, int normalize,
                float* out, _Float16* out16, int pdim16, hipStream_t stream) {
  hipLaunchKernelGGL(pool_kernel, dim3(batch), dim3(256), 0, stream, x32, lens, batch, seq, hidden,
                     pooling, normalize, out, out16, pdim16);
  return (int)hipGetLastError();
}

}  // namespace crs
