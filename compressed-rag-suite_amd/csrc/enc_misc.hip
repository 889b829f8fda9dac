// enc_misc.hip -- K1 (embedding gather + LayerNorm), LayerNorm, K7/K8 (pooling + L2 normalise).
// All HBM-bound row kernels: one wave64 per token row (hidden <= 1024 -> <= 16 elements per lane),
// statistics in fp32 with a two-pass (mean, then centred variance) form, eps inside the sqrt
// exactly as torch.nn.LayerNorm.  Each writes the fp32 residual stream AND the fp16 copy the next
// GEMM reads, so no separate cast pass exists.

#include "enc.h"

namespace crs {
namespace {

constexpr int kMaxPerLane = 16;  // hidden <= 1024

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
  return x;
}

// v[] holds this lane's strided elements (index c = lane + 64*i); normalise and store
__device__ __forceinline__ void ln_store(float (&v)[kMaxPerLane], int hidden, int lane, const float* g,
                                         const float* b, float eps, float* x32, _Float16* x16) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < kMaxPerLane; ++i) s += (lane + 64 * i < hidden) ? v[i] : 0.f;
  const float mean = wave_sum(s) / hidden;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < kMaxPerLane; ++i) {
    const float d = v[i] - mean;
    q += (lane + 64 * i < hidden) ? d * d : 0.f;
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / hidden + eps);
#pragma unroll
  for (int i = 0; i < kMaxPerLane; ++i) {
    const int c = lane + 64 * i;
    if (c < hidden) {
      const float o = (v[i] - mean) * rstd * g[c] + b[c];
      x32[c] = o;
      x16[c] = (_Float16)o;
    }
  }
}

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
  float v[kMaxPerLane];
#pragma unroll
  for (int i = 0; i < kMaxPerLane; ++i) {
    const int c = lane + 64 * i;
    v[i] = (c < hidden) ? (w[c] + type0[c]) + p[c] : 0.f;   // (word + token_type) + position, as modeling_bert
  }
  ln_store(v, hidden, lane, g, b, eps, x32 + (size_t)t * hidden, x16 + (size_t)t * hidden);
}

__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ y, const float* __restrict__ g,
                                                       const float* __restrict__ b, float eps, int tokens,
                                                       int hidden, float* __restrict__ x32,
                                                       _Float16* __restrict__ x16) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= tokens) return;
  const float* src = y + (size_t)t * hidden;
  float v[kMaxPerLane];
#pragma unroll
  for (int i = 0; i < kMaxPerLane; ++i) {
    const int c = lane + 64 * i;
    v[i] = (c < hidden) ? src[c] : 0.f;
  }
  ln_store(v, hidden, lane, g, b, eps, x32 + (size_t)t * hidden, x16 + (size_t)t * hidden);
}

// one wave per sentence: mean over the real tokens (sum / clamp(count, 1e-9)) or the [CLS] row,
// then x / max(||x||, 1e-12)
__global__ __launch_bounds__(256) void pool_kernel(const float* __restrict__ x32, const int* __restrict__ lens,
                                                  int batch, int seq, int hidden, int pooling, int normalize,
                                                  float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int bi = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (bi >= batch) return;
  const float* base = x32 + (size_t)bi * seq * hidden;
  int len = lens[bi];
  len = len < 0 ? 0 : (len > seq ? seq : len);
  float v[kMaxPerLane];
#pragma unroll
  for (int i = 0; i < kMaxPerLane; ++i) v[i] = 0.f;
  if (pooling == 1) {
#pragma unroll
    for (int i = 0; i < kMaxPerLane; ++i) {
      const int c = lane + 64 * i;
      if (c < hidden) v[i] = base[c];
    }
  } else {
    for (int s = 0; s < len; ++s) {
#pragma unroll
      for (int i = 0; i < kMaxPerLane; ++i) {
        const int c = lane + 64 * i;
        if (c < hidden) v[i] += base[(size_t)s * hidden + c];
      }
    }
    const float den = fmaxf((float)len, 1e-9f);
#pragma unroll
    for (int i = 0; i < kMaxPerLane; ++i) v[i] /= den;
  }
  float scale = 1.f;
  if (normalize) {
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxPerLane; ++i) q += v[i] * v[i];
    scale = 1.0f / fmaxf(sqrtf(wave_sum(q)), 1e-12f);
  }
#pragma unroll
  for (int i = 0; i < kMaxPerLane; ++i) {
    const int c = lane + 64 * i;
    if (c < hidden) out[(size_t)bi * hidden + c] = v[i] * scale;
  }
}

}  // namespace

int embed_ln_launch(const int* ids, const float* word, const float* pos, const float* type0, const float* g,
                    const float* b, float eps, int tokens, int seq, int hidden, int vocab, float* x32,
                    _Float16* x16, hipStream_t stream) {
  hipLaunchKernelGGL(embed_ln_kernel, dim3((tokens + 3) / 4), dim3(256), 0, stream, ids, word, pos, type0, g, b,
                     eps, tokens, seq, hidden, vocab, x32, x16);
  return (int)hipGetLastError();
}

int layernorm_launch(const float* y, const float* g, const float* b, float eps, int tokens, int hidden,
                     float* x32, _Float16* x16, hipStream_t stream) {
  hipLaunchKernelGGL(layernorm_kernel, dim3((tokens + 3) / 4), dim3(256), 0, stream, y, g, b, eps, tokens, hidden,
                     x32, x16);
  return (int)hipGetLastError();
}

int pool_launch(const float* x32, const int* lens, int batch, int seq, int hidden, int pooling, int normalize,
                float* out, hipStream_t stream) {
  hipLaunchKernelGGL(pool_kernel, dim3((batch + 3) / 4), dim3(256), 0, stream, x32, lens, batch, seq, hidden,
                     pooling, normalize, out);
  return (int)hipGetLastError();
}

}  // namespace crs
