// enc_attn.hip -- K3: padding-masked, non-causal self-attention for the encoder (gfx950).
//
//   ctx[b, s, h*hd : (h+1)*hd] = softmax(Q K^T / sqrt(hd) + mask) V        (per batch b, head h)
//
// Flash-style: one 256-thread workgroup per (64 query rows, head, batch); wave w owns 16 query
// rows and keeps their Q fragments, running max / sum and the O accumulators in registers.
// Keys are visited in blocks of 64: K is staged row-major and V TRANSPOSED in LDS so that both
// MFMA B operands are 8-byte contiguous reads; scores never leave the chip.  S = Q K^T and
// O += P V run on v_mfma_f32_16x16x16_f16 (head_dim 16 / 32 / 64 are whole multiples of its K);
// P goes through a per-wave LDS tile to turn the accumulator layout into the A-operand layout.
// Softmax is fp32 with the usual online rescaling.  Sequences are right-padded: keys >= lens[b]
// get -1e30 before the max (every row sees key 0, so the max is always finite).
// <= 11 % of the encoder's FLOPs at the BASELINE shapes.

#include "enc.h"

namespace crs {
namespace {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int KB = 64;   // keys per block
constexpr int QB = 64;   // query rows per workgroup

template <int HD>
__global__ __launch_bounds__(kThreads) void attention_kernel(const _Float16* __restrict__ qkv,
                                                            const int* __restrict__ lens,
                                                            _Float16* __restrict__ ctx, int seq, int hidden) {
  constexpr int KS = HD / 16;          // k-steps of the QK^T contraction
  constexpr int NT = HD / 16;          // 16-wide output column tiles of O
  constexpr int KROW = HD + 4;         // padded K row (halves)
  constexpr int VROW = KB + 4;         // padded V^T row (halves)
  constexpr int PROW = KB + 4;
  __shared__ __attribute__((aligned(16))) _Float16 sK[KB * KROW];
  __shared__ __attribute__((aligned(16))) _Float16 sVt[HD * VROW];
  __shared__ __attribute__((aligned(16))) _Float16 sP[4 * 16 * PROW];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, g = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * QB + wave * 16;
  const int len = min(max(lens[b], 1), seq);
  const size_t row_stride = (size_t)3 * hidden;
  const _Float16* base = qkv + (size_t)b * seq * row_stride + h * HD;
  const float scale = 1.0f / sqrtf((float)HD);

  // Q fragments: A[row lr][k = 4g + j + 16 ks]
  f16x4 qf[KS];
  {
    const int qr = q0 + lr;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      f16x4 z = {0, 0, 0, 0};
      qf[ks] = (qr < seq) ? *reinterpret_cast<const f16x4*>(base + (size_t)qr * row_stride + ks * 16 + g * 4) : z;
    }
  }
  f32x4 o[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) o[n] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run[4], l_run[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { m_run[i] = -1e30f; l_run[i] = 0.f; }

  _Float16* myP = sP + wave * 16 * PROW;
  for (int kb = 0; kb < len; kb += KB) {
    __syncthreads();  // previous block's K / V^T fully consumed
    // ---- stage K (row-major) and V (transposed) for keys kb .. kb+63
    constexpr int CH = HD / 8;  // 16-byte chunks per key row
    for (int id = tid; id < KB * CH; id += kThreads) {
      const int key = id / CH, c = id % CH;
      const int kr = kb + key;
      f16x8 kv = {0, 0, 0, 0, 0, 0, 0, 0}, vv = kv;
      if (kr < seq) {
        const _Float16* p = base + (size_t)kr * row_stride + c * 8;
        kv = *reinterpret_cast<const f16x8*>(p + hidden);
        vv = *reinterpret_cast<const f16x8*>(p + 2 * hidden);
      }
      *reinterpret_cast<f16x4*>(&sK[key * KROW + c * 8]) = f16x4{kv[0], kv[1], kv[2], kv[3]};
      *reinterpret_cast<f16x4*>(&sK[key * KROW + c * 8 + 4]) = f16x4{kv[4], kv[5], kv[6], kv[7]};
#pragma unroll
      for (int e = 0; e < 8; ++e) sVt[(c * 8 + e) * VROW + key] = vv[e];
    }
    __syncthreads();

    // ---- S = Q K^T for 4 column tiles of 16 keys
    f32x4 s[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      s[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const f16x4 kf = *reinterpret_cast<const f16x4*>(&sK[(ct * 16 + lr) * KROW + ks * 16 + g * 4]);
        s[ct] = __builtin_amdgcn_mfma_f32_16x16x16f16(qf[ks], kf, s[ct], 0, 0, 0);
      }
    }
    // lane holds rows 4g+i, key column kb + ct*16 + lr
    float mx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) mx[i] = -1e30f;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const bool valid = (kb + ct * 16 + lr) < len;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        s[ct][i] = valid ? s[ct][i] * scale : -1e30f;
        mx[i] = fmaxf(mx[i], s[ct][i]);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) mx[i] = fmaxf(mx[i], __shfl_xor(mx[i], off));
    }
    float alpha[4], rs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float mn = fmaxf(m_run[i], mx[i]);
      alpha[i] = __expf(m_run[i] - mn);
      m_run[i] = mn;
      rs[i] = 0.f;
    }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float p = __expf(s[ct][i] - m_run[i]);   // masked keys: exp(-1e30 - m) = 0
        rs[i] += p;
        myP[(4 * g + i) * PROW + ct * 16 + lr] = (_Float16)p;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) rs[i] += __shfl_xor(rs[i], off);
      l_run[i] = l_run[i] * alpha[i] + rs[i];
    }
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int i = 0; i < 4; ++i) o[n][i] *= alpha[i];
    // ---- O += P V   (A = P[row lr][key 4g+j+16ks], B = V^T[col n*16+lr][key 4g+j+16ks]); wave-local LDS tile
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ks = 0; ks < KB / 16; ++ks) {
      const f16x4 pf = *reinterpret_cast<const f16x4*>(&myP[lr * PROW + ks * 16 + g * 4]);
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const f16x4 vf = *reinterpret_cast<const f16x4*>(&sVt[(n * 16 + lr) * VROW + ks * 16 + g * 4]);
        o[n] = __builtin_amdgcn_mfma_f32_16x16x16f16(pf, vf, o[n], 0, 0, 0);
      }
    }
  }
  // ---- normalise and store: rows 4g+i, columns n*16 + lr
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int qr = q0 + 4 * g + i;
    if (qr >= seq) continue;
    const float inv = 1.0f / l_run[i];
    _Float16* dst = ctx + ((size_t)b * seq + qr) * hidden + h * HD;
#pragma unroll
    for (int n = 0; n < NT; ++n) dst[n * 16 + lr] = (_Float16)(o[n][i] * inv);
  }
}

}  // namespace

int attention_launch(const _Float16* qkv, const int* lens, _Float16* ctx, int batch, int seq, int hidden,
                     int heads, hipStream_t stream) {
  const int hd = hidden / heads;
  dim3 grid((seq + QB - 1) / QB, heads, batch);
  switch (hd) {
    case 16: hipLaunchKernelGGL((attention_kernel<16>), grid, dim3(kThreads), 0, stream, qkv, lens, ctx, seq, hidden); break;
    case 32: hipLaunchKernelGGL((attention_kernel<32>), grid, dim3(kThreads), 0, stream, qkv, lens, ctx, seq, hidden); break;
    case 64: hipLaunchKernelGGL((attention_kernel<64>), grid, dim3(kThreads), 0, stream, qkv, lens, ctx, seq, hidden); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}

}  // namespace crs
