// enc_attn.hip -- K3: padding-masked, non-causal self-attention for the encoder (gfx950).
//
//   ctx[b, s, h*hd : (h+1)*hd] = softmax(Q K^T / sqrt(hd) + mask) V        (per batch b, head h)
//
// Flash-style, computed transposed (S^T = K Q^T, O^T = V^T P^T: see the kernel): one 256-thread workgroup per
// (64 query rows, head, batch); wave w owns 16 query rows and keeps their Q fragments, running max / sum and the
// O^T accumulators in registers.
// Keys are visited in blocks of 64: K is staged row-major and V TRANSPOSED in LDS so that both
// MFMA B operands are 8-byte contiguous reads; scores never leave the chip.  S = Q K^T and
// O += P V run on v_mfma_f32_16x16x16_f16 (head_dim 16 / 32 / 64 are whole multiples of its K);
// P goes through a per-wave LDS tile to turn the accumulator layout into the A-operand layout.
// Softmax is fp32 with the usual online rescaling.  Sequences are right-padded: keys >= lens[b]
// get -1e30 before the max (every row sees key 0, so the max is always finite).
// <= 11 % of the encoder's FLOPs at the BASELINE shapes.

#include "enc.h"

#include <stdlib.h>

namespace crs {
namespace {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int KB = 64;   // keys per block
constexpr int QB = 64;   // query rows per workgroup

// One online-softmax step for this lane's query over the 16 scores it holds of a 64-key block (raw Q.K products in s).
// Works in the base-2 domain with the 1/sqrt(hd) scale folded into one constant c = log2(e) / sqrt(hd): per score a
// max on the raw product, one fma and one v_exp_f32 (the softmax, not the MFMAs, bounds this kernel: 16 scores per
// lane and block against 16 small MFMAs per wave).  Keys >= len (right padding, last block only) are set to -1e30
// before the max, so their weight is exp2(-huge) = 0; every query sees key 0, so the running max is finite.
__device__ __forceinline__ void softmax_step(f32x4 (&s)[4], int key0, int len, float c, float& m_run, float& l_run,
                                             float& alpha, f16x4 (&pf)[4]) {
  float mx = -1e30f;
  if (key0 - (key0 & 15) + KB > len) {     // block reaches past len (wave-uniform: key0 = kb + 4 g)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (key0 + ct * 16 + i >= len) s[ct][i] = -1e30f;
  }
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int i = 0; i < 4; ++i) mx = fmaxf(mx, s[ct][i]);
  mx = fmaxf(mx, __shfl_xor(mx, 16));
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  const float mn = fmaxf(m_run, mx * c);
  alpha = __builtin_amdgcn_exp2f(m_run - mn);
  m_run = mn;
  float rs = 0.f;
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float p = __builtin_amdgcn_exp2f(fmaf(s[ct][i], c, -mn));
      rs += p;
      pf[ct][i] = (_Float16)p;
    }
  }
  rs += __shfl_xor(rs, 16);
  rs += __shfl_xor(rs, 32);
  l_run = l_run * alpha + rs;
}

template <int HD>
__global__ __launch_bounds__(kThreads) void attention_kernel(const _Float16* __restrict__ qkv,
                                                            const int* __restrict__ lens,
                                                            _Float16* __restrict__ ctx, int seq, int hidden) {
  constexpr int KS = HD / 16;          // k-steps of the Q K^T contraction
  constexpr int NT = HD / 16;          // 16-row tiles of O^T (head-dim index)
  constexpr int KROW = HD + 4;         // padded K row (halves)
  constexpr int VROW = KB + 4;         // padded V^T row (halves)
  __shared__ __attribute__((aligned(16))) _Float16 sK[KB * KROW];
  __shared__ __attribute__((aligned(16))) _Float16 sVt[HD * VROW];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, g = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * QB + wave * 16;
  const int len = min(max(lens[b], 1), seq);
  const size_t row_stride = (size_t)3 * hidden;
  const _Float16* base = qkv + (size_t)b * seq * row_stride + h * HD;
  const float scale = 1.44269504088896341f / sqrtf((float)HD);   // log2(e) / sqrt(hd): softmax_step works in base 2

  // Everything is computed TRANSPOSED: S^T = K Q^T and O^T = V^T P^T.  With the 16x16x16 accumulator layout
  // (column = lane & 15, rows 4 (lane >> 4) + i) a lane then holds the scores of ONE query (lane & 15) for 16
  // keys per block, so the softmax row reductions are 15 register ops + two cross-lane steps instead of four
  // shuffle steps per row, the per-query rescaling is a per-lane scalar, and P^T leaves the accumulators in
  // exactly the B-operand layout of the second product (4 consecutive keys of one query): no LDS round trip
  // for P.  Q fragments (B operand of S^T): lane holds Q[query lr][hd 16 ks + 4 g .. + 4].
  f16x4 qf[KS];
  {
    const int qr = q0 + lr;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      f16x4 z = {0, 0, 0, 0};
      qf[ks] = (qr < seq) ? *reinterpret_cast<const f16x4*>(base + (size_t)qr * row_stride + ks * 16 + g * 4) : z;
    }
  }
  f32x4 o[NT];                          // O^T tile n: rows = head-dim 16 n + 4 g + i, column = query lr
#pragma unroll
  for (int n = 0; n < NT; ++n) o[n] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -1e30f, l_run = 0.f;   // of this lane's query

  for (int kb = 0; kb < len; kb += KB) {
    __syncthreads();  // previous block's K / V^T fully consumed
    // ---- stage K (row-major): one 16-byte chunk per thread and pass
    constexpr int CH = HD / 8;  // 16-byte chunks per key row
    for (int id = tid; id < KB * CH; id += kThreads) {
      const int key = id / CH, c = id % CH;
      const int kr = kb + key;
      f16x8 kv = {0, 0, 0, 0, 0, 0, 0, 0};
      if (kr < seq) kv = *reinterpret_cast<const f16x8*>(base + (size_t)kr * row_stride + c * 8 + hidden);
      *reinterpret_cast<f16x4*>(&sK[key * KROW + c * 8]) = f16x4{kv[0], kv[1], kv[2], kv[3]};
      *reinterpret_cast<f16x4*>(&sK[key * KROW + c * 8 + 4]) = f16x4{kv[4], kv[5], kv[6], kv[7]};
    }
    // ---- stage V transposed: a thread takes 4 consecutive keys x 8 head-dim columns and writes eight
    // 8-byte pieces (4 keys of one column) instead of 32 two-byte stores
    for (int id = tid; id < (KB / 4) * CH; id += kThreads) {
      const int kg = id / CH, c = id % CH;
      f16x8 vv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kr = kb + kg * 4 + j;
        const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        vv[j] = (kr < seq) ? *reinterpret_cast<const f16x8*>(base + (size_t)kr * row_stride + c * 8 + 2 * hidden) : z;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e)
        *reinterpret_cast<f16x4*>(&sVt[(c * 8 + e) * VROW + kg * 4]) = f16x4{vv[0][e], vv[1][e], vv[2][e], vv[3][e]};
    }
    __syncthreads();

    // ---- S^T = K Q^T for 4 tiles of 16 keys: lane holds keys kb + 16 ct + 4 g + i of query lr
    f32x4 s[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      s[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const f16x4 kf = *reinterpret_cast<const f16x4*>(&sK[(ct * 16 + lr) * KROW + ks * 16 + g * 4]);
        s[ct] = __builtin_amdgcn_mfma_f32_16x16x16f16(kf, qf[ks], s[ct], 0, 0, 0);
      }
    }
    float alpha;
    f16x4 pf[4];
    softmax_step(s, kb + 4 * g, len, scale, m_run, l_run, alpha, pf);
    // ---- O^T = O^T alpha + V^T P^T   (A = V^T[hd 16 n + lr][key 16 ct + 4 g + j], B = P^T from the registers)
#pragma unroll
    for (int n = 0; n < NT; ++n) {
#pragma unroll
      for (int i = 0; i < 4; ++i) o[n][i] *= alpha;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const f16x4 vf = *reinterpret_cast<const f16x4*>(&sVt[(n * 16 + lr) * VROW + ct * 16 + g * 4]);
        o[n] = __builtin_amdgcn_mfma_f32_16x16x16f16(vf, pf[ct], o[n], 0, 0, 0);
      }
    }
  }
  // ---- normalise and store: this lane's query q0 + lr, head-dim columns 16 n + 4 g .. + 4 (8-byte stores)
  const int qr = q0 + lr;
  if (qr < seq) {
    const float inv = 1.0f / l_run;
    _Float16* dst = ctx + ((size_t)b * seq + qr) * hidden + h * HD;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const f16x4 v = {(_Float16)(o[n][0] * inv), (_Float16)(o[n][1] * inv), (_Float16)(o[n][2] * inv), (_Float16)(o[n][3] * inv)};
      *reinterpret_cast<f16x4*>(dst + n * 16 + 4 * g) = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Whole-sequence variant (index build): one workgroup per (batch, head) stages K and V^T of the WHOLE sequence once
// (head_dim 32, <= 256 tokens: 18 + 17 KB of LDS, four waves; head_dim 64, <= 512 tokens: 70 + 66 KB, eight waves; one
// barrier) and its waves walk the query tiles (16 queries each: tiles w, w + NW, ...) over all key blocks without
// further barriers.
// The kernel above gives each 64-query block its own workgroup, which re-stages the same K / V per query block and
// pays two barriers and an unprefetched global round trip per 64 keys -- at 256-token sequences four times the loads
// and eight barriers for 16 small MFMAs per block.  Arithmetic, masking and output are identical (same accumulation
// order per query), so the results are bit-equal to the blocked kernel's.
template <int HD, int SMAX, int NW>
__global__ __launch_bounds__(NW * 64) void attention_seq_kernel(const _Float16* __restrict__ qkv, const int* __restrict__ lens,
                                                               _Float16* __restrict__ ctx, int seq, int hidden) {
  constexpr int KS = HD / 16, NT = HD / 16;
  constexpr int KROW = HD + 4, VROW = SMAX + 4;
  constexpr int kThreads = NW * 64;
  extern __shared__ __attribute__((aligned(16))) char attn_smem[];
  _Float16* sK = reinterpret_cast<_Float16*>(attn_smem);                       // [SMAX][KROW]
  _Float16* sVt = sK + SMAX * KROW;                                            // [HD][VROW]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, g = lane >> 4;
  // (batch, head) unit of this workgroup.  Workgroups go round-robin over the 8 XCDs (id % 8), and two heads that are
  // neighbours in a token's Q / K / V row share 128-byte lines when head_dim is 32: units 2p and 2p + 1 are given to
  // the SAME XCD back to back (ids 16 j + x and 16 j + 8 + x), so the second one's K / V come out of that XCD's L2.
  const int heads = hidden / HD;
  int unit = blockIdx.x;
  if ((gridDim.x & 15) == 0) unit = (((unit >> 4) << 3) + (unit & 7)) * 2 + ((unit >> 3) & 1);
  const int b = unit / heads, h = unit % heads;
  const int len = min(max(lens[b], 1), seq);
  const size_t row_stride = (size_t)3 * hidden;
  const _Float16* base = qkv + (size_t)b * seq * row_stride + h * HD;
  const float scale = 1.44269504088896341f / sqrtf((float)HD);   // log2(e) / sqrt(hd): softmax_step works in base 2
  const int kend = (len + KB - 1) / KB * KB;       // keys staged: whole 64-key blocks up to len (rows >= seq are zero)

  constexpr int CH = HD / 8;
  for (int id = tid; id < kend * CH; id += kThreads) {
    const int key = id / CH, c = id % CH;
    f16x8 kv = {0, 0, 0, 0, 0, 0, 0, 0};
    if (key < seq) kv = *reinterpret_cast<const f16x8*>(base + (size_t)key * row_stride + c * 8 + hidden);
    *reinterpret_cast<f16x4*>(&sK[key * KROW + c * 8]) = f16x4{kv[0], kv[1], kv[2], kv[3]};
    *reinterpret_cast<f16x4*>(&sK[key * KROW + c * 8 + 4]) = f16x4{kv[4], kv[5], kv[6], kv[7]};
  }
  for (int id = tid; id < (kend / 4) * CH; id += kThreads) {
    const int kg = id / CH, c = id % CH;
    f16x8 vv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kr = kg * 4 + j;
      const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      vv[j] = (kr < seq) ? *reinterpret_cast<const f16x8*>(base + (size_t)kr * row_stride + c * 8 + 2 * hidden) : z;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
      *reinterpret_cast<f16x4*>(&sVt[(c * 8 + e) * VROW + kg * 4]) = f16x4{vv[0][e], vv[1][e], vv[2][e], vv[3][e]};
  }
  __syncthreads();

  for (int q0 = wave * 16; q0 < seq; q0 += NW * 16) {
    f16x4 qf[KS];
    const int qr = q0 + lr;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      f16x4 z = {0, 0, 0, 0};
      qf[ks] = (qr < seq) ? *reinterpret_cast<const f16x4*>(base + (size_t)qr * row_stride + ks * 16 + g * 4) : z;
    }
    f32x4 o[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) o[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -1e30f, l_run = 0.f;
    for (int kb = 0; kb < len; kb += KB) {
      f32x4 sc[4];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        sc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const f16x4 kf = *reinterpret_cast<const f16x4*>(&sK[(kb + ct * 16 + lr) * KROW + ks * 16 + g * 4]);
          sc[ct] = __builtin_amdgcn_mfma_f32_16x16x16f16(kf, qf[ks], sc[ct], 0, 0, 0);
        }
      }
      float alpha;
      f16x4 pf[4];
      softmax_step(sc, kb + 4 * g, len, scale, m_run, l_run, alpha, pf);
#pragma unroll
      for (int n = 0; n < NT; ++n) {
#pragma unroll
        for (int i = 0; i < 4; ++i) o[n][i] *= alpha;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          const f16x4 vf = *reinterpret_cast<const f16x4*>(&sVt[(n * 16 + lr) * VROW + kb + ct * 16 + g * 4]);
          o[n] = __builtin_amdgcn_mfma_f32_16x16x16f16(vf, pf[ct], o[n], 0, 0, 0);
        }
      }
    }
    if (qr < seq) {
      const float inv = 1.0f / l_run;
      _Float16* dst = ctx + ((size_t)b * seq + qr) * hidden + h * HD;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const f16x4 v = {(_Float16)(o[n][0] * inv), (_Float16)(o[n][1] * inv), (_Float16)(o[n][2] * inv), (_Float16)(o[n][3] * inv)};
        *reinterpret_cast<f16x4*>(dst + n * 16 + 4 * g) = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Round 3: the whole-sequence kernel on v_mfma_f32_16x16x32_f16 with QT query tiles per wave (index build, head_dim 32 / 64).
// The kernel above reads an 8-byte K or V^T fragment from LDS for EVERY 16x16x16 MFMA and uses it for one 16-query tile:
// 16 flops per LDS byte, i.e. the LDS port bounds it at half the matrix pipe's rate before any latency shows (measured:
// 170 TFLOP/s on bge-base's 512-token sequences, 21 % of the index build).  Here
//   * a wave owns QT = 2 query tiles (32 queries): a fragment is read once (ds_read_b128, 16 bytes per lane) and feeds
//     QT MFMAs of twice the depth -- 4 x fewer LDS instructions per flop;
//   * S^T = K Q^T contracts 32 head-dim elements per MFMA (K rows in natural order: lane = key row, 8 elements of the step);
//   * O^T += V^T P^T contracts 32 KEYS per MFMA.  P^T leaves the score accumulators as 4 consecutive keys of a 16-key tile
//     per lane; two tiles (2 s, 2 s + 1) side by side ARE a lane's 8 k-values of the 32-key step if k is numbered
//     (lane group g, tile, i) instead of key order -- any order works as long as V^T uses the same one, so V^T is STAGED
//     with its keys permuted inside every 32-key group: position = 8 (key % 16 / 4) + 4 (key / 16 % 2) + key % 4.
//     No shuffle, no LDS round trip for P;
//   * row pitches HD + 8 and SMAX + 8 halves: 16-byte aligned rows whose 16 lanes land on 16 distinct 4-bank groups.
// Same masking, scaling and online softmax (softmax_step, per query tile); accumulation ORDER differs from the kernel above
// (32-deep products), so results agree to fp32 rounding, not bit for bit.
typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));

template <int HD, int SMAX, int NW, int QT>
__global__ __launch_bounds__(NW * 64) void attention_seq32_kernel(const _Float16* __restrict__ qkv, const int* __restrict__ lens,
                                                                 _Float16* __restrict__ ctx, int seq, int hidden) {
  constexpr int KS = HD / 32, NT = HD / 16;
  constexpr int KROW = HD + 8, VROW = SMAX + 8;
  constexpr int kThr = NW * 64;
  extern __shared__ __attribute__((aligned(16))) char attn_smem[];
  _Float16* sK = reinterpret_cast<_Float16*>(attn_smem);                       // [SMAX][KROW]
  _Float16* sVt = sK + SMAX * KROW;                                            // [HD][VROW], keys permuted per 32-key group
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, g = lane >> 4;
  const int heads = hidden / HD;
  int unit = blockIdx.x;
  if ((gridDim.x & 15) == 0) unit = (((unit >> 4) << 3) + (unit & 7)) * 2 + ((unit >> 3) & 1);   // neighbouring heads on one XCD (see above)
  const int b = unit / heads, h = unit % heads;
  const int len = min(max(lens[b], 1), seq);
  const size_t row_stride = (size_t)3 * hidden;
  const _Float16* base = qkv + (size_t)b * seq * row_stride + h * HD;
  const float scale = 1.44269504088896341f / sqrtf((float)HD);
  const int kend = (len + KB - 1) / KB * KB;

  constexpr int CH = HD / 8;
  for (int id = tid; id < kend * CH; id += kThr) {
    const int key = id / CH, c = id % CH;
    f16x8 kv = {0, 0, 0, 0, 0, 0, 0, 0};
    if (key < seq) kv = *reinterpret_cast<const f16x8*>(base + (size_t)key * row_stride + c * 8 + hidden);
    *reinterpret_cast<f16x8*>(&sK[key * KROW + c * 8]) = kv;
  }
  for (int id = tid; id < (kend / 4) * CH; id += kThr) {
    const int kg = id / CH, c = id % CH;        // keys 4 kg .. 4 kg + 3, head-dim elements 8 c .. 8 c + 7
    f16x8 vv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kr = kg * 4 + j;
      const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      vv[j] = (kr < seq) ? *reinterpret_cast<const f16x8*>(base + (size_t)kr * row_stride + c * 8 + 2 * hidden) : z;
    }
    const int k0 = kg * 4, kk = k0 & 31;
    const int pos = (k0 & ~31) + ((kk & 15) >> 2) * 8 + (kk >> 4) * 4;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      *reinterpret_cast<f16x4*>(&sVt[(c * 8 + e) * VROW + pos]) = f16x4{vv[0][e], vv[1][e], vv[2][e], vv[3][e]};
  }
  __syncthreads();

  for (int q0 = wave * (16 * QT); q0 < seq; q0 += NW * 16 * QT) {
    f16x8 qf[QT][KS];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      const int qr = q0 + t * 16 + lr;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        qf[t][ks] = (qr < seq) ? *reinterpret_cast<const f16x8*>(base + (size_t)qr * row_stride + ks * 32 + g * 8) : z;
      }
    }
    f32x4 o[QT][NT];
    float m_run[QT], l_run[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      m_run[t] = -1e30f;
      l_run[t] = 0.f;
#pragma unroll
      for (int n = 0; n < NT; ++n) o[t][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int kb = 0; kb < len; kb += KB) {
      f32x4 sc[QT][4];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
#pragma unroll
        for (int t = 0; t < QT; ++t) sc[t][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const f16x8 kf = *reinterpret_cast<const f16x8*>(&sK[(kb + ct * 16 + lr) * KROW + ks * 32 + g * 8]);
#pragma unroll
          for (int t = 0; t < QT; ++t) sc[t][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[t][ks], sc[t][ct], 0, 0, 0);
        }
      }
      f16x8 pb[QT][2];      // P^T as B operands of the two 32-key steps: (tile 2 s, tile 2 s + 1) side by side
#pragma unroll
      for (int t = 0; t < QT; ++t) {
        float alpha;
        f16x4 pf[4];
        softmax_step(sc[t], kb + 4 * g, len, scale, m_run[t], l_run[t], alpha, pf);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
          pb[t][s2] = f16x8{pf[2 * s2][0], pf[2 * s2][1], pf[2 * s2][2], pf[2 * s2][3],
                            pf[2 * s2 + 1][0], pf[2 * s2 + 1][1], pf[2 * s2 + 1][2], pf[2 * s2 + 1][3]};
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int i = 0; i < 4; ++i) o[t][n][i] *= alpha;
      }
#pragma unroll
      for (int n = 0; n < NT; ++n) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const f16x8 vf = *reinterpret_cast<const f16x8*>(&sVt[(n * 16 + lr) * VROW + kb + s2 * 32 + g * 8]);
#pragma unroll
          for (int t = 0; t < QT; ++t) o[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pb[t][s2], o[t][n], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      const int qr = q0 + t * 16 + lr;
      if (qr < seq) {
        const float inv = 1.0f / l_run[t];
        _Float16* dst = ctx + ((size_t)b * seq + qr) * hidden + h * HD;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const f16x4 v = {(_Float16)(o[t][n][0] * inv), (_Float16)(o[t][n][1] * inv), (_Float16)(o[t][n][2] * inv), (_Float16)(o[t][n][3] * inv)};
          *reinterpret_cast<f16x4*>(dst + n * 16 + 4 * g) = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Round 3: query-length sequences (<= 16 tokens, head_dim 32 / 64): one WAVE per (sequence, head), no workgroup barrier.
// The blocked kernel gives a 64-query block to four waves: at 16 tokens three of them idle through the staging and the barriers
// (bge-base over eight C3 batches: 24 576 (sequence, head) units = 24 576 workgroups; 163 us per layer in the mix).  Here a wave
// loads its unit's Q and K rows as MFMA fragments straight from global memory (16-byte loads), S^T = K Q^T is HD / 32 MFMAs of
// 16x16x32, the softmax over the 16 keys is four values per lane + two cross-lane steps, V goes through a wave-private 2.5 KB
// LDS tile to become V^T fragments, O^T = V^T P^T is HD / 16 MFMAs of 16x16x16 with P^T straight from the score registers.
template <int HD>
__global__ __launch_bounds__(256) void attention_short_kernel(const _Float16* __restrict__ qkv, const int* __restrict__ lens,
                                                             _Float16* __restrict__ ctx, int seq, int hidden, int n_units) {
  constexpr int KS = HD / 32, NT = HD / 16, VROW = 20;        // V^T rows of 16 keys + 4 halves of padding (8-byte aligned rows)
  __shared__ __attribute__((aligned(16))) _Float16 sVt[4][HD * VROW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, g = lane >> 4;
  const int unit = blockIdx.x * 4 + wave;
  if (unit >= n_units) return;                                 // (no workgroup barrier below)
  const int heads = hidden / HD;
  const int b = unit / heads, h = unit % heads;
  const int len = min(max(lens[b], 1), seq);
  const size_t row_stride = (size_t)3 * hidden;
  const _Float16* base = qkv + (size_t)b * seq * row_stride + h * HD;
  const float c = 1.44269504088896341f / sqrtf((float)HD);     // log2(e) / sqrt(hd)
  const f16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
  f32x4 sc = {0.f, 0.f, 0.f, 0.f};                             // S^T: rows = keys 4 g + i, column = query lr
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const _Float16* rowp = base + (size_t)lr * row_stride + ks * 32 + g * 8;
    const f16x8 qf = (lr < seq) ? *reinterpret_cast<const f16x8*>(rowp) : z8;
    const f16x8 kf = (lr < seq) ? *reinterpret_cast<const f16x8*>(rowp + hidden) : z8;
    sc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf, sc, 0, 0, 0);
  }
  // V rows -> V^T in this wave's LDS tile: lane (key = lane >> 2, quarter = lane & 3) takes HD / 4 head-dim elements of its key
  {
    const int key = lane >> 2, qt = lane & 3;
    constexpr int PER = HD / 4;                                // 16 (head_dim 64) or 8 (32) elements per lane
#pragma unroll
    for (int u = 0; u < PER / 8; ++u) {
      const f16x8 v = (key < seq) ? *reinterpret_cast<const f16x8*>(base + (size_t)key * row_stride + 2 * hidden + qt * PER + u * 8) : z8;
#pragma unroll
      for (int e = 0; e < 8; ++e) sVt[wave][(qt * PER + u * 8 + e) * VROW + key] = v[e];
    }
  }
  // softmax over the keys of query lr (keys >= len masked; key 0 is always real, so the maximum is finite)
  float mx = -1e30f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (4 * g + i >= len) sc[i] = -1e30f;
    mx = fmaxf(mx, sc[i]);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16));
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  const float mn = mx * c;
  float rs = 0.f;
  f16x4 pf;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float p = __builtin_amdgcn_exp2f(fmaf(sc[i], c, -mn));
    rs += p;
    pf[i] = (_Float16)p;
  }
  rs += __shfl_xor(rs, 16);
  rs += __shfl_xor(rs, 32);
  __builtin_amdgcn_wave_barrier();                             // the wave's V^T stores precede its fragment reads (LDS ops of one wave stay in order)
  const float inv = 1.0f / rs;
  // (every lane multiplies: as A operand lane lr is head-dim row 16 n + lr of V^T, whatever the sequence length; only the store
  // -- column = query lr -- depends on it)
  _Float16* dst = ctx + ((size_t)b * seq + (lr < seq ? lr : 0)) * hidden + h * HD;
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const f16x4 vf = *reinterpret_cast<const f16x4*>(&sVt[wave][(n * 16 + lr) * VROW + 4 * g]);
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    o = __builtin_amdgcn_mfma_f32_16x16x16f16(vf, pf, o, 0, 0, 0);
    const f16x4 ov = {(_Float16)(o[0] * inv), (_Float16)(o[1] * inv), (_Float16)(o[2] * inv), (_Float16)(o[3] * inv)};
    if (lr < seq) *reinterpret_cast<f16x4*>(dst + n * 16 + 4 * g) = ov;
  }
}

}  // namespace

int attention_launch(const _Float16* qkv, const int* lens, _Float16* ctx, int batch, int seq, int hidden,
                     int heads, hipStream_t stream) {
  const int hd = hidden / heads;
  // whole sequence per workgroup when it is long enough to matter and short enough for LDS (CRS_ATTN_SEQ=0: off)
  static int seq_on = -1;
  if (seq_on < 0) { const char* e = getenv("CRS_ATTN_SEQ"); seq_on = (e && e[0] == '0') ? 0 : 1; }
  static int short_on = -1;   // CRS_ATTN_SHORT=0: query-length sequences on the blocked kernel (A/B runs)
  if (short_on < 0) { const char* e = getenv("CRS_ATTN_SHORT"); short_on = (e && e[0] == '0') ? 0 : 1; }
  if (short_on && seq <= 16 && (hd == 32 || hd == 64)) {
    const int units = heads * batch;
    if (hd == 32) hipLaunchKernelGGL((attention_short_kernel<32>), dim3((units + 3) / 4), dim3(256), 0, stream, qkv, lens, ctx, seq, hidden, units);
    else hipLaunchKernelGGL((attention_short_kernel<64>), dim3((units + 3) / 4), dim3(256), 0, stream, qkv, lens, ctx, seq, hidden, units);
    return (int)hipGetLastError();
  }
  if (seq_on && seq > 64) {
    dim3 g2(heads * batch);
    auto launch = [&](auto kernel, int threads, int lds) -> int {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e != hipSuccess) return (int)e;
      hipLaunchKernelGGL(kernel, g2, dim3(threads), lds, stream, qkv, lens, ctx, seq, hidden);
      return (int)hipGetLastError();
    };
    static int x32 = -1;   // CRS_ATTN_X32=0: the 16x16x16 whole-sequence kernel (A/B runs)
    if (x32 < 0) { const char* e = getenv("CRS_ATTN_X32"); x32 = (e && e[0] == '0') ? 0 : 1; }
    static int qt4 = -1;   // CRS_ATTN_QT=4: four query tiles per wave (A/B)
    if (qt4 < 0) { const char* e = getenv("CRS_ATTN_QT"); qt4 = (e && e[0] == '4') ? 1 : 0; }
    if (x32 && qt4 && hd == 32 && seq <= 256) return launch(&attention_seq32_kernel<32, 256, 4, 4>, 256, (256 * 40 + 32 * 264) * 2);
    if (x32 && qt4 && hd == 64 && seq <= 512 && seq > 256) return launch(&attention_seq32_kernel<64, 512, 8, 4>, 512, (512 * 72 + 64 * 520) * 2);
    if (x32 && hd == 32 && seq <= 256) return launch(&attention_seq32_kernel<32, 256, 4, 2>, 256, (256 * 40 + 32 * 264) * 2);
    if (x32 && hd == 64 && seq <= 512 && seq > 256) return launch(&attention_seq32_kernel<64, 512, 8, 2>, 512, (512 * 72 + 64 * 520) * 2);
    if (x32 && hd == 64 && seq <= 256) return launch(&attention_seq32_kernel<64, 256, 4, 2>, 256, (256 * 72 + 64 * 264) * 2);
    if (hd == 32 && seq <= 256) return launch(&attention_seq_kernel<32, 256, 4>, 256, (256 * 36 + 32 * 260) * 2);
    if (hd == 16 && seq <= 256) return launch(&attention_seq_kernel<16, 256, 4>, 256, (256 * 20 + 16 * 260) * 2);
    if (hd == 64 && seq <= 512 && seq > 256) return launch(&attention_seq_kernel<64, 512, 8>, 512, (512 * 68 + 64 * 516) * 2);
    if (hd == 64 && seq <= 256) return launch(&attention_seq_kernel<64, 256, 4>, 256, (256 * 68 + 64 * 260) * 2);
  }
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
