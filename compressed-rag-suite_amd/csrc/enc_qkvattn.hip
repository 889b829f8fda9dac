// enc_qkvattn.hip -- QKV projection + self-attention of SHORT sequences in one kernel (query batches).
//
//   ctx[t, h*hd : (h+1)*hd] = softmax(Q_h K_h^T / sqrt(hd) + padding mask) V_h,   [Q|K|V] = x W_qkv^T + b
//
// On the retrieve path a step is bound by the encoder's launch chain, not by arithmetic: a dependent
// kernel costs ~6 us on this part however little it does (one query of 16 tokens: 44 launches, 266 us;
// 64 queries: 292 us), and every launch also costs the command processor ~1.3 us that the other in-flight
// batches cannot use.  The attention kernel (enc_attn.hip) at 16 tokens per sequence is such a launch.
// Attention only mixes tokens of ONE sequence and ONE head, so a workgroup that computes the Q, K and V
// columns of one head for 64 tokens = 64/S whole sequences has everything it needs:
//   * grid = (token blocks of 64) x heads; 8 waves;
//   * phase 1, the panel GEMM of enc_gemm.hip restricted to that head: the x16 panel [64, H] and the
//     3*hd rows of W_qkv that belong to the head (Q, K and V slices) are fetched in one shot by LDS-DMA
//     (swizzle on the source side), one wait, then 32x32x16 MFMAs from LDS: [64, 3*hd] = 2 x (3*hd/32) tiles;
//     + bias, cast to fp16, written to LDS as Q, K (row-major) and V TRANSPOSED;
//   * phase 2, one wave per 16 query rows: S = Q K^T over the sequence's S <= 64 keys
//     (v_mfma_f32_16x16x16_f16), fp32 softmax of the full row in registers (no online rescaling needed at
//     this length), P through a wave-private LDS tile into the A-operand layout, O = P V, fp16 out.
// Requirements (the launcher reports "unsupported" otherwise and enc_capi.hip keeps the two-kernel path):
// H <= 384 (the one-shot panels must fit LDS), head_dim 32 or 64, S in {16, 32, 64}.

#include "enc.h"

namespace crs {
namespace {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int kQaThreads = 512;
constexpr int QM = 64;   // tokens per workgroup

template <int HD>
struct QaCfg {
  static constexpr int NC = 3 * HD;            // output columns of the head: Q | K | V
  static constexpr int QROW = HD + 8;          // halves per row of the Q / K tiles (16-byte aligned, bank-spread)
  static constexpr int VROW = QM + 8;          // halves per row of V^T: [HD][QM]
  static constexpr int PROW = 64 + 8;          // halves per row of a wave's P tile: [16][<= 64 keys]
  static constexpr int kQkvBytes = (2 * QM * QROW + HD * VROW) * 2;
  static constexpr int kPBytes = 4 * 16 * PROW * 2;
};

template <int HD>
__global__ __launch_bounds__(kQaThreads, 1) void qkv_attn_kernel(const _Float16* __restrict__ x16,
                                                                const _Float16* __restrict__ W,     // [3H, H]
                                                                const float* __restrict__ bias,     // [3H]
                                                                const int* __restrict__ lens, _Float16* __restrict__ ctx,
                                                                int T, int seq, int H) {
  using C = QaCfg<HD>;
  extern __shared__ __attribute__((aligned(16))) char qsm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * QM, head = blockIdx.y;
  const int cpr = H >> 3;                       // 16-byte chunks per panel row (H multiple of 128 -> multiple of 16)
  char* sa = qsm;                               // x16 panel [QM][H]
  char* sw = qsm + QM * cpr * 16;               // W slice  [NC][H]: rows 0..HD-1 = Q rows of the head, then K, then V
  char* sqkv = sw + C::NC * cpr * 16;           // Q [QM][QROW], K [QM][QROW], V^T [HD][VROW]
  char* sp = sqkv + C::kQkvBytes;               // P tiles, one per attention wave

  // ---- phase 1a: one-shot LDS-DMA of both panels (rows past T are clamped; their results are never stored)
  {
    const int total = (QM + C::NC) * cpr;       // 16-byte pieces; a multiple of 64 (cpr is a multiple of 16)
    for (int base = wave * 64; base < total; base += kQaThreads) {
      const int p = base + lane;
      const int row = p / cpr, cp = p - row * cpr;
      const int c = (cp & ~15) | ((cp ^ row) & 15);
      const _Float16* g;
      if (row < QM) {
        g = x16 + (size_t)min(m0 + row, T - 1) * H + c * 8;
      } else {
        const int wr = row - QM;                // 0 .. 3*HD-1: part = wr / HD (Q, K, V), row of the head = wr % HD
        g = W + (size_t)((wr / HD) * H + head * HD + (wr % HD)) * H + c * 8;
      }
      __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(qsm + base * 16), 16, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- phase 1b: [QM, NC] = x W^T: 2 row blocks x NC/32 column blocks of 32x32, one per wave (6 of 8 waves at HD = 32)
  constexpr int kColBlocks = C::NC / 32;
  for (int tile = wave; tile < 2 * kColBlocks; tile += 8) {
    const int rb = tile % 2, cb = tile / 2;
    const int fr = lane & 31, fh = lane >> 5;
    const int arow = rb * 32 + fr, wrow = cb * 32 + fr;
    const char* pa = sa + arow * (cpr * 16);
    const char* pw = sw + wrow * (cpr * 16);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int ksteps = H >> 4;
#pragma unroll 4
    for (int ks = 0; ks < ksteps; ++ks) {
      const int c = ks * 2 + fh;
      const f16x8 af = *reinterpret_cast<const f16x8*>(pa + (((c & ~15) | ((c ^ arow) & 15)) << 4));
      const f16x8 bf = *reinterpret_cast<const f16x8*>(pw + (((c & ~15) | ((c ^ wrow) & 15)) << 4));
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc, 0, 0, 0);
    }
    // lane holds column n = cb*32 + fr of the head's [Q|K|V], rows rb*32 + (r&3) + 8(r>>2) + 4 fh
    const int n = cb * 32 + fr;
    const int part = n / HD, col = n % HD;      // 0 Q, 1 K, 2 V
    const float bv = bias[part * H + head * HD + col];
    _Float16* q_or_k = reinterpret_cast<_Float16*>(sqkv) + part * (QM * C::QROW);
    _Float16* vt = reinterpret_cast<_Float16*>(sqkv) + 2 * QM * C::QROW;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
      const _Float16 v = (_Float16)(acc[r] + bv);
      if (part < 2) q_or_k[row * C::QROW + col] = v;
      else vt[col * C::VROW + row] = v;
    }
  }
  __syncthreads();
  if (wave >= 4) return;                        // attention: one wave per 16 query rows

  // ---- phase 2: wave w owns tokens 16 w .. 16 w + 15 of the block; their sequence starts at token s0
  const int lr = lane & 15, g = lane >> 4;
  const int q0 = wave * 16;
  const int s0 = (q0 / seq) * seq;              // first token (in the block) of this wave's sequence
  const int b = (m0 + s0) / seq;                // batch row
  if (m0 + s0 >= T) return;                     // wave-uniform: block tail past the last sequence
  const int len = min(max(lens[b], 1), seq);
  const _Float16* sQ = reinterpret_cast<const _Float16*>(sqkv);
  const _Float16* sK = sQ + QM * C::QROW;
  const _Float16* sVt = sK + QM * C::QROW;
  _Float16* myP = reinterpret_cast<_Float16*>(sp) + wave * 16 * C::PROW;
  const float scale = 1.0f / sqrtf((float)HD);
  constexpr int KS = HD / 16;
  f16x4 qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const f16x4*>(&sQ[(q0 + lr) * C::QROW + ks * 16 + g * 4]);
  const int nkt = seq >> 4;                     // key tiles of 16: 1, 2 or 4
  f32x4 s[4];
  float mx[4] = {-1e30f, -1e30f, -1e30f, -1e30f};
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    s[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (ct < nkt) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const f16x4 kf = *reinterpret_cast<const f16x4*>(&sK[(s0 + ct * 16 + lr) * C::QROW + ks * 16 + g * 4]);
        s[ct] = __builtin_amdgcn_mfma_f32_16x16x16f16(qf[ks], kf, s[ct], 0, 0, 0);
      }
      const bool valid = (ct * 16 + lr) < len;  // lane holds rows 4g+i, key column ct*16 + lr
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        s[ct][i] = valid ? s[ct][i] * scale : -1e30f;
        mx[i] = fmaxf(mx[i], s[ct][i]);
      }
    }
  }
  float rs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) mx[i] = fmaxf(mx[i], __shfl_xor(mx[i], off));
  }
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    if (ct < nkt) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float p = __expf(s[ct][i] - mx[i]);   // masked keys: exp(-1e30 - m) = 0
        rs[i] += p;
        myP[(4 * g + i) * C::PROW + ct * 16 + lr] = (_Float16)p;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) rs[i] += __shfl_xor(rs[i], off);
  }
  __builtin_amdgcn_wave_barrier();
  // O = P V: A = P[row lr][key 4g+j+16 kt], B = V^T[col n*16+lr][key s0 + 4g+j+16 kt]
  constexpr int NT = HD / 16;
  f32x4 o[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) o[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    if (kt < nkt) {
      const f16x4 pf = *reinterpret_cast<const f16x4*>(&myP[lr * C::PROW + kt * 16 + g * 4]);
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const f16x4 vf = *reinterpret_cast<const f16x4*>(&sVt[(n * 16 + lr) * C::VROW + s0 + kt * 16 + g * 4]);
        o[n] = __builtin_amdgcn_mfma_f32_16x16x16f16(pf, vf, o[n], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int tok = m0 + q0 + 4 * g + i;
    if (tok >= T) continue;
    const float inv = 1.0f / rs[i];
    _Float16* dst = ctx + (size_t)tok * H + head * HD;
#pragma unroll
    for (int n = 0; n < NT; ++n) dst[n * 16 + lr] = (_Float16)(o[n][i] * inv);
  }
}

template <int HD>
int launch_qa(const _Float16* x16, const _Float16* w, const float* bias, const int* lens, _Float16* ctx, int T, int seq,
              int H, int heads, hipStream_t stream) {
  using C = QaCfg<HD>;
  const int lds = (QM + C::NC) * H * 2 + C::kQkvBytes + C::kPBytes;
  if (lds > 160 * 1024) return -1;
  static int attr_lds = 0;
  auto kernel = &qkv_attn_kernel<HD>;
  if (lds > attr_lds) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    attr_lds = lds;
  }
  hipLaunchKernelGGL(kernel, dim3((T + QM - 1) / QM, heads), dim3(kQaThreads), lds, stream, x16, w, bias, lens, ctx, T, seq, H);
  return (int)hipGetLastError();
}

}  // namespace

bool qkv_attn_supported(int hidden, int heads, int seq) {
  if (hidden > 384 || hidden % 128) return false;
  const int hd = hidden / heads;
  if (hd != 32 && hd != 64) return false;
  const int lds = (QM + 3 * hd) * hidden * 2 + (hd == 32 ? QaCfg<32>::kQkvBytes + QaCfg<32>::kPBytes : QaCfg<64>::kQkvBytes + QaCfg<64>::kPBytes);
  if (lds > 160 * 1024) return false;
  return seq == 16 || seq == 32 || seq == 64;
}

int qkv_attn_launch(const _Float16* x16, const _Float16* w_qkv, const float* b_qkv, const int* lens, _Float16* ctx, int batch,
                    int seq, int hidden, int heads, hipStream_t stream) {
  if (!qkv_attn_supported(hidden, heads, seq)) return -1;
  const int T = batch * seq;
  if (hidden / heads == 32) return launch_qa<32>(x16, w_qkv, b_qkv, lens, ctx, T, seq, hidden, heads, stream);
  return launch_qa<64>(x16, w_qkv, b_qkv, lens, ctx, T, seq, hidden, heads, stream);
}

}  // namespace crs
