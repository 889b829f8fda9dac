// enc_rowln.hip -- GEMM + bias + residual + LayerNorm in ONE kernel, for the launch-bound query-batch
// side of the encoder (a few hundred to a few thousand token rows).
//
//   x = LayerNorm( A[M, K] W[H, K]^T + bias + residual )      -> x32 (fp32 residual stream), x16 (next GEMM's input)
//
// replaces the pair "panel GEMM (split-K fp32 partials) -> layernorm_kernel" after the attention output
// projection (K = H) and after the FFN down projection (K = F): at 1024 tokens each of those kernels
// costs 5.5-9 us, almost all of it launch ramp and dependent memory round trips, and the encoder's
// 44-launch chain -- not any kernel's arithmetic -- bounds the retrieve step.
//
// A LayerNorm needs whole rows, so a workgroup owns TM = 64 (H = 384) or 32 (H = 768) FULL output rows:
//   * 8 waves; wave w accumulates a 32-row x 96-column block (three 32x32 accumulators,
//     v_mfma_f32_32x32x16_f16) of the TM x H tile over the whole contraction;
//   * K is walked in chunks of KC = 64 (H = 384) or 32 (H = 768) halves: the A chunk [TM, KC] and the W
//     chunk [H, KC] (<= 57 KB together) are fetched by LDS-DMA (global_load_lds_dwordx4, nothing staged in
//     VGPRs) into a double buffer, one barrier per chunk, the next chunk's DMA issued before the current
//     chunk's MFMAs.  LDS rows are KC halves (128 / 64 bytes); their 16-byte pieces are XOR-swizzled by row
//     on the SOURCE side (the DMA writes LDS linearly).  Rows 8 apart still share banks (2-way conflict
//     on a fragment read): 16 reads per chunk and wave, invisible next to the DMA issue;
//   * epilogue: the accumulators (+ nothing yet) go to an fp32 LDS tile that re-uses the operand buffers;
//     then every wave normalises rows of it exactly like enc_misc.hip's layernorm_kernel (bias and
//     residual added in fp32, two-pass statistics, eps inside the sqrt), so the global loads and stores
//     of the epilogue are row-contiguous.
// STATUS: parity-green but measured SLOWER than the two launches it replaces and therefore off by default
// (enc_capi.hip, CRS_ENC_ROWLN=1 to enable): owning whole rows means every workgroup streams all of W, and
// one CU's LDS-DMA issue rate (~50 B/clk) makes that 10 us at K = 384 and 41 us at K = 1536 on 16 CUs.
// The contraction is accumulated in one MFMA chain instead of per-384 split-K partials summed later:
// last-bit differences only (tests: 1 - cos < 2e-4 against the fp32 oracle, unchanged).

#include "enc.h"

namespace crs {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int kRlThreads = 512;

__device__ __forceinline__ float wave_sum_rl(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
  return x;
}

template <int H>
struct RlCfg {
  static constexpr int TM = H <= 384 ? 64 : 32;          // rows per workgroup
  static constexpr int KC = H <= 384 ? 64 : 32;          // halves of K per chunk
  static constexpr int CPR = KC / 8;                     // 16-byte pieces per LDS row
  static constexpr int NRB = TM / 32;                    // 32-row blocks
  static constexpr int NCG = 8 / NRB;                    // column groups (one per wave and row block)
  static constexpr int CB = (H / 32) / NCG;              // 32-column blocks per wave
  static constexpr int kRows = TM + H;                   // LDS rows per stage: A rows, then W rows
  static constexpr int kStageBytes = kRows * KC * 2;
  static constexpr int kPieces = kRows * CPR;            // 16-byte pieces per stage
  static constexpr int kInstr = kPieces / 64;            // DMA wave-instructions per stage
  static constexpr int kPerWave = (kInstr + 7) / 8;
  static constexpr int kTileStride = H + 4;              // floats; epilogue tile row stride
  static constexpr int kTileBytes = TM * kTileStride * 4;
  static constexpr int kLds = (2 * kStageBytes > kTileBytes) ? 2 * kStageBytes : kTileBytes;
  static_assert(H % 128 == 0 && (H / 32) % NCG == 0, "hidden size must split into 32-column blocks over the waves");
  static_assert(kPieces % 64 == 0, "a stage must be a whole number of DMA instructions");
};

template <int H>
__global__ __launch_bounds__(kRlThreads, 1) void gemm_rowln_kernel(const _Float16* __restrict__ A,
                                                                  const _Float16* __restrict__ W,
                                                                  const float* __restrict__ bias,
                                                                  const float* residual,   // may alias x32
                                                                  const float* __restrict__ g,
                                                                  const float* __restrict__ b, float eps, int M,
                                                                  int K, float* x32, _Float16* __restrict__ x16) {
  using C = RlCfg<H>;
  extern __shared__ __attribute__((aligned(16))) char rsm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * C::TM;

  // ---- DMA geometry: instruction i (this wave's i-th) covers pieces (i * 8 + wave) * 64 + lane
  const char* src[C::kPerWave];
  int dst[C::kPerWave];
#pragma unroll
  for (int i = 0; i < C::kPerWave; ++i) {
    const int ins = i * 8 + wave;
    const int p = ins * 64 + lane;
    const int row = min(p / C::CPR, C::kRows - 1), cp = p % C::CPR;
    const int c = cp ^ (row & (C::CPR - 1));
    const _Float16* base = (row < C::TM) ? A + (size_t)min(m0 + row, M - 1) * K : W + (size_t)(row - C::TM) * K;
    src[i] = reinterpret_cast<const char*>(base + c * 8);
    dst[i] = ins * 1024;   // LDS byte offset of the instruction's 1 KiB (lane * 16 added by the hardware)
  }
  auto issue = [&](int chunk, int stage) {
    char* sb = rsm + stage * C::kStageBytes;
#pragma unroll
    for (int i = 0; i < C::kPerWave; ++i) {
      if (i * 8 + wave < C::kInstr)   // wave-uniform
        __builtin_amdgcn_global_load_lds((gptr_t)(src[i] + (size_t)chunk * (C::KC * 2)), (lptr_t)(sb + dst[i]), 16, 0, 0);
    }
  };

  // ---- MFMA geometry
  const int fr = lane & 31, fh = lane >> 5;
  const int rb = wave % C::NRB, cg = wave / C::NRB;
  const int arow = rb * 32 + fr;                          // LDS row of this lane's A fragment
  int wrow[C::CB];
#pragma unroll
  for (int j = 0; j < C::CB; ++j) wrow[j] = C::TM + (cg * C::CB + j) * 32 + fr;
  f32x16 acc[C::CB];
#pragma unroll
  for (int j = 0; j < C::CB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const int nchunks = K / C::KC;
  issue(0, 0);
  for (int ch = 0; ch < nchunks; ++ch) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's share of chunk ch has landed
    __syncthreads();                                      // everyone's has; everyone is done reading the other stage
    if (ch + 1 < nchunks) issue(ch + 1, (ch + 1) & 1);
    const char* sb = rsm + (ch & 1) * C::kStageBytes;
#pragma unroll
    for (int ks = 0; ks < C::KC / 16; ++ks) {
      const int c = 2 * ks + fh;
      const f16x8 af = *reinterpret_cast<const f16x8*>(sb + arow * (C::KC * 2) + ((c ^ (arow & (C::CPR - 1))) << 4));
#pragma unroll
      for (int j = 0; j < C::CB; ++j) {
        const f16x8 bf = *reinterpret_cast<const f16x8*>(sb + wrow[j] * (C::KC * 2) + ((c ^ (wrow[j] & (C::CPR - 1))) << 4));
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc[j], 0, 0, 0);
      }
    }
  }
  __syncthreads();   // all fragment reads done: the operand buffers become the fp32 tile

  // ---- accumulators -> LDS tile [TM][H (+4)]
  float* tile = reinterpret_cast<float*>(rsm);
#pragma unroll
  for (int j = 0; j < C::CB; ++j) {
    const int col = (cg * C::CB + j) * 32 + fr;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
      tile[row * C::kTileStride + col] = acc[j][r];
    }
  }
  __syncthreads();

  // ---- row LayerNorm (as layernorm_kernel in enc_misc.hip): wave w takes rows w, w + 8, ...
  constexpr int PL = H / 64;
  float gg[PL], bb[PL], bi[PL];
#pragma unroll
  for (int i = 0; i < PL; ++i) {
    const int c = lane + 64 * i;
    gg[i] = g[c];
    bb[i] = b[c];
    bi[i] = bias ? bias[c] : 0.f;
  }
  for (int row = wave; row < C::TM; row += 8) {
    const int gr = m0 + row;
    if (gr >= M) break;   // wave-uniform
    float v[PL];
#pragma unroll
    for (int i = 0; i < PL; ++i) {
      const int c = lane + 64 * i;
      v[i] = (tile[row * C::kTileStride + c] + bi[i]) + residual[(size_t)gr * H + c];
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < PL; ++i) s += v[i];
    const float mean = wave_sum_rl(s) / H;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < PL; ++i) {
      const float d = v[i] - mean;
      q += d * d;
    }
    const float rstd = 1.0f / sqrtf(wave_sum_rl(q) / H + eps);
#pragma unroll
    for (int i = 0; i < PL; ++i) {
      const int c = lane + 64 * i;
      const float o = (v[i] - mean) * rstd * gg[i] + bb[i];
      x32[(size_t)gr * H + c] = o;
      x16[(size_t)gr * H + c] = (_Float16)o;
    }
  }
}

template <int H>
int launch_rowln(const _Float16* a, const _Float16* w, const float* bias, const float* residual, const float* g,
                 const float* b, float eps, int m, int k, float* x32, _Float16* x16, hipStream_t stream) {
  using C = RlCfg<H>;
  static bool done = false;
  auto kernel = &gemm_rowln_kernel<H>;
  if (!done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, C::kLds);
    if (e != hipSuccess) return (int)e;
    done = true;
  }
  hipLaunchKernelGGL(kernel, dim3((m + C::TM - 1) / C::TM), dim3(kRlThreads), C::kLds, stream, a, w, bias, residual, g,
                     b, eps, m, k, x32, x16);
  return (int)hipGetLastError();
}

}  // namespace

// hidden sizes the fused kernel is instantiated for; K must be a multiple of its chunk (64 / 32)
bool gemm_rowln_supported(int hidden, int k) {
  if (hidden == 384) return k % 64 == 0;
  if (hidden == 768) return k % 32 == 0;
  return false;
}

int gemm_rowln_launch(const _Float16* a, const _Float16* w, const float* bias, const float* residual, const float* g,
                      const float* b, float eps, int m, int hidden, int k, float* x32, _Float16* x16,
                      hipStream_t stream) {
  if (!gemm_rowln_supported(hidden, k)) return -1;
  if (hidden == 384) return launch_rowln<384>(a, w, bias, residual, g, b, eps, m, k, x32, x16, stream);
  return launch_rowln<768>(a, w, bias, residual, g, b, eps, m, k, x32, x16, stream);
}

}  // namespace crs
