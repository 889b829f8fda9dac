// enc_rowln.hip -- projection + bias + residual + LayerNorm in ONE pipelined kernel for the index-build side of the
// encoder (large token counts, hidden = 384): gemm_rowln2_kernel.
//
//   x = LayerNorm( A[M, K] W[H, K]^T + bias + residual )      -> x32 (fp32 residual stream), x16 (next GEMM's input)
//
// (Round 1 also kept a small-token variant of this idea, gemm_rowln_kernel, and a fused feed-forward kernel, enc_ffn.hip:
// both parity-green, both measured SLOWER than the launches they replaced -- a workgroup that owns whole rows streams all
// of W through one CU -- and both were removed in round 2 instead of staying opt-in; DESIGN.md section 3.4 keeps the numbers.)

#include "enc.h"

#include <stdlib.h>

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

// A workgroup owns 128 full rows and walks K in chunks through an LDS ring:
//   * a stage = A chunk [128, KC] + W chunk [384, KC] = 512 rows; (KC, stages) = (32, 4): 64-byte row pieces, 32 KB per
//     stage = exactly 32 LDS-DMA instructions, four per wave, three chunks in flight -- uniform, so a counted
//     s_waitcnt vmcnt(8) means "chunk ch has landed, the next two may still be in flight"; (64, 2), the default: 128-byte
//     pieces, one chunk in flight; 16-byte pieces XOR-swizzled on the source side;
//   * per chunk and wave 12 MFMAs (one 32-row block x six 32-column blocks x two k-steps, 7 fragment reads per
//     6 MFMAs) with the next stage's DMA instructions issued between them;
//   * the 128 x 384 fp32 result leaves through the ring's LDS as two 64-row tiles, each normalised row-wise
//     exactly as in layernorm2_kernel.
// Replaces gemm_f16_kernel<2> + layernorm2_kernel on the index-build side (65 536 tokens of MiniLM: out-proj
// 64 us + 47 us, FFN-down 147 us + 47 us before).  Measured: the pair of fused launches costs ~245 us per layer
// (was 305), i.e. ~4 us per 64 KB chunk and CU -- the same ~16-20 GB/s per CU an HBM sweep delivers, although W
// comes from L2; neither deeper prefetch, nor 128-byte pieces, nor rotating the K walk per workgroup moved it by
// more than a few per cent.
constexpr int kStageInstrPerWave(int kc) { return (128 + 384) * (kc / 8) / 64 / 8; }   // 4 (KC 32) or 8 (KC 64)
constexpr int kR2Rows = 128, kR2H = 384;
constexpr int kR2StageRows = kR2Rows + kR2H;                  // 512
template <int KC, int NST>
struct R2 {
  static constexpr int kCpr = KC / 8;                          // 16-byte pieces per row: 4 or 8
  static constexpr int kNi = kStageInstrPerWave(KC);
  static constexpr int kStageBytes = kR2StageRows * KC * 2;
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int KC, int NST>
__global__ __launch_bounds__(kRlThreads, 1) void gemm_rowln2_kernel(const _Float16* __restrict__ A,
                                                                   const _Float16* __restrict__ W,   // [384, K]
                                                                   const float* __restrict__ bias,
                                                                   const float* residual,            // may alias x32
                                                                   const float* __restrict__ g, const float* __restrict__ b,
                                                                   float eps, int M, int K, float* x32,
                                                                   _Float16* __restrict__ x16) {
  constexpr int H = kR2H;
  using C2 = R2<KC, NST>;
  constexpr int CPR = C2::kCpr, NI = C2::kNi, SH = (CPR == 4) ? 2 : 1;   // swizzle: piece ^ ((row >> SH) & (CPR - 1))
  extern __shared__ __attribute__((aligned(16))) char r2sm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * kR2Rows;

  // ---- DMA geometry: this wave issues instructions wave, wave + 8, wave + 16, wave + 24 of a stage
  const char* src[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int p = (i * 8 + wave) * 64 + lane;
    const int row = p / CPR, cp = p % CPR;
    const int c = cp ^ ((row >> SH) & (CPR - 1));
    const _Float16* base = (row < kR2Rows) ? A + (size_t)min(m0 + row, M - 1) * K : W + (size_t)(row - kR2Rows) * K;
    src[i] = reinterpret_cast<const char*>(base + c * 8);
  }
  // Workgroups that run side by side would all read the SAME W chunk at the same time (a few hundred L2 lines
  // hammered by every CU of the XCD); each starts its walk over K at a different chunk instead (the sum over
  // chunks is the same up to fp32 rounding order).
  const int nchunks = K / KC;
  const int rot = (int)(blockIdx.x % (unsigned)nchunks);
  auto issue_one = [&](int chunk, int i) {
    char* sb = r2sm + (chunk % NST) * C2::kStageBytes + (i * 8 + wave) * 1024;
    int kc = chunk + rot;
    kc = kc >= nchunks ? kc - nchunks : kc;
    __builtin_amdgcn_global_load_lds((gptr_t)(src[i] + (size_t)kc * (KC * 2)), (lptr_t)sb, 16, 0, 0);
  };

  // ---- MFMA geometry: wave -> row block (wave & 3), column blocks 6 (wave >> 2) .. + 5
  const int fr = lane & 31, fh = lane >> 5;
  const int rb = wave & 3, cb0 = (wave >> 2) * 6;
  const int arow = rb * 32 + fr;
  f32x16 acc[6];
#pragma unroll
  for (int j = 0; j < 6; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

#pragma unroll
  for (int pre = 0; pre < NST - 1; ++pre) {
    if (pre < nchunks) {
#pragma unroll
      for (int i = 0; i < NI; ++i) issue_one(pre, i);
    }
  }
  for (int ch = 0; ch < nchunks; ++ch) {
    const int ahead = min(NST - 2, nchunks - 1 - ch);   // chunks issued after ch that may stay in flight
    if (ahead == 2) wait_vmcnt<2 * NI>(); else if (ahead == 1) wait_vmcnt<NI>(); else wait_vmcnt<0>();
    __syncthreads();                                   // chunk ch complete for everyone; stage of chunk ch - 1 is free
    const bool more = ch + NST - 1 < nchunks;
    const char* sb = r2sm + (ch % NST) * C2::kStageBytes;
#pragma unroll
    for (int ks = 0; ks < KC / 16; ++ks) {
      const int c = 2 * ks + fh;
      const f16x8 af = *reinterpret_cast<const f16x8*>(sb + arow * (KC * 2) + ((c ^ ((arow >> SH) & (CPR - 1))) << 4));
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int wrow = kR2Rows + (cb0 + j) * 32 + fr;
        const f16x8 bf = *reinterpret_cast<const f16x8*>(sb + wrow * (KC * 2) + ((c ^ ((wrow >> SH) & (CPR - 1))) << 4));
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc[j], 0, 0, 0);
        if (more && (j == 1 || j == 4)) issue_one(ch + NST - 1, ks * 2 + (j == 4));   // NI DMA instructions spread over the MFMAs
      }
    }
  }
  __syncthreads();   // all fragment reads done: the ring becomes the fp32 tile

  // ---- two 64-row halves through LDS, then the row LayerNorm (8-byte form).  A wave normalises rows wave, wave + 8, ...
  // of a half; the residual values of ALL its eight rows are requested before the half's accumulators go to LDS, so the
  // epilogue pays one global round trip per half.  (As first written the loads sat inside the row loop: sixteen dependent
  // round trips per workgroup, ~29 us per round of workgroups with the matrix pipe idle -- "no epilogue traffic" took
  // 58 us off a 107 us out-proj launch, and staggering the workgroups did not, which is what gave it away.)
  float* tile = reinterpret_cast<float*>(r2sm);
  constexpr int ts = H + 4;
  // Row LayerNorm with HALF a wave per row: lane l of a half-wave holds 12 consecutive columns (48 bytes) of its row, so a
  // wave instruction advances two rows -- 16-byte loads and stores (five stores per row pair instead of twelve), five
  // cross-lane steps per reduction instead of six.  Same two-pass fp32 statistics, eps inside the sqrt.
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  const int hl = lane & 31, hw = lane >> 5;
  const int c0 = 12 * hl;
  f32x4 bi[3], gg[3], bb[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    bi[i] = bias ? *reinterpret_cast<const f32x4*>(bias + c0 + 4 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
    gg[i] = *reinterpret_cast<const f32x4*>(g + c0 + 4 * i);
    bb[i] = *reinterpret_cast<const f32x4*>(b + c0 + 4 * i);
  }
  auto half_sum = [](float x) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) x += __shfl_xor(x, o);
    return x;
  };
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    f32x4 re[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int gr = m0 + half * 64 + wave + 8 * (2 * j + hw);
#pragma unroll
      for (int i = 0; i < 3; ++i)
        re[j][i] = gr < M ? *reinterpret_cast<const f32x4*>(residual + (size_t)gr * H + c0 + 4 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if ((rb >> 1) == half) {
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int col = (cb0 + j) * 32 + fr;
#pragma unroll
        for (int r = 0; r < 16; ++r) tile[((rb & 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh) * ts + col] = acc[j][r];
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = wave + 8 * (2 * j + hw);
      const int gr = m0 + half * 64 + row;
      f32x4 v[3];
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(&tile[row * ts + c0 + 4 * i]);
        v[i] = (t4 + bi[i]) + re[j][i];
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
      }
      const float mean = half_sum(s) / H;
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const f32x4 d = v[i] - mean;
        q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
      }
      const float rstd = 1.0f / sqrtf(half_sum(q) / H + eps);
      if (gr < M) {                                  // per half-wave
        f32x4 o[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          o[i] = (v[i] - mean) * rstd * gg[i] + bb[i];
          *reinterpret_cast<f32x4*>(x32 + (size_t)gr * H + c0 + 4 * i) = o[i];
        }
        const h8 lo = {(_Float16)o[0][0], (_Float16)o[0][1], (_Float16)o[0][2], (_Float16)o[0][3],
                       (_Float16)o[1][0], (_Float16)o[1][1], (_Float16)o[1][2], (_Float16)o[1][3]};
        const h4 hi = {(_Float16)o[2][0], (_Float16)o[2][1], (_Float16)o[2][2], (_Float16)o[2][3]};
        *reinterpret_cast<h8*>(x16 + (size_t)gr * H + c0) = lo;
        *reinterpret_cast<h4*>(x16 + (size_t)gr * H + c0 + 8) = hi;
      }
    }
    __syncthreads();   // the tile is rewritten by the other half
  }
}

}  // namespace

bool gemm_rowln2_supported(int hidden, int k) { return hidden == kR2H && k % 64 == 0 && k >= 192; }

int gemm_rowln2_launch(const _Float16* a, const _Float16* w, const float* bias, const float* residual, const float* g,
                       const float* b, float eps, int m, int hidden, int k, float* x32, _Float16* x16, hipStream_t stream) {
  if (!gemm_rowln2_supported(hidden, k)) return -1;
  constexpr int lds = 128 * 1024;   // 4 x 32 KB or 2 x 64 KB (the 64 x 388 fp32 tile re-uses it)
  static int variant = -1;
  // default: 128-byte row pieces, two stages (3.79 ms per 65 536-token MiniLM forward against 3.86 ms for the
  // four-stage ring of 64-byte pieces); CRS_ROWLN2_VARIANT=0 selects the latter
  if (variant < 0) { const char* e = getenv("CRS_ROWLN2_VARIANT"); variant = (e && e[0] == '0') ? 0 : 1; }
  static bool done[2] = {false, false};
  auto k0 = &gemm_rowln2_kernel<32, 4>;
  auto k1 = &gemm_rowln2_kernel<64, 2>;
  const void* fn = variant ? reinterpret_cast<const void*>(k1) : reinterpret_cast<const void*>(k0);
  if (!done[variant]) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    done[variant] = true;
  }
  if (variant) hipLaunchKernelGGL(k1, dim3((m + kR2Rows - 1) / kR2Rows), dim3(kRlThreads), lds, stream, a, w, bias, residual, g, b, eps, m, k, x32, x16);
  else hipLaunchKernelGGL(k0, dim3((m + kR2Rows - 1) / kR2Rows), dim3(kRlThreads), lds, stream, a, w, bias, residual, g, b, eps, m, k, x32, x16);
  return (int)hipGetLastError();
}

}  // namespace crs
