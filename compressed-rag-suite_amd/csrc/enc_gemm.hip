// enc_gemm.hip -- K2/K4/K5/K6: fp16 MFMA GEMM with fused epilogues for the encoder (gfx950).
//
//   C[M, N] = epilogue( A[M, K] x W[N, K]^T + bias[N] )         (torch.nn.Linear layout)
//
// 128 x 128 output tile per 256-thread workgroup (4 wave64 as 2 x 2, 64 x 64 each, as 2 x 2
// v_mfma_f32_32x32x16_f16 tiles), K stepped 64 at a time through a double-buffered LDS stage.
// Both operands are K-contiguous, so a tile row is one 128-byte line: global loads are
// 16 bytes/lane, 8 lanes per row, and the LDS image is XOR-swizzled by row
// (chunk ^= (row >> 1) & 7) so every ds_read_b128 fragment read is bank-conflict free.
// The next K-tile's global loads are issued before the current tile's MFMAs and written to the
// other LDS buffer after them (one barrier per K-tile).
// Epilogues (fp32 accumulators -> ...):
//   0  + bias                      -> fp16      (QKV projection)
//   1  + bias, erf-GELU            -> fp16      (FFN up)
//   2  + bias + residual (fp32)    -> fp32      (attention out / FFN down; LayerNorm follows)
// MFMA-bound for index-build batches (thousands of tokens); launch-bound for a single short query.

#include "enc.h"

namespace crs {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int kThreads = 256;
constexpr int kStageBytes = (BM + BN) * BK * 2;  // 32 KiB

__device__ __forceinline__ int lds_off(int row, int chunk) {  // 128-byte rows, 8 x 16-byte chunks
  return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

template <int MODE>
__global__ __launch_bounds__(kThreads, 2) void gemm_f16_kernel(const _Float16* __restrict__ A,
                                                              const _Float16* __restrict__ W,
                                                              const float* __restrict__ bias,
                                                              const float* __restrict__ residual,
                                                              void* __restrict__ out, int M, int N, int K) {
  __shared__ __attribute__((aligned(16))) char smem[2 * kStageBytes];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

  // staging: 4 A chunks + 4 W chunks per thread and K-tile
  int g_row[4], g_chunk[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int id = j * kThreads + tid;
    g_row[j] = id >> 3;
    g_chunk[j] = id & 7;
  }
  u32x4 ra[4], rw[4];
  auto load_tile = [&](int k0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ar = m0 + g_row[j], wr = n0 + g_row[j];
      const u32x4 z = {0u, 0u, 0u, 0u};
      ra[j] = (ar < M) ? *reinterpret_cast<const u32x4*>(A + (size_t)ar * K + k0 + g_chunk[j] * 8) : z;
      rw[j] = (wr < N) ? *reinterpret_cast<const u32x4*>(W + (size_t)wr * K + k0 + g_chunk[j] * 8) : z;
    }
  };
  auto park_tile = [&](char* st) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      *reinterpret_cast<u32x4*>(st + lds_off(g_row[j], g_chunk[j])) = ra[j];
      *reinterpret_cast<u32x4*>(st + BM * 128 + lds_off(g_row[j], g_chunk[j])) = rw[j];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int fr = lane & 31, fh = lane >> 5;
  load_tile(0);
  park_tile(smem);
  __syncthreads();
  int cur = 0;
  for (int k0 = 0; k0 < K; k0 += BK) {
    const bool more = (k0 + BK) < K;
    if (more) load_tile(k0 + BK);
    const char* sa = smem + cur * kStageBytes;
    const char* sw = sa + BM * 128;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      f16x8 af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[i] = *reinterpret_cast<const f16x8*>(sa + lds_off(wm * 64 + i * 32 + fr, ks * 2 + fh));
        bf[i] = *reinterpret_cast<const f16x8*>(sw + lds_off(wn * 64 + i * 32 + fr, ks * 2 + fh));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (more) park_tile(smem + (cur ^ 1) * kStageBytes);
    __syncthreads();
    cur ^= 1;
  }

  // epilogue: lane holds column (lane & 31), rows (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wn * 64 + j * 32 + fr;
    if (col >= N) continue;
    const float b = bias ? bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        if (row >= M) continue;
        float v = acc[i][j][r] + b;
        const size_t at = (size_t)row * N + col;
        if (MODE == 1) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
        if (MODE == 2) {
          v += residual[at];
          reinterpret_cast<float*>(out)[at] = v;
        } else {
          reinterpret_cast<_Float16*>(out)[at] = (_Float16)v;
        }
      }
    }
  }
}

}  // namespace

int gemm_f16_launch(const _Float16* a, const _Float16* w, const float* bias, const float* residual,
                    void* out, int m, int n, int k, int mode, hipStream_t stream) {
  dim3 grid((n + BN - 1) / BN, (m + BM - 1) / BM);
  switch (mode) {
    case 0: hipLaunchKernelGGL((gemm_f16_kernel<0>), grid, dim3(kThreads), 0, stream, a, w, bias, residual, out, m, n, k); break;
    case 1: hipLaunchKernelGGL((gemm_f16_kernel<1>), grid, dim3(kThreads), 0, stream, a, w, bias, residual, out, m, n, k); break;
    case 2: hipLaunchKernelGGL((gemm_f16_kernel<2>), grid, dim3(kThreads), 0, stream, a, w, bias, residual, out, m, n, k); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}

}  // namespace crs
