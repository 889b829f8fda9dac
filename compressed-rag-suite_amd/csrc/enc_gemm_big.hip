// enc_gemm_big.hip -- 256-row-tile MFMA GEMM for the index-build side of the encoder (tens of thousands of token
// rows; bge-base: K = 768 / 3072).
//
//   C[M, N] = epilogue( A[M, K] x W[N, K]^T + bias[N] )        mode 0: fp16; 1: erf-GELU, fp16; 2: + residual fp32, fp32
//
// Why another tile shape.  A CU takes global memory in through ONE texture-address path: 64 B/clk, i.e. one 1 KB
// LDS-DMA wave-instruction per 16 cycles for the whole CU, and the issuing wave sits in that instruction until the
// path accepts it (measured in round 2 on the scan kernels: ~95 cycles per transfer instruction and wave with four to
// eight waves issuing at once; tools/scan_w1_probe).  The 128 x 128 x 64 kernel of enc_gemm.hip moves 32 KB per
// 2.1 MFLOP -- 64 flop per staged byte, exactly the ratio of the CU's MFMA rate (4096 flop/clk) to that path -- so its
// eight transfer instructions per wave and k-step (760 cycles of issue) outweigh the sixteen MFMAs they feed (512
// cycles): 380-400 TFLOP/s on bge-base's shapes.  Here a workgroup owns 256 x BN (BN = 256: 128 flop per staged byte;
// BN = 128: 85), eight waves as 4 (rows) x 2 (columns), a wave tile of 64 x BN/2 = 2 x (BN/64) MFMA tiles
// (v_mfma_f32_32x32x16_f16; 128 accumulator registers), 32 columns of K per stage, FOUR stages of LDS fed by LDS-DMA
// (source-side swizzle) with three in flight behind counted vmcnt waits, one barrier per k-step; per k-step and wave 4
// transfer instructions against 16 MFMAs.  (A first cut with two 64-deep stages and a vmcnt(0) per k-step had one stage
// in flight: a k-step's 1 us of MFMAs cannot cover a 1.5-2 us fetch.)  One workgroup per CU (128 KB of LDS), two waves
// per SIMD: one multiplies while the other issues.
// The epilogue leaves through wave-private LDS tiles as 16-byte row-contiguous stores (enc_gemm.hip's panel kernel).
#include "enc.h"
#include "enc_gelu.h"
#include "lds_dma.h"

#include <stdlib.h>

namespace crs {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int GM = 256, GK = 32, GS = 4;   // rows per workgroup; contraction depth per stage; LDS stages (three in flight)
constexpr int kBigThreads = 512;

// 64-byte LDS rows (GK = 32 halves), 4 x 16-byte chunks: rows 4 apart share banks, so the chunk index is XORed with
// (row >> 2) & 3 -- the 16 rows a 16-lane read group touches then cover all 64 banks
__device__ __forceinline__ int lds_off_b(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int MODE, int BN>
__global__ __launch_bounds__(kBigThreads, 2) void gemm_big_kernel(const _Float16* __restrict__ A, const _Float16* __restrict__ W,
                                                                 const float* __restrict__ bias, const float* __restrict__ residual,
                                                                 void* __restrict__ out, int M, int N, int K) {
  constexpr int NT = BN / 64;                       // 32-column MFMA tiles per wave (wave tile 64 x BN/2)
  constexpr int kStage = (GM + BN) * GK * 2;        // bytes per stage: 32 KB (BN 256)
  constexpr int kLa = GM * 4 / kBigThreads;         // 16-byte chunks of the A panel per thread and stage: 2
  constexpr int kLw = BN * 4 / kBigThreads;         // ... of the W panel: 2
  constexpr int kL = kLa + kLw;                     // transfer instructions per wave and stage: 4
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  extern __shared__ __attribute__((aligned(16))) char bsm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // column blocks of one row block are neighbours in one XCD's range: they share the A panel through that L2
  const int ncb = (N + BN - 1) / BN;
  const int wid = xcd_chunked_id((int)blockIdx.x, (int)gridDim.x);
  const int m0 = (wid / ncb) * GM, n0 = (wid % ncb) * BN;

  // LDS position P = j * 512 + tid (16-byte units) of a panel = (row P >> 2, slot P & 3) receives source chunk
  // slot ^ ((row >> 2) & 3) of that row (source-side swizzle; the transfer writes LDS linearly)
  const char* ga[kLa];
  const char* gw[kLw];
#pragma unroll
  for (int j = 0; j < kLa; ++j) {
    const int id = j * kBigThreads + tid, row = id >> 2, cp = id & 3;
    ga[j] = reinterpret_cast<const char*>(A + (size_t)min(m0 + row, M - 1) * K) + ((cp ^ ((row >> 2) & 3)) << 4);
  }
#pragma unroll
  for (int j = 0; j < kLw; ++j) {
    const int id = j * kBigThreads + tid, row = id >> 2, cp = id & 3;
    gw[j] = reinterpret_cast<const char*>(W + (size_t)min(n0 + row, N - 1) * K) + ((cp ^ ((row >> 2) & 3)) << 4);
  }
  const unsigned lds_wave = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr_t)bsm + (unsigned)wave * 1024u);
  auto stage = [&](int buf, int k0) {   // k0 in halves; asm transfers (lds_dma.h): counted by hand, not by the compiler
    const unsigned d0 = lds_wave + (unsigned)(buf * kStage);
#pragma unroll
    for (int j = 0; j < kLa; ++j) lds_dma16(d0 + (unsigned)(j * kBigThreads * 16), ga[j] + (size_t)k0 * 2);
#pragma unroll
    for (int j = 0; j < kLw; ++j) lds_dma16(d0 + (unsigned)(GM * 64 + j * kBigThreads * 16), gw[j] + (size_t)k0 * 2);
  };

  f32x16 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int fr = lane & 31, fh = lane >> 5;
  const int nk = K / GK;                            // k-steps (K is a multiple of 64)
  // prologue: stages 0 .. GS - 2 in flight
#pragma unroll
  for (int s = 0; s < GS - 1; ++s)
    if (s < nk) stage(s, s * GK);
  for (int i = 0; i < nk; ++i) {
    // stage i must have landed; the (up to) two younger stages may stay in flight
    const int younger = min(nk - 1 - i, GS - 2);
    if (younger >= 2) wait_vm<2 * kL>(); else if (younger == 1) wait_vm<kL>(); else wait_vm<0>();
    __syncthreads();                                 // every wave's share of stage i is in LDS; stage i - 1's buffer is free
    if (i + GS - 1 < nk) stage((i + GS - 1) % GS, (i + GS - 1) * GK);
    const char* sa = bsm + (i % GS) * kStage;
    const char* sw = sa + GM * 64;
#pragma unroll
    for (int ks = 0; ks < GK / 16; ++ks) {
      f16x8 af[2], bf[NT];
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) af[ii] = *reinterpret_cast<const f16x8*>(sa + lds_off_b(wm * 64 + ii * 32 + fr, ks * 2 + fh));
#pragma unroll
      for (int j = 0; j < NT; ++j) bf[j] = *reinterpret_cast<const f16x8*>(sw + lds_off_b(wn * (BN / 2) + j * 32 + fr, ks * 2 + fh));
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[ii][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ii], bf[j], acc[ii][j], 0, 0, 0);
    }
  }
  __syncthreads();   // all fragment reads done: the stage buffers become the waves' output tiles

  // ---- epilogue: one 32 x 32 accumulator tile at a time through this wave's private LDS tile, then 16-byte
  // row-contiguous loads / stores.  Lane holds column (lane & 31), rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of a tile.
  constexpr int TS = 36;                            // floats per tile row (144 bytes: 16-byte aligned, bank-spread)
  float* my = reinterpret_cast<float*>(bsm) + wave * (32 * TS);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int r0 = m0 + wm * 64 + i * 32, c0 = n0 + wn * (BN / 2) + j * 32;
#pragma unroll
      for (int r = 0; r < 16; ++r) my[((r & 3) + 8 * (r >> 2) + 4 * fh) * TS + fr] = acc[i][j][r];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int lrow = q * 8 + (lane >> 3), lc = (lane & 7) * 4;
        const int row = r0 + lrow, col = c0 + lc;
        if (row < M && col < N) {
          f32x4 v = *reinterpret_cast<const f32x4*>(&my[lrow * TS + lc]);
          const size_t at = (size_t)row * N + col;
          if (col + 3 < N) {
            if (bias) v += *reinterpret_cast<const f32x4*>(bias + col);
            if (MODE == 2) {
              v += *reinterpret_cast<const f32x4*>(residual + at);
              *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + at) = v;
            } else {
              typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
              f16x4 h;
#pragma unroll
              for (int e = 0; e < 4; e += 2) {
                gelu_f32x2 x = {v[e], v[e + 1]};
                if (MODE == 1) x = gelu_erf2(x);
                h[e] = (_Float16)x[0];
                h[e + 1] = (_Float16)x[1];
              }
              *reinterpret_cast<f16x4*>(reinterpret_cast<_Float16*>(out) + at) = h;
            }
          } else {
            for (int e = 0; e < 4 && col + e < N; ++e) {
              float x = v[e] + (bias ? bias[col + e] : 0.f);
              if (MODE == 2) reinterpret_cast<float*>(out)[at + e] = x + residual[at + e];
              else reinterpret_cast<_Float16*>(out)[at + e] = (_Float16)(MODE == 1 ? gelu_erf(x) : x);
            }
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

template <int MODE, int BN>
int launch_big(const _Float16* a, const _Float16* w, const float* bias, const float* residual, void* out, int m, int n, int k,
               hipStream_t stream) {
  constexpr int lds = GS * (GM + BN) * GK * 2;
  static bool done = false;
  auto kernel = &gemm_big_kernel<MODE, BN>;
  if (!done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    done = true;
  }
  const int grid = ((n + BN - 1) / BN) * ((m + GM - 1) / GM);
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBigThreads), lds, stream, a, w, bias, residual, out, m, n, k);
  return (int)hipGetLastError();
}

template <int BN>
int launch_big_mode(const _Float16* a, const _Float16* w, const float* bias, const float* residual, void* out, int m, int n, int k, int mode,
                    hipStream_t stream) {
  switch (mode) {
    case 0: return launch_big<0, BN>(a, w, bias, residual, out, m, n, k, stream);
    case 1: return launch_big<1, BN>(a, w, bias, residual, out, m, n, k, stream);
    case 2: return launch_big<2, BN>(a, w, bias, residual, out, m, n, k, stream);
    default: return -1;
  }
}

}  // namespace

// Shapes the 256 x 256 tiles take (measured, tools/bench_gemm.py, same box): fp16-epilogue projections with N a multiple
// of 256 and K >= 512 at index-build row counts -- bge-base QKV 32768 x 2304 x 768: 457 -> 686 TFLOP/s, FFN-up
// 32768 x 3072 x 768: 507 -> 659 (both were on the row-streaming kernel); 4096^3: 742 -> 970.  NOT taken: the fp32 +
// residual epilogue at N = 768 (three column blocks = 384 workgroups = one and a half waves of the chip: 582 against
// 667 for the 128 x 128 kernel's 1536 workgroups), and MiniLM's K = 384 shapes (six k-steps: the row-streaming kernel's
// resident weights win, 733 / 541 against 419 / 439 with 256 x 128 tiles).  CRS_GEMM_BIG=0 disables it (A/B runs).
// It is taken from 128 workgroups on (CRS_GEMM_BIG_MIN_WGS): at 4096 tokens of bge-base (C3's query batch) 144 / 192
// workgroups of 256 x 256 beat the row-streaming kernel's 144 (QKV 42.0 -> 35.9 us, FFN-up 57.4 -> 41.5); at 2048 tokens
// (72 / 96 workgroups) they lose (25.7 -> 31.5, 33.1 -> 36.6).
int gemm_big_block_n(int m, int n, int k, int mode) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("CRS_GEMM_BIG"); on = (e && e[0] == '0') ? 0 : 1; }
  if (!on || mode == 2 || k % 64 != 0 || k < 512 || n % 256 != 0) return 0;
  const long wgs = (long)(n / 256) * ((m + GM - 1) / GM);
  static long min_wgs = -1;
  if (min_wgs < 0) { const char* e = getenv("CRS_GEMM_BIG_MIN_WGS"); min_wgs = e ? atol(e) : 128; }
  return wgs >= min_wgs ? 256 : 0;
}

int gemm_big_launch(const _Float16* a, const _Float16* w, const float* bias, const float* residual, void* out, int m, int n, int k, int mode,
                    hipStream_t stream) {
  if (gemm_big_block_n(m, n, k, mode) != 256) return -1;
  return launch_big_mode<256>(a, w, bias, residual, out, m, n, k, mode, stream);
}

}  // namespace crs
