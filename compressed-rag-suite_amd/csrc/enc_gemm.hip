// enc_gemm.hip -- K2/K4/K5/K6: fp16 MFMA GEMM with fused epilogues for the encoder (gfx950).
//
//   C[M, N] = epilogue( A[M, K] x W[N, K]^T + bias[N] )         (torch.nn.Linear layout)
//
// 128 x 128 output tile per 256-thread workgroup (4 wave64 as 2 x 2, 64 x 64 each, as 2 x 2
// v_mfma_f32_32x32x16_f16 tiles), K stepped 64 at a time through a double-buffered LDS stage.
// Both operands are K-contiguous, so a tile row is one 128-byte line: global loads are
// 16 bytes/lane, 8 lanes per row, and the LDS image is XOR-swizzled by row
// (chunk ^= (row >> 1) & 7) so every ds_read_b128 fragment read is bank-conflict free.
// Staging is LDS-DMA (global_load_lds_dwordx4): the next K-tile is fetched into the other buffer
// under the current tile's MFMAs, one vmcnt(0) + barrier per K-tile; the epilogue goes through
// LDS so that global loads / stores are 16-byte coalesced rows.
// Epilogues (fp32 accumulators -> ...):
//   0  + bias                      -> fp16      (QKV projection)
//   1  + bias, erf-GELU            -> fp16      (FFN up)
//   2  + bias + residual (fp32)    -> fp32      (attention out / FFN down; LayerNorm follows)
// MFMA-bound for index-build batches (thousands of tokens); launch-bound for a single short query.

#include "enc.h"
#include "enc_gelu.h"

#include <stdlib.h>

namespace crs {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int kThreads = 256;
constexpr int kStageBytes = (BM + BN) * BK * 2;  // 32 KiB

__device__ __forceinline__ int lds_off(int row, int chunk) {  // 128-byte rows, 8 x 16-byte chunks
  return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int MODE>
__global__ __launch_bounds__(kThreads, 2) void gemm_f16_kernel(const _Float16* __restrict__ A,
                                                              const _Float16* __restrict__ W,
                                                              const float* __restrict__ bias,
                                                              const float* __restrict__ residual,
                                                              void* __restrict__ out, int M, int N, int K) {
  __shared__ __attribute__((aligned(16))) char smem[2 * kStageBytes];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // 1-D grid; the column blocks of one row block are neighbours in one XCD's range (enc.h) and share the A panel there
  const int ncb = (N + BN - 1) / BN;
  const int wid = xcd_chunked_id((int)blockIdx.x, (int)gridDim.x);
  const int m0 = (wid / ncb) * BM, n0 = (wid % ncb) * BN;

  // staging by LDS-DMA: chunk P = j*256 + tid of a [128 rows][8 chunks] panel lands at LDS offset P*16
  // (linear); the bank swizzle chunk ^= (row >> 1) & 7 is applied to the SOURCE column instead.
  // Rows past M / N are clamped to the last valid row (their products are never stored).
  const _Float16* ga[4];
  const _Float16* gw[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int id = j * kThreads + tid;
    const int row = id >> 3, cp = id & 7;
    const int c = cp ^ ((row >> 1) & 7);
    ga[j] = A + (size_t)min(m0 + row, M - 1) * K + c * 8;
    gw[j] = W + (size_t)min(n0 + row, N - 1) * K + c * 8;
  }
  auto stage = [&](int buf, int k0) {
    char* sa = smem + buf * kStageBytes + wave * 1024;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      __builtin_amdgcn_global_load_lds((gptr_t)(ga[j] + k0), (lptr_t)(sa + j * (kThreads * 16)), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(gw[j] + k0), (lptr_t)(sa + BM * 128 + j * (kThreads * 16)), 16, 0, 0);
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
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int cur = 0;
  for (int k0 = 0; k0 < K; k0 += BK) {
    if (k0 + BK < K) stage(cur ^ 1, k0 + BK);     // next tile in flight under this tile's MFMAs
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue through LDS: accumulators -> fp32 tile [128][128] (the two staging buffers, 64 KiB),
  // then each thread finishes 4 (fp32 out) or 8 (fp16 out) consecutive columns of a row with
  // 16-byte coalesced loads / stores.  Lane holds column (lane & 31), rows (r&3) + 8(r>>2) + 4(lane>>5).
  float* tile = reinterpret_cast<float*>(smem);
  // mode 2: the residual values this thread will add (16 x 16 bytes) are requested BEFORE the accumulators go through
  // LDS -- one global round trip per workgroup.  (As first written the load sat inside the store loop: load, wait, store,
  // sixteen dependent round trips per thread with the matrix pipe idle; same finding as in enc_rowln.hip.)
  constexpr int kIt2 = BM * (BN / 4) / kThreads;   // 16
  const int c4 = (tid % (BN / 4)) * 4, gc2 = n0 + c4;          // this thread's column group is the same in every pass
  const bool cols_in = gc2 + 3 < N;
  f32x4 rs[kIt2];
  f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
  if (MODE == 2) {
    if (bias && cols_in) b4 = *reinterpret_cast<const f32x4*>(bias + gc2);
#pragma unroll
    for (int it = 0; it < kIt2; ++it) {
      const int gr = m0 + (tid + it * kThreads) / (BN / 4);
      rs[it] = (gr < M && cols_in) ? *reinterpret_cast<const f32x4*>(residual + (size_t)gr * N + gc2) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        tile[row * BN + wn * 64 + j * 32 + fr] = acc[i][j][r];
      }
  __syncthreads();
  if (MODE == 2) {
    float* o = reinterpret_cast<float*>(out);
#pragma unroll
    for (int it = 0; it < kIt2; ++it) {
      const int row = (tid + it * kThreads) / (BN / 4);
      const int gr = m0 + row;
      if (gr >= M || gc2 >= N) continue;
      const f32x4 v = *reinterpret_cast<const f32x4*>(tile + row * BN + c4);
      const size_t at = (size_t)gr * N + gc2;
      if (cols_in) {
        *reinterpret_cast<f32x4*>(o + at) = v + b4 + rs[it];
      } else {
        for (int e = 0; e < 4 && gc2 + e < N; ++e) o[at + e] = v[e] + (bias ? bias[gc2 + e] : 0.f) + residual[at + e];
      }
    }
  } else {
    _Float16* o = reinterpret_cast<_Float16*>(out);
    for (int id = tid; id < BM * (BN / 8); id += kThreads) {
      const int row = id / (BN / 8), c8 = (id % (BN / 8)) * 8;
      const int gr = m0 + row, gc = n0 + c8;
      if (gr >= M || gc >= N) continue;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        gelu_f32x2 x = {tile[row * BN + c8 + e] + ((bias && gc + e < N) ? bias[gc + e] : 0.f),
                        tile[row * BN + c8 + e + 1] + ((bias && gc + e + 1 < N) ? bias[gc + e + 1] : 0.f)};
        if (MODE == 1) x = gelu_erf2(x);
        v[e] = x[0];
        v[e + 1] = x[1];
      }
      const size_t at = (size_t)gr * N + gc;
      if (gc + 7 < N) {
        f16x8 h;
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = (_Float16)v[e];
        *reinterpret_cast<f16x8*>(o + at) = h;
      } else {
        for (int e = 0; e < 8 && gc + e < N; ++e) o[at + e] = (_Float16)v[e];
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------
// Panel variant for the launch/latency-bound regime (query batches: M of a few hundred to a few
// thousand rows).  There the pipelined loop above pays one HBM/L2 round trip per 64-deep K step
// on a grid far smaller than the chip.  Here a workgroup owns a 64 x 64 output tile and ONE K
// chunk of <= 384: it issues every load of both operand panels at once as LDS-DMA
// (global_load_lds_dwordx4, nothing staged in VGPRs), waits once, and runs all its MFMAs out of
// LDS.  Long contractions are split over blockIdx.z (split-K); the fp32 partial tiles are summed
// by the LayerNorm kernel that follows anyway (enc_misc.hip), so no extra pass or atomics exist.
// LDS image: rows of kc halves, 16-byte chunks XOR-swizzled by row inside each 256-byte group;
// the DMA writes LDS linearly, so the swizzle is applied to the per-lane SOURCE address.
typedef float f32x4v __attribute__((ext_vector_type(4)));
constexpr int PN = 64, PKC = 384;

// MODE 0: +bias -> fp16; 1: +bias, GELU -> fp16; 3: raw fp32 partial tile -> out[z][M][N].
// TM = rows of A per workgroup: 64 (most workgroups, for tiny M) or 128 (halves the operand re-reads
// and the workgroup count -- one wave of workgroups on 256 CUs for the 1024-token query batch).
constexpr int kPanelThreads = 512;   // 8 waves: the LDS-DMA issue (1 KiB per wave-instruction) is the long
                                      // pole of a one-shot panel fetch, so it is spread over more waves

template <int MODE, int TM>
__global__ __launch_bounds__(kPanelThreads, 1) void gemm_panel_kernel(const _Float16* __restrict__ A,
                                                                     const _Float16* __restrict__ W,
                                                                     const float* __restrict__ bias,
                                                                     void* __restrict__ out, int M, int N, int K,
                                                                     int kc, int kin) {
  // kin > 1: the workgroup's K range (kc * kin) is walked in kin chunks through ONE staging buffer of kc columns --
  // (TM + 64) * kc * 2 bytes of LDS (48 KB at TM = 128, kc = 128), small enough to sit on a CU beside two scan
  // workgroups, so that the encoder of the next query batch runs in the matrix/issue slots the HBM-bound scan leaves
  // idle instead of waiting for a scan boundary ("encoder beside the scan")
  constexpr int kTiles = (TM / 32) * (PN / 32);   // 32 x 32 output tiles: one per wave (8 or 4)
  extern __shared__ __attribute__((aligned(16))) char psm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * PN;
  const int cpr = kc >> 3;                 // 16-byte chunks per panel row (multiple of 16)
  char* sa = psm;
  char* sw = psm + TM * cpr * 16;
  const bool computes = wave < kTiles;   // TM = 64: waves 4..7 only help fetching
  const int fr = lane & 31, fh = lane >> 5;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  for (int ci = 0; ci < kin; ++ci) {
  const int k0 = (blockIdx.z * kin + ci) * kc;
  // ---- one shot: every chunk of both panels (rows past M / N are clamped; their results are never
  // stored).  LDS position P = base + tid walks (row, chunk) incrementally: no divisions in the loop.
  {
    int row = tid / cpr, cp = tid - row * cpr;
    const int drow = kPanelThreads / cpr, dcp = kPanelThreads - drow * cpr;
    for (int base = 0; base < TM * cpr; base += kPanelThreads) {
      const int c = (cp & ~15) | ((cp ^ row) & 15);
      const _Float16* ga = A + (size_t)min(m0 + row, M - 1) * K + k0 + c * 8;
      __builtin_amdgcn_global_load_lds((gptr_t)ga, (lptr_t)(sa + (base + wave * 64) * 16), 16, 0, 0);
      if (base < PN * cpr) {
        const _Float16* gw = W + (size_t)min(n0 + row, N - 1) * K + k0 + c * 8;
        __builtin_amdgcn_global_load_lds((gptr_t)gw, (lptr_t)(sw + (base + wave * 64) * 16), 16, 0, 0);
      }
      row += drow;
      cp += dcp;
      if (cp >= cpr) { cp -= cpr; ++row; }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (computes) {
    const int arow = wm * 32 + fr, wrow = wn * 32 + fr;
    const char* pa = sa + arow * (cpr * 16);
    const char* pw = sw + wrow * (cpr * 16);
    const int ksteps = kc >> 4;
#pragma unroll 4
    for (int ks = 0; ks < ksteps; ++ks) {
      const int c = ks * 2 + fh;
      const f16x8 af = *reinterpret_cast<const f16x8*>(pa + (((c & ~15) | ((c ^ arow) & 15)) << 4));
      const f16x8 bf = *reinterpret_cast<const f16x8*>(pw + (((c & ~15) | ((c ^ wrow) & 15)) << 4));
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc, 0, 0, 0);
    }
  }
  __syncthreads();   // every fragment read done: the staging buffer may be refilled / becomes the waves' output tiles
  }
  if (!computes) return;

  // Epilogue through a wave-private LDS tile so that a lane stores 16 bytes of one row: 4 (fp32) or 2 (fp16)
  // store instructions per 32 x 32 tile instead of 16 -- a vector-memory instruction costs the wave ~100
  // cycles of issue whatever its width, and these kernels are latency chains, not bandwidth.
  const int col = n0 + wn * 32 + fr;
  const float b = (MODE != 3 && bias && col < N) ? bias[col] : 0.f;
  const int c0 = n0 + wn * 32, r0 = m0 + wm * 32;
  if (MODE == 3) {
    constexpr int TS = 36;   // floats per tile row (144 bytes: 16-byte aligned, bank-spread)
    float* my = reinterpret_cast<float*>(psm) + wave * (32 * TS);
#pragma unroll
    for (int r = 0; r < 16; ++r) my[((r & 3) + 8 * (r >> 2) + 4 * fh) * TS + fr] = acc[r];
    __builtin_amdgcn_wave_barrier();
    float* o = reinterpret_cast<float*>(out) + (size_t)blockIdx.z * M * N;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int lrow = q * 8 + (lane >> 3), lc = (lane & 7) * 4;
      const int row = r0 + lrow;
      if (row >= M) continue;
      const f32x4v v = *reinterpret_cast<const f32x4v*>(&my[lrow * TS + lc]);
      float* dst = o + (size_t)row * N + c0 + lc;
      if (c0 + lc + 3 < N && (N & 3) == 0) {
        *reinterpret_cast<f32x4v*>(dst) = v;
      } else {
        for (int e = 0; e < 4 && c0 + lc + e < N; ++e) dst[e] = v[e];
      }
    }
  } else {
    constexpr int TS = 40;   // halves per tile row (80 bytes)
    _Float16* my = reinterpret_cast<_Float16*>(psm) + wave * (32 * TS);
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      gelu_f32x2 v = {acc[r] + b, acc[r + 1] + b};
      if (MODE == 1) v = gelu_erf2(v);
      my[((r & 3) + 8 * (r >> 2) + 4 * fh) * TS + fr] = (_Float16)v[0];
      my[(((r + 1) & 3) + 8 * (r >> 2) + 4 * fh) * TS + fr] = (_Float16)v[1];
    }
    __builtin_amdgcn_wave_barrier();
    _Float16* o = reinterpret_cast<_Float16*>(out);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int lrow = q * 16 + (lane >> 2), lc = (lane & 3) * 8;
      const int row = r0 + lrow;
      if (row >= M) continue;
      const f16x8 h = *reinterpret_cast<const f16x8*>(&my[lrow * TS + lc]);
      _Float16* dst = o + (size_t)row * N + c0 + lc;
      if (c0 + lc + 7 < N && (N & 7) == 0) {
        *reinterpret_cast<f16x8*>(dst) = h;
      } else {
        for (int e = 0; e < 8 && c0 + lc + e < N; ++e) dst[e] = h[e];
      }
    }
  }
}

template <int MODE, int TM>
int launch_panel_t(const _Float16* a, const _Float16* w, const float* bias, void* out, int m, int n, int k, int kc,
                   int splitk, int kin_req, int small_lds, hipStream_t stream) {
  static bool attr_done = false;
  // small_lds (crs_encoder_desc.flags & CRS_ENC_SMALL_LDS): stage the K range in 128-column chunks (<= 48 KB of LDS: the
  // forward can then run beside a scan's resident workgroups).  CRS_PANEL_KC=128|256|384 forces a chunk size (A/B runs).
  static int kc_env = -1;
  if (kc_env < 0) { const char* e = getenv("CRS_PANEL_KC"); kc_env = e ? atoi(e) : 0; if (kc_env != 128 && kc_env != 256 && kc_env != 384) kc_env = 0; }
  const int kc_cap = kc_env ? kc_env : (small_lds ? 128 : 0);
  // A launch of more workgroups than CUs stages 128 columns at a time: 48 KB of LDS, up to three workgroups resident
  // per CU, one workgroup's transfers under another's MFMAs (bge-base at query-batch sizes: QKV 288, FFN-up 384, FFN-down
  // 768 workgroups).  Measured on the bge-base query chain (tools/enc_chain_profile.py): 64 x 16 tokens 945 -> ~800 us per
  // forward, 16 x 16: 793 -> 584, 256 x 16: 2222 -> 2000.  A single wave of workgroups keeps the one-shot fetch (MiniLM:
  // every launch <= 192 workgroups; 237 us one-shot against 244-253 in 128-column pieces).  CRS_PANEL_KC forces a size.
  const long wgs = (long)((n + PN - 1) / PN) * ((m + TM - 1) / TM) * splitk;
  int kin = kin_req;
  if (kc_cap && kc > kc_cap && kc % kc_cap == 0) { kin *= kc / kc_cap; kc = kc_cap; }
  else if (!kc_cap && wgs > 256 && kc > 128 && kc % 128 == 0) { kin *= kc / 128; kc = 128; }
  const int lds = (TM + PN) * kc * 2;
  const int ep = (MODE == 3 ? 36 * 4 : 40 * 2) * 32 * ((TM / 32) * (PN / 32));   // the epilogue's wave-private tiles re-use the buffer
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_panel_kernel<MODE, TM>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (TM + PN) * PKC * 2);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  dim3 grid((n + PN - 1) / PN, (m + TM - 1) / TM, splitk);
  hipLaunchKernelGGL((gemm_panel_kernel<MODE, TM>), grid, dim3(kPanelThreads), lds > ep ? lds : ep, stream, a, w, bias, out, m, n, k, kc, kin);
  return (int)hipGetLastError();
}

// 128-row tiles once 64-row tiles would need more than one wave of workgroups on the chip
template <int MODE>
int launch_panel(const _Float16* a, const _Float16* w, const float* bias, void* out, int m, int n, int k, int kc,
                 int splitk, int kin, int small_lds, hipStream_t stream) {
  const long wgs64 = (long)((n + PN - 1) / PN) * ((m + 63) / 64) * splitk;
  // (forcing 64- or 128-row tiles everywhere measured within 2 % either way on both models' query chains)
  if (wgs64 > 256 && m > 64) return launch_panel_t<MODE, 128>(a, w, bias, out, m, n, k, kc, splitk, kin, small_lds, stream);
  return launch_panel_t<MODE, 64>(a, w, bias, out, m, n, k, kc, splitk, kin, small_lds, stream);
}

}  // namespace

static bool stream_enabled() {   // CRS_GEMM_STREAM=0: always the tiled kernel (A/B runs)
  static int v = -1;
  if (v < 0) { const char* e = getenv("CRS_GEMM_STREAM"); v = (e && e[0] == '0') ? 0 : 1; }
  return v == 1;
}

int gemm_f16_launch(const _Float16* a, const _Float16* w, const float* bias, const float* residual,
                    void* out, int m, int n, int k, int mode, hipStream_t stream) {
  // short contraction, many rows, wide output (the index-build side's QKV and FFN-up projections): the tiled kernel
  // below spends as long in its prologue and epilogue as in its K / 64 steps; stream rows past resident W
  if (gemm8_applies(m, n, k, mode)) return gemm8_launch(a, w, bias, residual, out, m, n, k, mode, stream);
  if (gemm_big_block_n(m, n, k, mode) != 0) return gemm_big_launch(a, w, bias, residual, out, m, n, k, mode, stream);
  if (m >= 512 && n >= 512 && mode != 2 && gemm_stream_supported(k) && stream_enabled())
    return gemm_stream_launch(a, w, bias, residual, out, m, n, k, mode, stream);
  dim3 grid(((n + BN - 1) / BN) * ((m + BM - 1) / BM));
  switch (mode) {
    case 0: hipLaunchKernelGGL((gemm_f16_kernel<0>), grid, dim3(kThreads), 0, stream, a, w, bias, residual, out, m, n, k); break;
    case 1: hipLaunchKernelGGL((gemm_f16_kernel<1>), grid, dim3(kThreads), 0, stream, a, w, bias, residual, out, m, n, k); break;
    case 2: hipLaunchKernelGGL((gemm_f16_kernel<2>), grid, dim3(kThreads), 0, stream, a, w, bias, residual, out, m, n, k); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}

}  // namespace crs

namespace crs {

// Chunk length for the panel kernel: the largest multiple of 128 that is <= 384 and divides K (0 = none)
int gemm_panel_chunk(int k) {
  for (int kc = 384; kc >= 128; kc -= 128)
    if (k % kc == 0) return kc;
  return 0;
}

// How many fp32 partial slabs a mode-3 launch of m rows over contraction length k leaves (the LayerNorm kernel that
// follows sums them): k / chunk, capped -- past the cap the workgroups walk several chunks each (kernel: kin).  The
// cap falls with the row count, because the slabs are m x N x 4 bytes each and the launch no longer lacks workgroups
// (bge-base FFN-down, K = 3072, tools/enc_chain_profile.py, whole forward): 1024 tokens: 8 / 4 / 2 slabs = 806 / 792 /
// 850 us; 2048 tokens: - / 1138 / 1132; 4096 tokens: 2109 / 1945 / 1855 (and 1996 through the 128 x 128 kernel).
int gemm_panel_splits(int k, int m) {
  const int kc = gemm_panel_chunk(k);
  if (kc == 0) return 0;
  static int cap_env = -1;   // CRS_PANEL_MAX_SPLIT: A/B runs
  if (cap_env < 0) { const char* e = getenv("CRS_PANEL_MAX_SPLIT"); cap_env = e ? atoi(e) : 0; }
  const int cap = cap_env > 0 ? cap_env : (m <= 1024 ? 4 : 2);
  int s = k / kc;
  while (s > cap && (s % 2) == 0) s /= 2;
  return s;
}

// out: mode 0/1 fp16 [M,N] (any k that is a multiple of the chunk: the workgroup walks the chunks); mode 3 fp32
// [gemm_panel_splits(k, m)][M][N] partials
int gemm_panel_launch(const _Float16* a, const _Float16* w, const float* bias, void* out, int m, int n, int k,
                      int mode, int small_lds, hipStream_t stream) {
  const int kc = gemm_panel_chunk(k);
  if (kc == 0) return -1;
  const int chunks = k / kc;
  switch (mode) {
    case 0: return launch_panel<0>(a, w, bias, out, m, n, k, kc, 1, chunks, small_lds, stream);
    case 1: return launch_panel<1>(a, w, bias, out, m, n, k, kc, 1, chunks, small_lds, stream);
    case 3: {
      const int splitk = gemm_panel_splits(k, m);
      return launch_panel<3>(a, w, bias, out, m, n, k, kc, splitk, chunks / splitk, small_lds, stream);
    }
    default: return -1;
  }
}

}  // namespace crs
