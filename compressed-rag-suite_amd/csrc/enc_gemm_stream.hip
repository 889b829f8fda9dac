// enc_gemm_stream.hip -- row-streaming MFMA GEMM for short contractions (K <= 512), the index-build
// side of the encoder (tens of thousands of token rows against a small weight matrix).
//
//   C[M, N] = epilogue( A[M, K] x W[N, K]^T + bias[N] )
//
// The tiled kernel in enc_gemm.hip re-fetches a 128 x 64 slice of BOTH operands every 64 steps of K
// and, at K = 384, spends as long in its prologue/epilogue as in its six K-steps (254-284 TFLOP/s
// on the MiniLM shapes).  Here the roles are those of the scan kernel (scan.hip):
//   * W is the "query" side: each wave keeps the B fragments of ITS 32 output columns for the whole
//     depth K in VGPRs (96 registers at K = 384) for the life of the workgroup -- W never touches LDS;
//   * A is the "slab": a workgroup walks a stream of 32-row tiles (whole rows, one contiguous
//     32*K*2-byte block of HBM per tile), staged global -> VGPR -> LDS with the same 16-byte XOR
//     swizzle, next tile's loads issued (inline asm) before the current tile's MFMAs;
//   * per tile and wave: K/16 v_mfma_f32_32x32x16_f16 on one accumulator, then the epilogue
//     (bias / erf-GELU / +residual) straight from the accumulator registers.
// Grid: 1-D, ncb * nstreams workgroups; workgroup b serves column block (b >> 3) % ncb of row stream
// ((b >> 3) / ncb) * 8 + (b & 7): the column blocks of one stream sit on ids b, b + 8, ... = ONE XCD, so
// an A tile is fetched from HBM / the Infinity Cache once and re-read from that XCD's L2 (with column block
// fastest the nine blocks of the QKV projection landed on eight XCDs and the 450 MB of re-reads per launch
// ran at the Infinity Cache's ~5 TB/s).  (Was: blockIdx.x = column block, blockIdx.y = row stream, so the column blocks that
// share an A tile run together and A is fetched from HBM once.  One barrier per tile.

#include "enc.h"
#include "lds_dma.h"
#include "enc_gelu.h"

namespace crs {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int TR = 32;     // A rows per tile
constexpr int WN = 128;    // output columns per workgroup (4 waves x 32)
constexpr int kEpiStride = 40;   // halves per row of the wave-private epilogue tile (32 + 8: 80-byte rows, 16-byte aligned)

template <int K, int MODE>
__global__ __launch_bounds__(kThreads, 2) void gemm_stream_kernel(const _Float16* __restrict__ A,
                                                                 const _Float16* __restrict__ W,
                                                                 const float* __restrict__ bias,
                                                                 const float* __restrict__ residual,
                                                                 void* __restrict__ out, int M, int N, int ncb, int nstreams) {
  constexpr int kCpr = K / 8;                     // 16-byte chunks per row
  constexpr int kTileBytes = TR * K * 2;
  constexpr int kLoads = kTileBytes / (kThreads * 16);
  constexpr int kKsteps = K / 16;
  static_assert(K % 128 == 0 && kTileBytes % (kThreads * 16) == 0, "K must be a multiple of 128");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, fh = lane >> 5;
  const int n_tiles = (M + TR - 1) / TR;
  const int bid = (int)blockIdx.x;
  const int cblk = (nstreams & 7) ? bid % ncb : (bid >> 3) % ncb;
  const int strm = (nstreams & 7) ? bid / ncb : ((bid >> 3) / ncb) * 8 + (bid & 7);
  const int col = cblk * WN + wave * 32 + fr;      // this lane's output column
  const bool col_ok = col < N;

  // staging geometry (chunk P = j*256 + tid of the tile -> swizzled LDS offset), as in scan.hip
  // tile transfer: global memory -> LDS directly (global_load_lds_dwordx4, lds_dma.h): LDS position P = j * T + tid
  // receives chunk swz(P) of the tile (source-side swizzle), no staging registers, no LDS stores.  (Round 1 parked the
  // tile in VGPRs through inline-asm loads with a separate counted wait -- a register the compiler may copy or spill
  // before the data has landed; that bug class produced one wrong answer in the scan and is gone from the library.)
  unsigned src_off[kLoads];
#pragma unroll
  for (int j = 0; j < kLoads; ++j) {
    const int P = j * kThreads + tid;
    const int r = P / kCpr, c = P % kCpr;
    src_off[j] = (unsigned)(r * kCpr + ((c & ~15) | ((c ^ r) & 15))) * 16u;
  }
  const char* a_bytes = reinterpret_cast<const char*>(A);
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const unsigned lds_wave = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr_t)smem + (unsigned)wave * 1024u);
  auto dma_tile = [&](int tile_, int buf) {
    const int tile = __builtin_amdgcn_readfirstlane(tile_);
    if (tile >= n_tiles) return;
    // only the ragged last tile is shorter than kTileBytes: clamp every lane's offset to its last valid 16 bytes
    const long long left = ((long long)M - (long long)tile * TR) * (K * 2);
    const unsigned lim = (unsigned)((left < kTileBytes ? left : kTileBytes) - 16);
    const char* base = uniform_ptr(a_bytes + (size_t)tile * kTileBytes);
#pragma unroll
    for (int j = 0; j < kLoads; ++j) {
      const unsigned off = src_off[j] < lim ? src_off[j] : lim;
      lds_dma16(lds_wave + (unsigned)(buf * kTileBytes) + (unsigned)(j * kThreads * 16), off, base);
    }
  };

  int t = strm;
  dma_tile(t, 0);

  // W fragments of this wave's 32 columns: B[k][n] with n = lane & 31, k = 16 ks + 8 (lane >> 5) + j
  f16x8 wf[kKsteps];
  {
    const _Float16* wrow = W + (size_t)(col_ok ? col : 0) * K + fh * 8;
#pragma unroll
    for (int ks = 0; ks < kKsteps; ++ks) {
      f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      wf[ks] = col_ok ? *reinterpret_cast<const f16x8*>(wrow + ks * 16) : z;
    }
#pragma unroll
    for (int ks = 0; ks < kKsteps; ++ks) {
      f16x8 x = wf[ks];
      asm volatile("" : "+v"(x));
      wf[ks] = x;
    }
  }
  const float bcol = (bias && col_ok) ? bias[col] : 0.f;
  // A-fragment read offset: row fr, chunk 2 ks + fh, swizzled by row
  int a_off[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) a_off[m] = fr * (kCpr * 16) + ((((m * 2 + fh) ^ fr) & 15) << 4);

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tile 0 has landed
  __syncthreads();
  int cur = 0;
  for (; t < n_tiles; t += nstreams) {
    dma_tile(t + nstreams, cur ^ 1);   // the next tile flies while this one is multiplied (its buffer was read one iteration ago)
    const char* buf = smem + cur * kTileBytes;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < kKsteps; ++ks) {
      const f16x8 af = *reinterpret_cast<const f16x8*>(buf + a_off[ks & 7] + (ks >> 3) * 256);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, wf[ks], acc, 0, 0, 0);
    }
    // epilogue: lane holds column `col`, rows (r & 3) + 8 (r >> 2) + 4 fh of the tile
    if (MODE == 2) {   // fp32 + residual: straight from the registers (64-byte segments per row)
      if (col_ok) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = t * TR + (r & 3) + 8 * (r >> 2) + 4 * fh;
          if (row < M) {
            const size_t at = (size_t)row * N + col;
            reinterpret_cast<float*>(out)[at] = (acc[r] + bcol) + residual[at];
          }
        }
      }
    } else {
      // fp16 outputs go through a wave-private LDS tile so that a lane stores 16 bytes of one row: two
      // global stores per 32 x 32 tile instead of sixteen 2-byte-per-lane ones (a vector-memory
      // instruction costs the wave ~100 cycles of issue whatever its width, and sixteen of them per tile
      // took longer than the tile's 24 MFMAs)
      _Float16* my = reinterpret_cast<_Float16*>(smem + 2 * kTileBytes) + wave * (32 * kEpiStride);
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        gelu_f32x2 v = {acc[r] + bcol, acc[r + 1] + bcol};
        if (MODE == 1) v = gelu_erf2(v);
        my[((r & 3) + 8 * (r >> 2) + 4 * fh) * kEpiStride + fr] = (_Float16)v[0];
        my[(((r + 1) & 3) + 8 * (r >> 2) + 4 * fh) * kEpiStride + fr] = (_Float16)v[1];
      }
      __builtin_amdgcn_wave_barrier();
      const int c0 = cblk * WN + wave * 32;            // first column of this wave's block
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int lrow = half * 16 + (lane >> 2), lc = (lane & 3) * 8;
        const f16x8 h = *reinterpret_cast<const f16x8*>(&my[lrow * kEpiStride + lc]);
        const int row = t * TR + lrow;
        if (row < M) {
          _Float16* dst = reinterpret_cast<_Float16*>(out) + (size_t)row * N + c0 + lc;
          if (c0 + lc + 7 < N && (N & 7) == 0) {
            *reinterpret_cast<f16x8*>(dst) = h;
          } else {
            for (int e = 0; e < 8 && c0 + lc + e < N; ++e) dst[e] = h[e];
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next tile landed (and this tile's stores retired)
    __syncthreads();
    cur ^= 1;
  }
}

// ---------------------------------------------------------------------------------------------
// K = 768 / 1024 (bge-class models): 32 columns x K of W would be 192 / 256 VGPRs, so the contraction is split
// over the two waves of a SIMD exactly as in scan_wide_ks.hip: 8 waves; waves p and p + 4 own the same 32
// output columns and one half of K each (K/8 VGPRs of W fragments); wave p + 4 hands its 32 x 32 half-sums
// over through a double-buffered LDS tile before the tile barrier, wave p adds them to its own (kept across
// the barrier) and runs the epilogue one tile late -- one wave of a SIMD multiplies while the other stores.
// fp16 outputs (modes 0 / 1) only; 128 columns per workgroup, one workgroup per CU.
constexpr int kKsThreads = 512;

template <int K, int MODE>
__global__ __launch_bounds__(kKsThreads, 2) void gemm_stream_ks_kernel(const _Float16* __restrict__ A,
                                                                      const _Float16* __restrict__ W,
                                                                      const float* __restrict__ bias,
                                                                      _Float16* __restrict__ out, int M, int N, int ncb,
                                                                      int nstreams) {
  constexpr int DH = K / 2;
  constexpr int kCpr = K / 8;
  constexpr int kTileBytes = TR * K * 2;
  constexpr int kLoads = kTileBytes / (kKsThreads * 16);
  constexpr int kKsteps = DH / 16;
  static_assert(DH % 128 == 0 && kTileBytes % (kKsThreads * 16) == 0, "K/2 must keep the 256-byte swizzle groups whole");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* xbuf = reinterpret_cast<float*>(smem + 2 * kTileBytes);                       // [pair][buffer][reg][lane]
  _Float16* epi = reinterpret_cast<_Float16*>(smem + 2 * kTileBytes + 4 * 2 * 16 * 64 * 4);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pair = wave & 3, kh = wave >> 2;
  const int fr = lane & 31, fh = lane >> 5;
  const int n_tiles = (M + TR - 1) / TR;
  const int bid = (int)blockIdx.x;
  const int cblk = (nstreams & 7) ? bid % ncb : (bid >> 3) % ncb;
  const int strm = (nstreams & 7) ? bid / ncb : ((bid >> 3) / ncb) * 8 + (bid & 7);
  const int col = cblk * WN + pair * 32 + fr;
  const bool col_ok = col < N;

  // tile transfer: global memory -> LDS directly (global_load_lds_dwordx4, lds_dma.h): LDS position P = j * T + tid
  // receives chunk swz(P) of the tile (source-side swizzle), no staging registers, no LDS stores.  (Round 1 parked the
  // tile in VGPRs through inline-asm loads with a separate counted wait -- a register the compiler may copy or spill
  // before the data has landed; that bug class produced one wrong answer in the scan and is gone from the library.)
  unsigned src_off[kLoads];
#pragma unroll
  for (int j = 0; j < kLoads; ++j) {
    const int P = j * kKsThreads + tid;
    const int r = P / kCpr, c = P % kCpr;
    src_off[j] = (unsigned)(r * kCpr + ((c & ~15) | ((c ^ r) & 15))) * 16u;
  }
  const char* a_bytes = reinterpret_cast<const char*>(A);
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const unsigned lds_wave = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr_t)smem + (unsigned)wave * 1024u);
  auto dma_tile = [&](int tile_, int buf) {
    const int tile = __builtin_amdgcn_readfirstlane(tile_);
    if (tile >= n_tiles) return;
    // only the ragged last tile is shorter than kTileBytes: clamp every lane's offset to its last valid 16 bytes
    const long long left = ((long long)M - (long long)tile * TR) * (K * 2);
    const unsigned lim = (unsigned)((left < kTileBytes ? left : kTileBytes) - 16);
    const char* base = uniform_ptr(a_bytes + (size_t)tile * kTileBytes);
#pragma unroll
    for (int j = 0; j < kLoads; ++j) {
      const unsigned off = src_off[j] < lim ? src_off[j] : lim;
      lds_dma16(lds_wave + (unsigned)(buf * kTileBytes) + (unsigned)(j * kKsThreads * 16), off, base);
    }
  };

  int t = strm;
  dma_tile(t, 0);
  f16x8 wf[kKsteps];   // B[k][n]: n = lane & 31, k = kh DH + 16 ks + 8 (lane >> 5) + j
  {
    const _Float16* wrow = W + (size_t)(col_ok ? col : 0) * K + kh * DH + fh * 8;
#pragma unroll
    for (int ks = 0; ks < kKsteps; ++ks) {
      f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      wf[ks] = col_ok ? *reinterpret_cast<const f16x8*>(wrow + ks * 16) : z;
    }
#pragma unroll
    for (int ks = 0; ks < kKsteps; ++ks) {
      f16x8 x = wf[ks];
      asm volatile("" : "+v"(x));
      wf[ks] = x;
    }
  }
  const float bcol = (bias && col_ok) ? bias[col] : 0.f;
  int a_off[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) a_off[m] = fr * (kCpr * 16) + kh * (DH * 2) + ((((m * 2 + fh) ^ fr) & 15) << 4);

  auto sweep = [&](const char* buf) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < kKsteps; ++ks) {
      const f16x8 af = *reinterpret_cast<const f16x8*>(buf + a_off[ks & 7] + (ks >> 3) * 256);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, wf[ks], acc, 0, 0, 0);
    }
    return acc;
  };
  // epilogue of tile te (the ie-th of the stream): own half-sums + the partner's -> bias / GELU -> LDS tile -> 16-byte stores
  auto finish = [&](const f32x16& own, int te, int ie) {
    const float* xb = xbuf + ((pair * 2 + (ie & 1)) * 16) * 64 + lane;
    _Float16* my = epi + pair * (32 * kEpiStride);
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      gelu_f32x2 v = {(own[r] + xb[r * 64]) + bcol, (own[r + 1] + xb[(r + 1) * 64]) + bcol};
      if (MODE == 1) v = gelu_erf2(v);
      my[((r & 3) + 8 * (r >> 2) + 4 * fh) * kEpiStride + fr] = (_Float16)v[0];
      my[(((r + 1) & 3) + 8 * (r >> 2) + 4 * fh) * kEpiStride + fr] = (_Float16)v[1];
    }
    __builtin_amdgcn_wave_barrier();
    const int c0 = cblk * WN + pair * 32;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int lrow = half * 16 + (lane >> 2), lc = (lane & 3) * 8;
      const f16x8 h = *reinterpret_cast<const f16x8*>(&my[lrow * kEpiStride + lc]);
      const int row = te * TR + lrow;
      if (row < M) {
        _Float16* dst = out + (size_t)row * N + c0 + lc;
        if (c0 + lc + 7 < N && (N & 7) == 0) {
          *reinterpret_cast<f16x8*>(dst) = h;
        } else {
          for (int e = 0; e < 8 && c0 + lc + e < N; ++e) dst[e] = h[e];
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  };

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tile 0 has landed
  __syncthreads();
  f32x16 acc_prev;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc_prev[r] = 0.f;
  int cur = 0, it = 0;
  for (; t < n_tiles; t += nstreams) {
    dma_tile(t + nstreams, cur ^ 1);   // the next tile flies while this one is multiplied (its buffer was read one iteration ago)
    const char* buf = smem + cur * kTileBytes;
    if (kh == 0) {
      if (it > 0) finish(acc_prev, t - nstreams, it - 1);
      acc_prev = sweep(buf);
    } else {
      const f32x16 acc = sweep(buf);
      float* xb = xbuf + ((pair * 2 + (it & 1)) * 16) * 64 + lane;
#pragma unroll
      for (int r = 0; r < 16; ++r) xb[r * 64] = acc[r];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next tile landed (and this tile's stores retired)
    __syncthreads();
    cur ^= 1;
    ++it;
  }
  if (kh == 0 && it > 0) finish(acc_prev, t - nstreams, it - 1);
}

template <int K, int MODE>
int launch_ks(const _Float16* a, const _Float16* w, const float* bias, void* out, int m, int n, int cus, hipStream_t stream) {
  constexpr int lds = 2 * TR * K * 2 + 4 * 2 * 16 * 64 * 4 + 4 * 32 * kEpiStride * 2;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_stream_ks_kernel<K, MODE>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  const int colblocks = (n + WN - 1) / WN;
  const int n_tiles = (m + TR - 1) / TR;
  int streams = cus / colblocks;            // one workgroup per CU
  if (streams < 1) streams = 1;
  if (streams > n_tiles) streams = n_tiles;
  if (streams >= 8) streams &= ~7;
  hipLaunchKernelGGL((gemm_stream_ks_kernel<K, MODE>), dim3(colblocks * streams), dim3(kKsThreads), lds, stream, a, w, bias,
                     reinterpret_cast<_Float16*>(out), m, n, colblocks, streams);
  return (int)hipGetLastError();
}

template <int K, int MODE>
int launch_k(const _Float16* a, const _Float16* w, const float* bias, const float* residual, void* out, int m, int n,
             int cus, hipStream_t stream) {
  constexpr int lds = 2 * TR * K * 2 + 4 * 32 * kEpiStride * 2;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_stream_kernel<K, MODE>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  const int colblocks = (n + WN - 1) / WN;
  const int n_tiles = (m + TR - 1) / TR;
  int streams = (cus * 2) / colblocks;      // 2 workgroups / CU resident; column blocks of a stream run together
  if (streams < 1) streams = 1;
  if (streams > n_tiles) streams = n_tiles;
  if (streams >= 8) streams &= ~7;          // whole rounds over the 8 XCDs
  dim3 grid(colblocks * streams);
  hipLaunchKernelGGL((gemm_stream_kernel<K, MODE>), grid, dim3(kThreads), lds, stream, a, w, bias, residual, out, m, n,
                     colblocks, streams);
  return (int)hipGetLastError();
}

template <int K>
int launch_mode(const _Float16* a, const _Float16* w, const float* bias, const float* residual, void* out, int m,
                int n, int mode, int cus, hipStream_t stream) {
  switch (mode) {
    case 0: return launch_k<K, 0>(a, w, bias, residual, out, m, n, cus, stream);
    case 1: return launch_k<K, 1>(a, w, bias, residual, out, m, n, cus, stream);
    case 2: return launch_k<K, 2>(a, w, bias, residual, out, m, n, cus, stream);
    default: return -1;
  }
}

}  // namespace

bool gemm_stream_supported(int k) { return k == 128 || k == 256 || k == 384 || k == 512 || k == 768; }   // 768: fp16 outputs only

int gemm_stream_launch(const _Float16* a, const _Float16* w, const float* bias, const float* residual, void* out,
                       int m, int n, int k, int mode, hipStream_t stream) {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return (int)hipErrorInvalidDevice;
    cus = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
  }
  switch (k) {
    case 128: return launch_mode<128>(a, w, bias, residual, out, m, n, mode, cus, stream);
    case 256: return launch_mode<256>(a, w, bias, residual, out, m, n, mode, cus, stream);
    case 384: return launch_mode<384>(a, w, bias, residual, out, m, n, mode, cus, stream);
    case 512: return launch_mode<512>(a, w, bias, residual, out, m, n, mode, cus, stream);
    case 768:
      if (mode == 0) return launch_ks<768, 0>(a, w, bias, out, m, n, cus, stream);
      if (mode == 1) return launch_ks<768, 1>(a, w, bias, out, m, n, cus, stream);
      return -1;
    default: return -1;
  }
}

}  // namespace crs
