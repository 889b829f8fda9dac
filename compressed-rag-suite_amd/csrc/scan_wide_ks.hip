// scan_wide_ks.hip -- the large-batch scan (scan_wide.hip) for rows of 768 elements (bge-class models,
// BASELINE config #3: 1 M x 768, 256 queries).
//
// scan_wide.hip keeps the fragments of a wave's 32 queries for the FULL depth in registers (D/4 VGPRs); at
// D = 768 that is 192 of the 256 a wave may have, so those rows ran on scan_tb.hip's 8-wave 16x16x32 form --
// 16 queries per wave, one ds_read_b128 per 16-cycle MFMA, the LDS port as busy as the matrix pipe (C3:
// 0.31 of the roofline).  Here the CONTRACTION is split across the two waves of a SIMD instead:
//   * waves p and p + 4 serve the SAME 32 queries; wave p multiplies the first half of every row
//     (k < D/2), wave p + 4 the second half -- D/8 VGPRs of fragments each, v_mfma_f32_32x32x16_f16, one
//     ds_read_b128 per 32 MFMA cycles, and each wave reads only its half of the staged tile;
//   * wave p + 4 leaves its 32 x 32 partial sums in a double-buffered LDS exchange tile before the tile
//     barrier; after the barrier wave p adds them to its own (kept in registers across the barrier) and runs
//     the tile-best selection of scan_wide.hip one tile late.  That is also the stagger scan_wide.hip
//     builds by hand: on every SIMD one wave selects while the other multiplies;
//   * 4 pairs x 32 queries = 128 queries per workgroup, one workgroup per CU, 32-row tiles (48 KB at D = 768).
// Scores are the sum of two fp32 half-sums, i.e. equal to the single-chain value up to fp32 rounding;
// scan_refine.hip re-scores the winning tiles with the single chain and admits candidates a hair below the
// k-th representative, so the final lists are those of the other kernels.

#include "scan_common.h"

#include <stdlib.h>

namespace crs {
int scan_wide_slots(int k);
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int D>
struct KsCfg {
  static constexpr int kT = 512;
  static constexpr int DH = D / 2;                       // contraction range of one wave
  static constexpr int kCpr = D / 8;
  static constexpr int kTileBytes = 32 * D * 2;
  static constexpr int kLoads = kTileBytes / (kT * 16);
  static constexpr int kKsteps = DH / 16;
  static constexpr int kXBytes = 4 * 2 * 16 * 64 * 4;    // exchange: [pair][buffer][reg][lane] fp32
  static constexpr int kLds = 2 * kTileBytes + kXBytes;
  static_assert(DH % 128 == 0, "half rows must keep the 256-byte swizzle groups whole");
  static_assert(kTileBytes % (kT * 16) == 0, "tile must split into whole 16-byte loads");
};

template <int D, int K>
__global__ __launch_bounds__(512, 2) void scan_wide_ks_kernel(const ScanArgs a) {
  using C = KsCfg<D>;
  constexpr int kT = C::kT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* tile_buf = smem;
  float* xbuf = reinterpret_cast<float*>(smem + 2 * C::kTileBytes);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pair = wave & 3, kh = wave >> 2;              // kh = 0: first half of k + selection; 1: second half
  const int nwg = CRS_NSTREAMS;
  const int qblock = CRS_QBLOCK, stream = CRS_STREAM;
  const bool wave_active = (qblock * 128 + pair * 32) < a.nq;   // wave-uniform

  // tile transfer as in scan_tb.hip: global memory -> LDS directly (source-side swizzle, asm)
  unsigned src_off[C::kLoads];
#pragma unroll
  for (int j = 0; j < C::kLoads; ++j) {
    const int P = j * kT + tid;
    const int r = P / C::kCpr, cp = P % C::kCpr;
    src_off[j] = (unsigned)(r * C::kCpr + ((cp & ~15) | ((cp ^ r) & 15))) * 16u;
  }
  const char* slab = reinterpret_cast<const char*>(a.slab);
  const size_t last_chunk = (size_t)a.n_rows * (D * 2) - 16;
  const int n_full = a.n_rows / 32;
  const unsigned lds_wave = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr_t)tile_buf + (unsigned)wave * 1024u);
  // (a second look-ahead tile measured no faster here: 592 vs 558 us on C3)
  auto dma_tile = [&](int tile_, int buf) {
    const int tile = __builtin_amdgcn_readfirstlane(tile_);
    const unsigned dst0 = lds_wave + (unsigned)(buf * C::kTileBytes);
    if (tile < n_full) {
      const char* base = uniform_ptr(slab + (size_t)tile * C::kTileBytes);
#pragma unroll
      for (int j = 0; j < C::kLoads; ++j) {
        const unsigned dst = dst0 + (unsigned)(j * kT * 16);
        lds_dma16(dst, src_off[j], base);
      }
    } else {   // the ragged last tile: clamp every lane to the slab's last 16 bytes (rows past the end never rank)
#pragma unroll
      for (int j = 0; j < C::kLoads; ++j) {
        size_t off = (size_t)tile * C::kTileBytes + src_off[j];
        off = off > last_chunk ? last_chunk : off;
        const char* p = slab + off;
        const unsigned dst = dst0 + (unsigned)(j * kT * 16);
        lds_dma16(dst, p);
      }
    }
  };

  int t = stream;
  dma_tile(t, 0);

  // ---- this wave's half of its pair's 32 queries: lane (n, h) holds Q[n][kh DH + 16 ks + 8 h .. + 8]
  const int qn = lane & 31, h = lane >> 5;
  const int qi = qblock * 128 + pair * 32 + qn;
  const bool q_valid = qi < a.nq;
  f16x8 qf[C::kKsteps];
  {
    const _Float16* qrow = a.q + (size_t)(q_valid ? qi : 0) * D + kh * C::DH + h * 8;
#pragma unroll
    for (int ks = 0; ks < C::kKsteps; ++ks) {
      const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      qf[ks] = q_valid ? *reinterpret_cast<const f16x8*>(qrow + ks * 16) : z;
    }
#pragma unroll
    for (int ks = 0; ks < C::kKsteps; ++ks) {
      f16x8 x = qf[ks];
      asm volatile("" : "+v"(x));
      qf[ks] = x;
    }
  }
  // A fragment of k-step ks: row (l & 31), 16-byte chunk kh DH/8 + 2 ks + h (DH/8 is a multiple of 16, so the
  // swizzle group pattern of the half row equals that of a whole one)
  int a_off[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a_off[j] = qn * (C::kCpr * 16) + kh * (C::DH * 2) + (((2 * j + h) ^ qn) & 15) * 16;

  float ts[K];
  int tr[K];
#pragma unroll
  for (int j = 0; j < K; ++j) { ts[j] = kNegInf; tr[j] = -1; }
  float px = kNegInf;
  int pr = -1;
  auto insert = [&](float x, int xr) {
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const bool c = x > ts[j];
      const float s_old = ts[j];
      const int r_old = tr[j];
      ts[j] = c ? x : s_old;
      tr[j] = c ? xr : r_old;
      x = c ? s_old : x;
      xr = c ? r_old : xr;
    }
  };
  auto sweep = [&](const char* buf) {
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < C::kKsteps; ++ks) {
      const f16x8 af = *reinterpret_cast<const f16x8*>(buf + a_off[ks & 7] + (ks >> 3) * 256);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, qf[ks], acc, 0, 0, 0);
    }
    return acc;
  };
  // tile te (the ie-th of the stream): own half-sums + the partner's (exchange buffer ie & 1) -> tile best -> list
  auto select = [&](const f32x16& own, int te, int ie) {
    const float* xb = xbuf + ((pair * 2 + (ie & 1)) * 16) * 64 + lane;
    float x = kNegInf;
    const int row_base = te * 32 + 4 * h;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float sc = own[r] + xb[r * 64];
      const int row = row_base + 8 * (r >> 2) + (r & 3);
      x = (te < n_full || row < a.n_rows) ? __builtin_fmaxf(x, sc) : x;
    }
    x = pair_max(x);
    if ((ie & 1) == h) { px = x; pr = te * 32; }
    if (ie & 1) {
      insert(px, pr);
      px = kNegInf;
      pr = -1;
    }
  };

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  f32x16 acc_prev = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int cur = 0, it = 0;
  for (; t < a.n_tiles; t += nwg) {
    if (t + nwg < a.n_tiles) dma_tile(t + nwg, cur ^ 1);
    if (wave_active) {
      const char* buf = tile_buf + cur * C::kTileBytes;
      if (kh == 0) {   // last tile's selection (the partner's half-sums are behind the barrier), then this tile's MFMAs
        if (it > 0) select(acc_prev, t - nwg, it - 1);
        acc_prev = sweep(buf);
      } else {         // second half of k: MFMAs, then hand the half-sums over
        const f32x16 acc = sweep(buf);
        float* xb = xbuf + ((pair * 2 + (it & 1)) * 16) * 64 + lane;
#pragma unroll
        for (int r = 0; r < 16; ++r) xb[r * 64] = acc[r];
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
    ++it;
  }
  if (wave_active && kh == 0) {
    if (it > 0) select(acc_prev, t - nwg, it - 1);   // the deferred last tile
    if (it & 1) insert(px, pr);                      // odd tile count: the last (even) tile is still pending
    if (q_valid) {                                   // [nq, nwg, kp = 2 K]: lane half h owns slots h K .. h K + K - 1
      const size_t o = ((size_t)qi * nwg + stream) * a.kp + (size_t)h * K;
#pragma unroll
      for (int j = 0; j < K; ++j) {
        a.part_scores[o + j] = ts[j];
        a.part_rows[o + j] = tr[j];
      }
    }
  }
}

template <int D, int K>
int launch_ks(const ScanArgs& a, hipStream_t stream) {
  using C = KsCfg<D>;
  static bool done = false;
  auto kernel = &scan_wide_ks_kernel<D, K>;
  if (!done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, C::kLds);
    if (e != hipSuccess) return (int)e;
    done = true;
  }
  hipLaunchKernelGGL(kernel, dim3(a.nqb * a.nwg), dim3(512), C::kLds, stream, a);
  return (int)hipGetLastError();
}

template <int D>
int launch_ks_k(const ScanArgs& a, hipStream_t stream) {
  const int kk = scan_wide_slots(a.k);
  if (kk == 4) return launch_ks<D, 4>(a, stream);
  if (kk == 10) return launch_ks<D, 10>(a, stream);
  return launch_ks<D, 16>(a, stream);
}

}  // namespace

// 1: this launch (fp16 slab) takes the split-contraction kernel; CRS_SCAN_WIDE=0 disables it with scan_wide.hip
bool scan_wide_ks_applies(int nq, int k, int pdim) {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("CRS_SCAN_WIDE");
    on = (e && e[0] == '0') ? 0 : 1;
  }
  return on && nq > 64 && k <= 16 && pdim == 768;   
}

int scan_launch_wide_ks(const ScanArgs& a, int pdim, hipStream_t stream) {
  switch (pdim) {
    case 768: return launch_ks_k<768>(a, stream);
    default: return -1;
  }
}

}  // namespace crs
