// scan_i8.hip -- int8 slab variant of the exact cosine scan + in-kernel top-k (gfx950).
//
// BASELINE config #5: an int8-quantised index (one fp32 scale per row, s = max|x| / 127) searched
// with 16-bit queries.  Halving the bytes per row halves the HBM time of this bandwidth-bound
// kernel, PROVIDED the int8 -> MFMA path costs no VALU work: converting int8 to fp16 in registers
// (perm + pk_add per pair) would make the kernel VALU-bound.  Instead the QUERY is moved to the
// integer domain once, at kernel start: each fp16 query is scaled to 16-bit fixed point
// (qi = rint(q / sq), sq = max|q| / 32512) and split into two balanced int8 digits
// qi = 256 * hi + lo, lo, hi in [-128, 127].  The slab bytes then go from LDS straight into
// v_mfma_i32_16x16x64_i8 (twice: once against hi, once against lo; 64-deep, so the MFMA cycles per
// row equal the fp16 kernel's), and   score = (256 * S_hi + S_lo) * sq * s_row   is exact integer
// arithmetic up to the final fp32 scaling (|S_hi| <= 768 * 127 * 128 < 2^24: no int32 overflow).
// 16-bit fixed point resolves the query finer than fp16 does, so nothing is lost on that side.
//
// Everything else -- tile streaming (TR whole rows = one contiguous block of HBM, 16 B / lane,
// XOR-swizzled LDS image), query fragments resident in VGPRs, per-lane candidate lists, wave-level
// compaction, per-workgroup partial lists -- is shared with scan.hip through scan_common.h.
// The per-row scales a lane needs (4 consecutive rows per 16-row sub-tile) are fetched into VGPRs
// together with the tile they belong to, one tile ahead.
// Algorithmic bytes per launch = n_rows * (D + 4).

#include "scan_common.h"

namespace crs {
namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int D, int TR, int L>
struct CfgI8 {
  static constexpr int kCpr = D / 16;                      // 16-byte chunks per row
  static constexpr int kTileBytes = TR * D;
  static constexpr int kLoads = kTileBytes / (kThreads * 16);
  static constexpr int kKsteps = D / 64;
  static constexpr int kRt = TR / 16;
  static constexpr int kListBytes = kWaves * L * 64 * 4;
  static constexpr int kLds = 2 * kTileBytes + 2 * kListBytes;
  static_assert(D % 256 == 0, "int8 rows must be a multiple of 256 bytes");
  static_assert(kTileBytes % (kThreads * 16) == 0, "tile must split into whole 16-byte loads");
};

// TBK = -1: threshold + LDS lists + compaction (k > 16).  TBK >= 0: tile-best selection as in scan_tb.hip --
// 0 "dump" (short streams: every tile's best score goes to the partial list), > 0 "chain" (that many
// register-resident slots); scan_refine.hip's int8 kernel re-opens the winning tiles.
template <int D, int TR, int L, int TBK>
__global__ __launch_bounds__(kThreads, 2) void scan_i8_kernel(const ScanArgs a) {
  using C = CfgI8<D, TR, L>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* tile_buf = smem;
  float* sbuf_all = reinterpret_cast<float*>(smem + 2 * C::kTileBytes);
  int* ibuf_all = reinterpret_cast<int*>(smem + 2 * C::kTileBytes + C::kListBytes);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, kq = lane >> 4;
  const int nwg = CRS_NSTREAMS;
  WP_DECL;
  const bool wave_active = (CRS_QBLOCK * 64 + wave * 16) < a.nq;
  float* sbuf = sbuf_all + wave * (L * 64);
  int* ibuf = ibuf_all + wave * (L * 64);

  // ---- tile transfer (as scan_tb.hip): global memory -> LDS directly; LDS position P = j * 256 + tid receives the
  // tile's chunk swz(P) (the swizzle is applied on the source side, the LDS side of the transfer is linear)
  unsigned src_off[C::kLoads];
#pragma unroll
  for (int j = 0; j < C::kLoads; ++j) {
    const int P = j * kThreads + tid;
    const int r = P / C::kCpr, c = P % C::kCpr;
    src_off[j] = (unsigned)(r * C::kCpr + ((c & ~15) | ((c ^ r) & 15))) * 16u;
  }
  const unsigned lds_wave = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr_t)tile_buf + (unsigned)wave * 1024u);
  const char* slab = reinterpret_cast<const char*>(a.slab);
  const size_t last_chunk = (size_t)a.n_rows * D - 16;
  const int n_full = a.n_rows / TR;

  // the tile's TR row scales travel the same way, into a 128-byte slot per tile buffer behind everything else: ONE
  // 16-byte-per-lane transfer by wave 0 instead of two register loads in every wave (a load instruction costs a wave
  // ~200 cycles of issue under memory back-pressure, and the iteration is issue time + latency)
  constexpr int kScOff = TBK < 0 ? C::kLds : 2 * C::kTileBytes;
  char* sc_lds = smem + kScOff;
  const unsigned sc_m0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr_t)sc_lds);
  auto load_tile = [&](int tile, int buf) {
    const unsigned dst0 = lds_wave + (unsigned)(buf * C::kTileBytes);
    if (tile < n_full) {
      const char* base = uniform_ptr(slab + (size_t)tile * C::kTileBytes);
#pragma unroll
      for (int j = 0; j < C::kLoads; ++j) {
        const unsigned dst = dst0 + (unsigned)(j * kThreads * 16);
        lds_dma16(dst, src_off[j], base);
      }
      if (wave == 0 && lane < TR / 4) {
        const float* sb = uniform_ptr(a.scales + (size_t)tile * TR);
        const unsigned off = (unsigned)lane * 16u, dst = sc_m0 + (unsigned)(buf * (TR * 4));
        lds_dma16(dst, off, sb);
      }
    } else {
#pragma unroll
      for (int j = 0; j < C::kLoads; ++j) {
        size_t off = (size_t)tile * C::kTileBytes + src_off[j];
        off = off > last_chunk ? last_chunk : off;
        const char* p = slab + off;
        const unsigned dst = dst0 + (unsigned)(j * kThreads * 16);
        lds_dma16(dst, p);
      }
      if (wave == 0 && lane < TR) {   // the slab's one ragged tile (or past the end): ordinary loads, clamped rows
        const long row = min((long)tile * TR + lane, (long)a.n_rows - 1);
        reinterpret_cast<float*>(sc_lds + buf * (TR * 4))[lane] = a.scales[row];
      }
    }
  };
  // wait for the look-ahead tile and its scales (in LDS once vmcnt says so)
  unsigned tk = 0;   // the ticket in flight (dynamic tile schedule, as scan_tb.hip: chain modes on long streams)
  auto park_tile = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(tk) : : "memory");
    WP_LAP(5);   // wait for the look-ahead tile (and the ticket)
  };

  int t = CRS_STREAM;
  const int t_dyn = TBK > 0 ? a.t_dyn : a.n_tiles;
  int tn = t + nwg;
  int* sh_next = reinterpret_cast<int*>(sc_lds + 2 * TR * 4);   // two slots behind the row-scale slots
  load_tile(t, 0);

  // ---- this wave's queries -> 16-bit fixed point -> two int8 digit planes, resident in VGPRs.
  // B operand of v_mfma_i32_16x16x64_i8: lane holds query (lane & 15), k = 64 ks + 16 kq + j, j = 0..15.
  const int qi = CRS_QBLOCK * 64 + wave * 16 + lr;
  const bool q_valid = qi < a.nq;
  const _Float16* qrow = a.q + (size_t)(q_valid ? qi : 0) * D + kq * 16;
  float amax = 0.f;
#pragma unroll
  for (int ks = 0; ks < C::kKsteps; ++ks) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const f16x8 v = *reinterpret_cast<const f16x8*>(qrow + ks * 64 + h * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf((float)v[e]));
    }
  }
  amax = fmaxf(amax, __shfl_xor(amax, 16));
  amax = fmaxf(amax, __shfl_xor(amax, 32));
  const float qscale = (q_valid && amax > 0.f) ? amax / 32512.0f : 1.0f;
  i32x4 qhi[C::kKsteps], qlo[C::kKsteps];
#pragma unroll
  for (int ks = 0; ks < C::kKsteps; ++ks) {
    unsigned hw[4] = {0u, 0u, 0u, 0u}, lw[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const f16x8 v = *reinterpret_cast<const f16x8*>(qrow + ks * 64 + h * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int q16 = q_valid ? (int)rintf((float)v[e] / qscale) : 0;
        const int lo = ((q16 + 128) & 255) - 128;
        const int hi = (q16 - lo) >> 8;
        const int j = h * 8 + e;
        hw[j >> 2] |= (unsigned)(hi & 255) << ((j & 3) * 8);
        lw[j >> 2] |= (unsigned)(lo & 255) << ((j & 3) * 8);
      }
    }
    qhi[ks] = i32x4{(int)hw[0], (int)hw[1], (int)hw[2], (int)hw[3]};
    qlo[ks] = i32x4{(int)lw[0], (int)lw[1], (int)lw[2], (int)lw[3]};
  }
  int a_off[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) a_off[m] = lr * (C::kCpr * 16) + (((m * 4 + kq) ^ lr) & 15) * 16;

  float tau = q_valid ? kNegInf : __builtin_huge_valf();
  int cnt = 0;
  unsigned* tau_pub = (TBK < 0 && a.tau_shared && q_valid) ? a.tau_shared + qi : nullptr;
  unsigned tg = 0;
  // tile-best state (TBK >= 0): see scan_tb.hip
  constexpr int KK = TBK > 0 ? TBK : 1;
  float ts[KK];
  int tr[KK];
#pragma unroll
  for (int j = 0; j < KK; ++j) { ts[j] = kNegInf; tr[j] = -1; }
  int it = 0;
  auto insert = [&](float x, int xr) {
#pragma unroll
    for (int j = 0; j < KK; ++j) {
      const bool c = x > ts[j];
      const float s_old = ts[j];
      const int r_old = tr[j];
      ts[j] = c ? x : s_old;
      tr[j] = c ? xr : r_old;
      x = c ? s_old : x;
      xr = c ? r_old : xr;
    }
  };
  const size_t po = ((size_t)(q_valid ? qi : 0) * nwg + CRS_STREAM) * a.kp;   // this query's partial list [nq, nwg, kp]

  park_tile();
  __syncthreads();
  WP_LAP(0);   // prologue

  int cur = 0;
  while (t < a.n_tiles) {
    f32x4 sc_use[C::kRt];
#pragma unroll
    for (int rt = 0; rt < C::kRt; ++rt)   // this lane's rows 16 rt + 4 kq .. + 3
      sc_use[rt] = *reinterpret_cast<const f32x4*>(sc_lds + cur * (TR * 4) + (rt * 16 + kq * 4) * 4);
    const bool has_next = tn < a.n_tiles;
    if (has_next) load_tile(tn, cur ^ 1);
    const bool in_dyn = TBK > 0 && tn >= t_dyn;
    const bool draw = TBK > 0 && a.ticket != nullptr && (in_dyn ? ((tn - t_dyn) & a.dyn_mask) == a.dyn_mask : tn + nwg >= t_dyn);
    if constexpr (TBK > 0) {   // one lane draws the granule after tn's (scan_tb.hip explains the asm form)
      const unsigned mask = __builtin_amdgcn_readfirstlane((draw && has_next && wave == 0) ? 1u : 0u);
      unsigned long long keep;
      const unsigned zero = 0u, one = 1u;
      tk = 0;
      asm volatile("s_mov_b64 %1, exec\n\ts_mov_b32 exec_lo, %2\n\ts_mov_b32 exec_hi, 0\n\tglobal_atomic_add %0, %3, %4, %5 sc0\n\ts_mov_b64 exec, %1"
                   : "+v"(tk), "=&s"(keep) : "s"(mask), "v"(zero), "v"(one), "s"(a.ticket) : "memory");
    }
    WP_LAP(1);   // look-ahead issue
    if (tau_pub) {
      tg = __hip_atomic_load(tau_pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (wave_active) {
      const char* buf = tile_buf + cur * C::kTileBytes;
      float best = kNegInf;
#pragma unroll
      for (int rt = 0; rt < C::kRt; ++rt) {
        i32x4 acc_hi = {0, 0, 0, 0}, acc_lo = {0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < C::kKsteps; ++ks) {
          const i32x4 af = *reinterpret_cast<const i32x4*>(buf + a_off[ks & 3] + rt * 16 * (C::kCpr * 16) + (ks >> 2) * 256);
          acc_hi = __builtin_amdgcn_mfma_i32_16x16x64_i8(af, qhi[ks], acc_hi, 0, 0, 0);
          acc_lo = __builtin_amdgcn_mfma_i32_16x16x64_i8(af, qlo[ks], acc_lo, 0, 0, 0);
        }
        WP_LAP(3);   // fragment reads + MFMA
        const int row0 = t * TR + rt * 16 + kq * 4;
        const f32x4 rsc = sc_use[rt];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float sc = ((float)acc_hi[i] * 256.0f + (float)acc_lo[i]) * qscale * rsc[i];
          const int row = row0 + i;
          if constexpr (TBK >= 0) {
            best = (t < n_full || row < a.n_rows) ? fmaxf(best, sc) : best;   // t < n_full is uniform: full tiles skip the row test
          } else if (sc > tau && row < a.n_rows) {
            sbuf[cnt * 64 + lane] = sc;
            ibuf[cnt * 64 + lane] = row;
            ++cnt;
          }
        }
        if constexpr (TBK < 0) {
          if (__any(cnt > L - 4)) compact<L, false>(sbuf, ibuf, lane, cnt, tau, a.k, nullptr, nullptr, q_valid, 0, tau_pub);
        }
      }
      if constexpr (TBK >= 0) {
        best = quad_max(best);   // the query's four lanes: all 32 rows of the tile
        if constexpr (TBK == 0) {
          if (q_valid && kq == 0) {
            a.part_scores[po + it] = best;
            a.part_rows[po + it] = t * TR;
          }
        } else {
          insert(best, t * TR);   // all four lanes of the query keep the same list (scan_tb.hip): no fold at the exit
        }
      }
    }
    WP_LAP(4);   // scores + selection
    park_tile();
    WP_LAP(6);   // LDS store
    if (tau_pub) {
      unsigned x = tg;
      asm volatile("" : "+v"(x));
      tau = fmaxf(tau, foreign_tau(x));
    }
    if (TBK > 0 && draw && has_next && tid == 0) sh_next[it & 1] = t_dyn + (int)tk * (a.dyn_mask + 1);
    __syncthreads();
    WP_LAP(7);   // barrier
    int tnn = in_dyn ? tn + 1 : tn + nwg;
    if (TBK > 0 && draw && has_next) {
      tnn = __builtin_amdgcn_readfirstlane(sh_next[it & 1]);
      tnn = (unsigned)tnn < (unsigned)a.n_tiles ? tnn : a.n_tiles;   // a poisoned counter must not become an address
    }
    cur ^= 1;
    ++it;
    t = tn;
    tn = tnn;
  }
  if (wave_active) {
    if constexpr (TBK < 0) {
      flush_lists<L>(sbuf, ibuf, lane, cnt, tau, a.k, a.kp, a.part_scores + po, a.part_rows + po, q_valid);
    } else if constexpr (TBK == 0) {
      if (q_valid) {   // slots of tiles this (shorter) stream does not have
        for (int p = it + kq; p < a.kp; p += 4) {
          a.part_scores[po + p] = kNegInf;
          a.part_rows[po + p] = -1;
        }
      }
    } else {
      if (q_valid && kq == 0) {
#pragma unroll
        for (int j = 0; j < KK; ++j) {
          a.part_scores[po + j] = ts[j];
          a.part_rows[po + j] = tr[j];
        }
      }
    }
  }
  WP_LAP(10);
  WP_STORE(kWaves);
}

template <int D, int TR, int L, int TBK>
int launch_i8(const ScanArgs& a, hipStream_t stream) {
  using C = CfgI8<D, TR, L>;
  constexpr int lds = (TBK < 0 ? C::kLds : 2 * C::kTileBytes) + 2 * TR * 4 + 16;   // + the row-scale slots + the next-tile slots
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&scan_i8_kernel<D, TR, L, TBK>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  dim3 grid(a.nqb * a.nwg);           // 1-D; (query block, tile stream) from scan_common.h's grid mapping
  hipLaunchKernelGGL((scan_i8_kernel<D, TR, L, TBK>), grid, dim3(kThreads), lds, stream, a);
  return (int)hipGetLastError();
}

// slots: -1 classic, 0 dump, 4 / 10 / 16 / 32 / 48 chain
template <int D>
int launch_i8_d(const ScanArgs& a, int slots, hipStream_t stream) {
  switch (slots) {
    case 0: return launch_i8<D, 32, 16, 0>(a, stream);
    case 4: return launch_i8<D, 32, 16, 4>(a, stream);
    case 10: return launch_i8<D, 32, 16, 10>(a, stream);
    case 16: return launch_i8<D, 32, 16, 16>(a, stream);
    case 32: return launch_i8<D, 32, 16, 32>(a, stream);   // 16 < k <= 32 on long streams (a 64-slot chain spills at these row lengths)
    case 48: if constexpr (D <= 768) return launch_i8<D, 32, 16, 48>(a, stream); else return -1;   // k <= 48 (the reference's 2 k = 40)
    default: break;
  }
  if (a.k <= 16) return launch_i8<D, 32, 16, -1>(a, stream);
  return launch_i8<D, 32, 32, -1>(a, stream);
}

}  // namespace

int scan_i8_tile_rows() { return 32; }

int scan_i8_long_chain_slots(int pdim, int k) { return (k <= 16 || k > 48) ? 0 : k <= 32 ? 32 : (pdim <= 768 ? 48 : 0); }

int scan_launch_i8(const ScanArgs& a, int pdim, int slots, hipStream_t stream) {
  switch (pdim) {
    case 256: return launch_i8_d<256>(a, slots, stream);
    case 512: return launch_i8_d<512>(a, slots, stream);
    case 768: return launch_i8_d<768>(a, slots, stream);
    case 1024: return launch_i8_d<1024>(a, slots, stream);
    default: return -1;
  }
}

}  // namespace crs
