// scan_w1.hip -- the large-batch scan for 768-element rows (> 64 queries per launch: BASELINE config C3 = 1 M x 768 x
// 256 queries).  Replaces the HNSW walk behind collection.query (/root/reference/rag/indexing.py:171-176) for query
// batches; supersedes round 1's split-contraction kernel (scan_wide_ks.hip, removed).
//
//   * 8 waves per workgroup (two per SIMD), 32 queries each: 256 queries per workgroup, ONE workgroup per CU, so a
//     staged tile is fetched once per 256 queries (the split-contraction kernel served 128 and read every tile twice
//     through L2);
//   * a wave keeps the MFMA B fragments of its 32 queries for the FULL depth in registers -- 192 of its 256: the first 32
//     k-steps in the accumulator half of the register file ("a" operands; MFMA A/B may be AGPRs), the last 16 in arch
//     VGPRs (hipcc splits a 256-register wave 128 + 128 and offers no source-level way to move the line);
//   * the k-loop is written in asm, two k-steps per statement: hipcc, left to itself, funnels every A fragment through
//     one register quad (ds_read -> lgkmcnt(0) -> MFMA, the LDS latency exposed at every k-step).  A statement first
//     issues the two ds_read_b128 of the NEXT pair, then multiplies the two fragments the previous statement issued,
//     each behind a counted lgkmcnt (form (ii) of cdna_hip_programming.md section 5.7: every in-flight destination is a
//     "+v" operand of the statement that waits for it).  The eight A-fragment row addresses of a tile are ONE register
//     XOR an immediate (see a_issue2);
//   * selection is a DUMP: per (query, tile) only the tile's best score leaves the kernel -- 4 + 4 bytes into the
//     partial list [nq, streams, tiles per stream] -- no threshold, no register chain, no data-dependent branch, any
//     k <= 64.  The merge kernel picks the k best tiles per query and scan_refine.hip re-opens them (exactness: the k
//     best tiles by (best score desc, tile asc) contain every row of the exact top-k, see scan_refine.hip);
//   * three tile buffers: up to two 48 KB tiles in flight per workgroup by LDS-DMA (source-side swizzle), the transfer
//     of tile i + 2 issued piecewise between the MFMA pairs of tile i, one barrier per tile; the two waves of a SIMD are
//     staggered (one folds / stores while the other multiplies).
// Measured (tools/scan_w1_probe, tools/ab_wide.sh, same box): C3 0.552-0.567 ms against 0.557-0.572 for the kernel it
// replaces -- a tie; what was learned on the way is in DESIGN.md section 7 (a 1 KB transfer instruction holds the issuing
// wave ~95 cycles whatever the queue state, so one wave per SIMD cannot hide transfer issue; the clock sits at
// 1.3-1.6 GHz under this load; the transfer skeleton alone moves 3.9 TB/s).
#include "scan_common.h"

#include <stdlib.h>

namespace crs {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int D>
struct W1Cfg {
  static constexpr int kT = 256;
  static constexpr int kCpr = D / 8;                      // 16-byte chunks per row
  static constexpr int kTileBytes = 32 * D * 2;
  static constexpr int kLoads = kTileBytes / (kT * 16);
  static constexpr int kKsteps = D / 16;
  static constexpr int kBufs = 3;
  static constexpr int kLds = kBufs * kTileBytes;
  static_assert(D % 128 == 0, "rows must keep the 256-byte swizzle groups whole");
  static_assert(kTileBytes % (kT * 16) == 0, "tile must split into whole 16-byte loads");
};

template <int N>
__device__ __forceinline__ void wait_vmcnt_le() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// =================================================================================================================
// Eight waves per workgroup (two per SIMD), 32 queries each: 256 queries per workgroup, one workgroup per CU.
// Why the four-wave form above is not enough: a vector-memory instruction (the 1 KB LDS-DMA transfers included)
// holds the issuing wave for ~90-100 cycles whatever the state of the memory queue (tools/scan_w1_probe: 12
// transfers = 1.1 k cycles per tile and wave, in one burst or spread over the k-loop alike), and with ONE wave per
// SIMD nothing multiplies meanwhile -- 1.7 k cycles of MFMA + 1.1 k of transfer issue per tile.  With two waves
// per SIMD the partner's MFMAs fill those gaps, and with 256 queries per workgroup a staged tile feeds twice the
// arithmetic (48 transfers per 12 288 MFMA cycles per CU instead of per 6 144).  A wave then has 256 registers:
// 192 accumulator-file registers of query fragments (D = 768) and 64 arch VGPRs for everything else -- one
// accumulator set, two pairs of A fragments (k-steps go in pairs), eight row addresses, six transfer offsets.
template <int G, int NG>
struct KPair {   // k-steps 2 G, 2 G + 1
  static constexpr int j0 = (2 * G) & 7, j1 = (2 * G + 1) & 7;
  static constexpr int o0 = ((2 * G) >> 3) * 256, o1 = ((2 * G + 1) >> 3) * 256;
};

// The eight A-fragment row addresses of a tile differ only in their low byte: row (l & 31), 16-byte chunk 2 ks + h
// swizzled by the row = rowaddr + ((32 (ks & 7)) ^ u16), u16 = ((row ^ h) & 15) << 4, and rowaddr is a multiple of 256 --
// so address(ks) = base2 ^ (32 (ks & 7)) with ONE register base2 = rowaddr + u16 per tile.  The XOR is done inside the
// statements (two transient registers) instead of keeping eight addresses live: the wave has 128 arch VGPRs and 64 of
// them hold query fragments.
template <int NG>
__device__ __forceinline__ void a_issue2(f16x8 (&A)[2], unsigned base2) {
  using K = KPair<0, NG>;
  unsigned t0, t1;
  asm volatile("v_xor_b32 %2, %5, %4\n\tv_xor_b32 %3, %6, %4\n\tds_read_b128 %0, %2 offset:%7\n\tds_read_b128 %1, %3 offset:%8"
               : "=&v"(A[0]), "=&v"(A[1]), "=&v"(t0), "=&v"(t1)
               : "v"(base2), "n"(32 * K::j0), "n"(32 * K::j1), "n"(K::o0), "n"(K::o1)
               : "memory");
}

#define CRS_W2_PAIR(NAME, BC)                                                                                                      \
  template <int G, int NG>                                                                                                          \
  __device__ __forceinline__ void NAME(f32x16& acc, f16x8 (&Ac)[2], f16x8 (&An)[2], unsigned base2, const f16x8& b0, const f16x8& b1) { \
    using KN = KPair<(G + 1 < NG) ? G + 1 : G, NG>;                                                                                 \
    unsigned t0, t1;                                                                                                                \
    if constexpr (G == 0) {                                                                                                         \
      asm volatile("v_xor_b32 %[t0], %[k0], %[b2]\n\tv_xor_b32 %[t1], %[k1], %[b2]\n\t"                                               \
                   "ds_read_b128 %[n0], %[t0] offset:%[o0]\n\tds_read_b128 %[n1], %[t1] offset:%[o1]\n\t"                             \
                   "s_waitcnt lgkmcnt(3)\n\tv_mfma_f32_32x32x16_f16 %[acc], %[c0], %[b0], 0\n\t"                                     \
                   "s_waitcnt lgkmcnt(2)\n\tv_mfma_f32_32x32x16_f16 %[acc], %[c1], %[b1], %[acc]"                                    \
                   : [acc] "=&v"(acc), [n0] "=&v"(An[0]), [n1] "=&v"(An[1]), [t0] "=&v"(t0), [t1] "=&v"(t1), [c0] "+v"(Ac[0]),         \
                     [c1] "+v"(Ac[1])                                                                                               \
                   : [b0] BC(b0), [b1] BC(b1), [b2] "v"(base2), [k0] "n"(32 * KN::j0), [k1] "n"(32 * KN::j1), [o0] "n"(KN::o0),      \
                     [o1] "n"(KN::o1)                                                                                               \
                   : "memory");                                                                                                     \
    } else if constexpr (G + 1 < NG) {                                                                                              \
      asm volatile("v_xor_b32 %[t0], %[k0], %[b2]\n\tv_xor_b32 %[t1], %[k1], %[b2]\n\t"                                               \
                   "ds_read_b128 %[n0], %[t0] offset:%[o0]\n\tds_read_b128 %[n1], %[t1] offset:%[o1]\n\t"                             \
                   "s_waitcnt lgkmcnt(3)\n\tv_mfma_f32_32x32x16_f16 %[acc], %[c0], %[b0], %[acc]\n\t"                                \
                   "s_waitcnt lgkmcnt(2)\n\tv_mfma_f32_32x32x16_f16 %[acc], %[c1], %[b1], %[acc]"                                    \
                   : [acc] "+v"(acc), [n0] "=&v"(An[0]), [n1] "=&v"(An[1]), [t0] "=&v"(t0), [t1] "=&v"(t1), [c0] "+v"(Ac[0]),          \
                     [c1] "+v"(Ac[1])                                                                                               \
                   : [b0] BC(b0), [b1] BC(b1), [b2] "v"(base2), [k0] "n"(32 * KN::j0), [k1] "n"(32 * KN::j1), [o0] "n"(KN::o0),      \
                     [o1] "n"(KN::o1)                                                                                               \
                   : "memory");                                                                                                     \
    } else { /* last pair of the tile; 16 + 4 states before hipcc's code may read the accumulators */                              \
      asm volatile("s_waitcnt lgkmcnt(1)\n\tv_mfma_f32_32x32x16_f16 %[acc], %[c0], %[b0], %[acc]\n\t"                                \
                   "s_waitcnt lgkmcnt(0)\n\tv_mfma_f32_32x32x16_f16 %[acc], %[c1], %[b1], %[acc]\n\ts_nop 15\n\ts_nop 3"             \
                   : [acc] "+v"(acc), [c0] "+v"(Ac[0]), [c1] "+v"(Ac[1])                                                            \
                   : [b0] BC(b0), [b1] BC(b1)                                                                                       \
                   : "memory");                                                                                                     \
    }                                                                                                                               \
  }
CRS_W2_PAIR(k_pair_a, "a")
CRS_W2_PAIR(k_pair_v, "v")

template <int NG, int NA, int G = 0, typename Hook>
__device__ __forceinline__ void k_sweep2(f32x16& acc, f16x8 (&A0)[2], f16x8 (&A1)[2], unsigned ad, const f16x8 (&qf)[2 * NG],
                                         const Hook& hook) {
  if constexpr (G < NG) {
    // hipcc splits a 256-register wave 128 arch + 128 accumulator registers and offers no source-level way to move
    // the line: the first NA k-steps' query fragments live in the accumulator file ("a"), the rest in arch VGPRs ("v")
    if constexpr (2 * G < NA) {
      if constexpr ((G & 1) == 0) k_pair_a<G, NG>(acc, A0, A1, ad, qf[2 * G], qf[2 * G + 1]);
      else k_pair_a<G, NG>(acc, A1, A0, ad, qf[2 * G], qf[2 * G + 1]);
    } else {
      if constexpr ((G & 1) == 0) k_pair_v<G, NG>(acc, A0, A1, ad, qf[2 * G], qf[2 * G + 1]);
      else k_pair_v<G, NG>(acc, A1, A0, ad, qf[2 * G], qf[2 * G + 1]);
    }
    hook(G);
    k_sweep2<NG, NA, G + 1>(acc, A0, A1, ad, qf, hook);
  }
}

template <int D>
struct W2Cfg {
  static constexpr int kT = 512;
  static constexpr int kCpr = D / 8;
  static constexpr int kTileBytes = 32 * D * 2;
  static constexpr int kLoads = kTileBytes / (kT * 16);
  static constexpr int kKsteps = D / 16;
  static constexpr int kBufs = (3 * kTileBytes <= 160 * 1024) ? 3 : 2;
  static constexpr int kLds = kBufs * kTileBytes;
  static_assert(D % 128 == 0 && kTileBytes % (kT * 16) == 0, "whole swizzle groups, whole 16-byte loads");
};

template <int D>
__global__ __launch_bounds__(512, 2) void scan_w2_kernel(const ScanArgs a) {
  using C = W2Cfg<D>;
  constexpr int kT = C::kT;
  constexpr int NG = C::kKsteps / 2;
  constexpr int NA = C::kKsteps < 32 ? C::kKsteps : 32;   // k-steps whose query fragments sit in the accumulator file (128 registers)
  static_assert(C::kBufs == 3 && (NG % 2) == 0 && NG >= C::kLoads, "three tile buffers; an even number of k-step pairs; one transfer per pair at most");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwg = CRS_NSTREAMS;
  const int qblock = CRS_QBLOCK, stream = CRS_STREAM;
  WP_DECL;
  const int q0 = qblock * 256 + wave * 32;
  const bool wave_active = q0 < a.nq;                     // wave-uniform

  // source offset of the j-th 1 KB piece: LDS position P = j kT + tid receives chunk swz(P) of the tile.  Recomputed at
  // every use (a handful of VALU per transfer): six more live registers would push a query fragment to scratch
  auto src_off_of = [&](int j) {
    int tid_here = tid;
    asm volatile("" : "+v"(tid_here));   // opaque: keeps the compiler from hoisting all six results out of the tile loop
    const int P = j * kT + tid_here;
    const int r = P / C::kCpr, cp = P % C::kCpr;
    return (unsigned)(r * C::kCpr + ((cp & ~15) | ((cp ^ r) & 15))) * 16u;
  };
  const char* slab = reinterpret_cast<const char*>(a.slab);
  const int n_full = a.n_rows / 32;
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr_t)smem);
  const unsigned lds_wave = lds_base + (unsigned)wave * 1024u;
  // one 1 KB piece of a tile: scalar tile base + 32-bit lane offset, the offset clamped to the tile's last valid
  // 16 bytes (only the ragged last tile of the slab is shorter than kTileBytes; rows past the end never rank)
  auto tile_limit = [&](int tile) {
    const long long left = ((long long)a.n_rows - (long long)tile * 32) * (D * 2);
    return (unsigned)((left < C::kTileBytes ? (left > 16 ? left : 16) : C::kTileBytes) - 16);
  };
  auto dma_part = [&](unsigned lim, unsigned dst0, const char* base, int j) {
    const unsigned so = src_off_of(j);
    const unsigned off = so < lim ? so : lim;
    lds_dma16(dst0 + (unsigned)(j * kT * 16), off, base);
  };
  auto dma_tile = [&](int tile_, int buf) {
    const int tile = __builtin_amdgcn_readfirstlane(tile_);
    const char* base = uniform_ptr(slab + (size_t)tile * C::kTileBytes);
    const unsigned lim = tile_limit(tile);
#pragma unroll
    for (int j = 0; j < C::kLoads; ++j) dma_part(lim, lds_wave + (unsigned)(buf * C::kTileBytes), base, j);
  };

  const int n_mine = (a.n_tiles - stream + nwg - 1) / nwg;
  dma_tile(stream, 0);
  if (n_mine > 1) dma_tile(stream + nwg, 1);

  const int qn = lane & 31, h = lane >> 5;
  f16x8 qf[C::kKsteps];
  {
    const int qi = q0 + qn;
    const bool ok = qi < a.nq;
    const _Float16* qrow = a.q + (size_t)(ok ? qi : 0) * D + h * 8;
#pragma unroll
    for (int ks = 0; ks < C::kKsteps; ++ks) {
      const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      qf[ks] = ok ? *reinterpret_cast<const f16x8*>(qrow + ks * 16) : z;
    }
#pragma unroll
    for (int ks = 0; ks < C::kKsteps; ++ks) {   // pinned for the whole kernel: accumulator file first, arch VGPRs for the rest
      f16x8 x = qf[ks];
      if (ks < NA) asm volatile("" : "+a"(x)); else asm volatile("" : "+v"(x));
      qf[ks] = x;
    }
  }
  const int my_q = q0 + qn;
  const bool store_lane = h == 0 && my_q < a.nq;
  const size_t list0 = ((size_t)(my_q < a.nq ? my_q : 0) * nwg + stream) * a.kp;
  auto put = [&](float v, int slot, int te) {
    if (store_lane) {
      a.part_scores[list0 + slot] = v;
      a.part_rows[list0 + slot] = te * 32;
    }
  };

  if (n_mine > 1) wait_vmcnt_le<C::kLoads>(); else wait_vmcnt_le<0>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  WP_LAP(0);

  // The two waves of a SIMD (w and w + 4) are staggered: the early wave folds and stores a tile's result right after
  // its k-loop, the late wave at the top of the NEXT iteration -- so on every SIMD one wave multiplies while the other
  // selects, stores and addresses, instead of both leaving the matrix pipe idle together behind the tile barrier
  // (tools/scan_w1_probe before the stagger: 3.2 k of 4.9 k cycles per tile in the k-loop).
  const bool late = wave >= 4;
  auto tile_best = [&](const f32x16& acc, int te) {
    float x = kNegInf;
    if (te < n_full) {
      const float m0 = __builtin_fmaxf(__builtin_fmaxf(acc[0], acc[1]), __builtin_fmaxf(acc[2], acc[3]));
      const float m1 = __builtin_fmaxf(__builtin_fmaxf(acc[4], acc[5]), __builtin_fmaxf(acc[6], acc[7]));
      const float m2 = __builtin_fmaxf(__builtin_fmaxf(acc[8], acc[9]), __builtin_fmaxf(acc[10], acc[11]));
      const float m3 = __builtin_fmaxf(__builtin_fmaxf(acc[12], acc[13]), __builtin_fmaxf(acc[14], acc[15]));
      x = __builtin_fmaxf(__builtin_fmaxf(m0, m1), __builtin_fmaxf(m2, m3));
    } else {   // ragged tile: rows >= n_rows hold clamped garbage and must not rank
      const int row_base = te * 32 + 4 * h;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row_base + 8 * (r >> 2) + (r & 3);
        x = (row < a.n_rows) ? __builtin_fmaxf(x, acc[r]) : x;
      }
    }
    return pair_max(x);
  };
  f16x8 A0[2], A1[2];
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  int cur = 0;
  for (int i = 0; i < n_mine; ++i) {
    const int t = stream + i * nwg;
    if (wave_active && late && i > 0) put(tile_best(acc, t - nwg), i - 1, t - nwg);   // late wave: last tile's result first
    // base2 of this tile's buffer (see a_issue2)
    const unsigned ad = lds_base + (unsigned)(cur * C::kTileBytes) + (unsigned)(qn * (C::kCpr * 16)) + (unsigned)(((qn ^ h) & 15) << 4);
    if (wave_active) a_issue2<NG>(A0, ad);
    const bool more = i + 2 < n_mine;
    const int tn = __builtin_amdgcn_readfirstlane(more ? t + 2 * nwg : t);
    const unsigned dn = lds_wave + (unsigned)(((cur + 2) % 3) * C::kTileBytes);
    const char* bn = uniform_ptr(slab + (size_t)tn * C::kTileBytes);
    const unsigned ln = tile_limit(tn);
    // the transfers of tile i + 2 go out between the MFMA pairs (every fourth pair one): the partner wave on this
    // SIMD multiplies while this one sits in a transfer's issue
    constexpr int kEvery = NG / C::kLoads;
#ifdef CRS_STAMPS   /* tools/scan_w1_probe only: a.sched bit 0 = no transfers in the loop, bit 1 = no k-loop */
    const bool dbg_no_dma = a.sched & 1, dbg_no_mma = a.sched & 2;
#else
    constexpr bool dbg_no_dma = false, dbg_no_mma = false;
#endif
    auto hook = [&](int g) {
      if (more && !dbg_no_dma && (g % kEvery) == 0 && g / kEvery < C::kLoads) dma_part(ln, dn, bn, g / kEvery);
    };
    WP_LAP(1);
    if (wave_active && dbg_no_mma) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int g = 0; g < NG; ++g) hook(g);
    } else if (wave_active) {
      k_sweep2<NG, NA>(acc, A0, A1, ad, qf, hook);
      WP_LAP(2);
      if (!late) put(tile_best(acc, t), i, t);
    } else {
#pragma unroll
      for (int g = 0; g < NG; ++g) hook(g);
    }
    WP_LAP(3);
    // tile i + 1 must have landed.  Younger operations that may stay in flight: this iteration's transfer (kLoads)
    // and, for an early wave, its two result stores (vmcnt counts loads, stores and LDS-DMA together, in issue order;
    // a late wave's stores are OLDER than the transfer)
    if (late) {
      if (more) wait_vmcnt_le<C::kLoads>(); else wait_vmcnt_le<0>();
    } else {
      if (more) wait_vmcnt_le<C::kLoads + 2>(); else wait_vmcnt_le<2>();
    }
    WP_LAP(4);
    __builtin_amdgcn_s_barrier();
    WP_LAP(5);
    cur = (cur + 1) % 3;
  }
  if (wave_active && late) put(tile_best(acc, stream + (n_mine - 1) * nwg), n_mine - 1, stream + (n_mine - 1) * nwg);
  if (wave_active && store_lane) {
    for (int s = n_mine; s < a.kp; ++s) {
      a.part_scores[list0 + s] = kNegInf;
      a.part_rows[list0 + s] = -1;
    }
  }
  WP_LAP(6);
  WP_STORE(8);
}

template <int D>
int launch_w2(const ScanArgs& a, hipStream_t stream) {
  using C = W2Cfg<D>;
  static bool done = false;
  auto kernel = &scan_w2_kernel<D>;
  if (!done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, C::kLds);
    if (e != hipSuccess) return (int)e;
    done = true;
  }
  hipLaunchKernelGGL(kernel, dim3(a.nqb * a.nwg), dim3(512), C::kLds, stream, a);
  return (int)hipGetLastError();
}

}  // namespace

// Queries per workgroup when this kernel takes the launch (0: it does not).  CRS_SCAN_W1=0 hands 768-element rows to
// scan_tb.hip's eight-wave form instead (A/B runs).
int scan_w1_queries_per_wg(int nq, int k, int pdim) {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("CRS_SCAN_W1");
    on = (e && e[0] == '0') ? 0 : 1;
  }
  return (on && nq > 64 && k <= 64 && pdim == 768) ? 256 : 0;
}

int scan_launch_w1(const ScanArgs& a, int pdim, hipStream_t stream) {
  switch (pdim) {
    case 768: return launch_w2<768>(a, stream);
    default: return -1;
  }
}

}  // namespace crs
