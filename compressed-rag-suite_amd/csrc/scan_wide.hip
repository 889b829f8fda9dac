// scan_wide.hip -- the exact cosine scan for LARGE query batches (65 .. 256 queries per workgroup).
//
// scan.hip serves 64 queries per workgroup: 4 waves x 16 queries, one ds_read_b128 of the slab tile
// per v_mfma_f32_16x16x32_f16.  A batch of 256 queries (BASELINE config #3, and every N-GPU step,
// where each rank scans its shard for the queries of ALL ranks) then needs four workgroups per tile
// stream, i.e. four stagings of every tile (L2 -> VGPR -> LDS) and four times the LDS reads: measured
// on C3 (1 M x 768, 256 queries) that path ran at 23 % MFMA / 36 % LDS / 48 % HBM utilisation with the
// waves parked 47 % of the time -- latency-bound on the re-staging, not on any pipe.
//
// Here ONE workgroup of NW = 4 or 8 waves serves 32 * NW queries from ONE staged copy of the tile:
//   * v_mfma_f32_32x32x16_f16, A = 32 slab rows of the tile (LDS), B = the wave's 32 queries, whose
//     fragments for the full depth D stay in VGPRs for the whole kernel (D/4 registers);
//     one ds_read_b128 now feeds 32 MFMA cycles instead of 16, and a tile is staged once per 256 queries;
//   * result layout: lane l holds query (l & 31), rows (reg & 3) + 8 (reg >> 2) + 4 (l >> 5) of the tile:
//     16 rows of the tile per lane, a query's scores live in a lane PAIR (l, l ^ 32);
//   * selection is "tile best" (scan_refine.hip): per tile the lane pair reduces its 2 x 16 scores to the
//     tile's best row (one xor-shuffle), and that single candidate is inserted into a register-resident
//     sorted list of the K best so far (a branch-free compare-exchange chain, 5 VALU per slot).  The two
//     lanes of a pair take turns -- lane half h inserts the tiles of parity h -- so the chain runs once
//     per TWO tiles; nothing is filtered against a threshold, no LDS lists, no compaction.
//     The first version of this kernel used scan.hip's per-lane LDS candidate lists + wave compaction:
//     measured on C4 with 256 queries (tools/scan_wide_probe), of 1.13 M cycles per wave 0.38 M went to
//     compactions (15 k cycles each, and with 8 waves behind one barrier every one of them stalls the
//     workgroup), 0.26 M to the 16 data-dependent appends per tile and 0.24 M to the MFMAs.
//     At the end every lane writes its K (score, row) pairs; merge.hip picks the k best representatives
//     over all workgroups and scan_refine.hip re-opens their row groups;
//   * tile staging, XOR swizzle (conflict-free for the 32-row fragment reads too: every 16-lane group of
//     a ds_read_b128 covers 16 distinct rows mod 16), early asm loads and the one-barrier double buffer
//     are those of scan.hip.
// k <= 16 only (the launcher falls back to scan.hip otherwise).

#include "scan_common.h"

#include <stdlib.h>

namespace crs {
int scan_wide_slots(int k);
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;




// rows per tile: 32 * RB (RB 32x32 MFMA row blocks).  The 8-wave kernel takes 64-row tiles where the registers
// allow: the per-tile costs that are not MFMAs (barrier, selection chain, load issue, LDS store: ~1.9 k cycles
// per wave and 32-row tile against 1.5 k of matrix work) are then paid once per 64 rows.
template <int D, int NW>
constexpr int wide_rb() { return (NW == 8 && D <= 384) ? 2 : 1; }

template <int D, int NW>
struct WCfg {
  static constexpr int kThreadsW = NW * 64;
  static constexpr int kCpr = D / 8;
  static constexpr int RB = wide_rb<D, NW>();
  static constexpr int TR = 32 * RB;
  static constexpr int kTileBytes = TR * D * 2;
  static constexpr int kLoads = kTileBytes / (kThreadsW * 16);
  static constexpr int kKsteps = D / 16;
  static constexpr int kQStage = NW * 4096;   // prologue: 4 KB per wave for the query transpose, in the second tile buffer
  static constexpr int kLds = 2 * kTileBytes > kTileBytes + kQStage ? 2 * kTileBytes : kTileBytes + kQStage;
  static_assert(D % 128 == 0, "row length must be a multiple of 128 elements");
  static_assert(kTileBytes % (kThreadsW * 16) == 0, "tile must split into whole 16-byte loads");
};

// (one look-ahead tile per workgroup: a second one measured no faster -- the kernel is LDS- / barrier-bound, below)
template <int D, int NW, int K>
__global__ __launch_bounds__(NW * 64, 2) void scan_wide_kernel(const ScanArgs a) {
  using C = WCfg<D, NW>;
  constexpr int kT = C::kThreadsW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* tile_buf = smem;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwg = CRS_NSTREAMS;
  const int qblock = CRS_QBLOCK, stream = CRS_STREAM;
  WP_DECL;
  const bool wave_active = (qblock * (NW * 32) + wave * 32) < a.nq;   // wave-uniform

  // tile transfer as in scan_tb.hip: global memory -> LDS directly, LDS position P = j * kT + tid receives the
  // tile's chunk swz(P) (source-side swizzle); asm, because the compiler would wait for every transfer before
  // every fragment read
  unsigned src_off[C::kLoads];
#pragma unroll
  for (int j = 0; j < C::kLoads; ++j) {
    const int P = j * kT + tid;
    const int r = P / C::kCpr, cp = P % C::kCpr;
    src_off[j] = (unsigned)(r * C::kCpr + ((cp & ~15) | ((cp ^ r) & 15))) * 16u;
  }
  const char* slab = reinterpret_cast<const char*>(a.slab);
  const size_t last_chunk = (size_t)a.n_rows * (D * 2) - 16;
  constexpr int RB = C::RB, WTR = C::TR;
  const int n_full = a.n_rows / WTR;
  const unsigned lds_wave = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr_t)tile_buf + (unsigned)wave * 1024u);
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
  dma_tile(t, 0);   // goes out before the query fragments are fetched, so the two latencies overlap

  // ---- this wave's 32 queries, full depth, as B fragments: lane (n = l & 31, h = l >> 5) holds
  // Q[n][16 ks + 8 h .. + 8] for every k-step
  const int qn = lane & 31, h = lane >> 5;
  const int qi = qblock * (NW * 32) + wave * 32 + qn;
  const bool q_valid = qi < a.nq;
  f16x8 qf[C::kKsteps];
  {
    // Fetched straight into this layout a load instruction touches 32 rows x 2 pieces of 16 bytes -- 64 cache
    // lines per instruction, D/16 instructions per wave: 12 k cycles of the texture addresser per workgroup,
    // 42 % of a 128-query launch over 100 k rows (tools/scan_wide_probe).  So the wave's 32 query rows come in
    // as whole 128-byte column blocks (8 lanes per row), pass through a wave-private 4 KB of the still idle
    // SECOND tile buffer (chunk XOR-swizzled by (row >> 1) & 7) and are read back as fragments.  LDS operations of a
    // wave execute in order, so the block's reads see its writes and the next block's writes come after them.
    constexpr int kBlocks = C::kCpr / 8;   // 128-byte column blocks of a query row
    constexpr int kGroup = 3;              // blocks whose loads are in flight together (12 x 16 bytes per lane)
    char* qs = tile_buf + C::kTileBytes + wave * 4096;
    const int q0 = qblock * (NW * 32) + wave * 32;
    const char* qbytes = reinterpret_cast<const char*>(a.q);
    const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int g0 = 0; g0 < kBlocks; g0 += kGroup) {
      u32x4 tmp[kGroup * 4];
#pragma unroll
      for (int p = 0; p < kGroup; ++p) {
        if (g0 + p < kBlocks) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int c = j * 64 + lane, q = c >> 3, col = c & 7;
            const int qq = q0 + q < a.nq ? q0 + q : a.nq - 1;
            tmp[p * 4 + j] = *reinterpret_cast<const u32x4*>(qbytes + (size_t)qq * (D * 2) + (g0 + p) * 128 + col * 16);
          }
        }
      }
#pragma unroll
      for (int p = 0; p < kGroup; ++p) {
        if (g0 + p < kBlocks) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int c = j * 64 + lane, q = c >> 3, col = c & 7;
            *reinterpret_cast<u32x4*>(qs + q * 128 + ((col ^ ((q >> 1) & 7)) * 16)) = tmp[p * 4 + j];
          }
#pragma unroll
          for (int ksl = 0; ksl < 4; ++ksl) {
            const int col = 2 * ksl + h;
            const f16x8 v = *reinterpret_cast<const f16x8*>(qs + qn * 128 + ((col ^ ((qn >> 1) & 7)) * 16));
            qf[(g0 + p) * 4 + ksl] = q_valid ? v : z;
          }
        }
      }
    }
#pragma unroll
    for (int ks = 0; ks < C::kKsteps; ++ks) {   // retire these loads here, not somewhere in the loop
      f16x8 x = qf[ks];
      asm volatile("" : "+v"(x));
      qf[ks] = x;
    }
  }
  // A fragment of k-step ks: row (l & 31) of the tile, 16-byte chunk 2 ks + h, through the swizzle
  int a_off[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a_off[j] = qn * (C::kCpr * 16) + (((2 * j + h) ^ qn) & 15) * 16;

  // this lane's K best tiles so far as (best score, first row), sorted: score desc, earlier tile first on ties
  float ts[K];
  int tr[K];
#pragma unroll
  for (int j = 0; j < K; ++j) { ts[j] = kNegInf; tr[j] = -1; }

  float px = kNegInf;   // pending candidate of this lane (see the loop)
  int pr = -1;
  // insert into the sorted list: one compare-exchange per slot, the loser moves on
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

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();   // the first tile has landed, and every wave has its fragments: the second buffer is free
  WP_LAP(0);   // prologue

  // Stagger (MI355X_MICROARCH.md, "two waves per SIMD", item 9).  Waves w and w + 4 share a SIMD, and with
  // one barrier per tile they run in lockstep: both in their MFMA sweep (halving each other's rate), then
  // both in the VALU selection with the matrix pipe idle -- measured 3.7 k cycles per tile against 1.5 k
  // of MFMA work.  So waves 4..7 defer the selection of a tile by one iteration (its 16 accumulators stay
  // in registers across the barrier): on every SIMD one wave multiplies while the other selects.
  const bool late = (NW == 8 && D <= 384 && !(D == 384 && K == 16)) && wave >= 4;   // wave-uniform; off where the
                                                                                      // deferred accumulators would spill
  struct Acc { f32x16 a[RB]; };
  Acc acc_prev;
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_prev.a[rb][r] = 0.f;
  auto sweep = [&](const char* buf) {
    Acc acc;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc.a[rb][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < C::kKsteps; ++ks) {
        const f16x8 af = *reinterpret_cast<const f16x8*>(buf + rb * 32 * (C::kCpr * 16) + a_off[ks & 7] + (ks >> 3) * 256);
        acc.a[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, qf[ks], acc.a[rb], 0, 0, 0);
      }
    }
    return acc;
  };
  // tile te (the ie-th of this stream): the tile's best score for this lane's query -> sorted list.
  // The representative is (best score, first row of the tile): tiles are contiguous row ranges, so on
  // equal scores the lower tile holds the lower rows and no arg-max is needed (scan_refine.hip).
  auto select = [&](const Acc& accs, int te, int ie) {
    float x = kNegInf;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const f32x16& acc = accs.a[rb];
      if (te < n_full) {
        const float m0 = __builtin_fmaxf(__builtin_fmaxf(acc[0], acc[1]), acc[2]);
        const float m1 = __builtin_fmaxf(__builtin_fmaxf(acc[3], acc[4]), acc[5]);
        const float m2 = __builtin_fmaxf(__builtin_fmaxf(acc[6], acc[7]), acc[8]);
        const float m3 = __builtin_fmaxf(__builtin_fmaxf(acc[9], acc[10]), acc[11]);
        const float m4 = __builtin_fmaxf(__builtin_fmaxf(acc[12], acc[13]), acc[14]);
        x = __builtin_fmaxf(x, __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(m0, m1), __builtin_fmaxf(m2, m3)), __builtin_fmaxf(m4, acc[15])));
      } else {   // ragged last tile: rows past the end must not win
        const int row_base = te * WTR + rb * 32 + 4 * h;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = row_base + 8 * (r >> 2) + (r & 3);
          x = (row < a.n_rows) ? __builtin_fmaxf(x, acc[r]) : x;
        }
      }
    }
    x = pair_max(x);                              // both halves: the tile's best
    if ((ie & 1) == h) { px = x; pr = te * WTR; } // lane half h is responsible for the tiles of parity h
    if (ie & 1) {
      insert(px, pr);
      px = kNegInf;
      pr = -1;
    }
  };

  int cur = 0, it = 0;
  // one iteration: tile t sits in LDS buffer `cur`, tile t + nwg streams into the other one
  while (t < a.n_tiles) {
    if (t + nwg < a.n_tiles) dma_tile(t + nwg, cur ^ 1);
    if (wave_active) {
      WP_LAP(1);   // tile-load issue
      const char* buf = tile_buf + cur * C::kTileBytes;
      if (!late) {
        const Acc acc = sweep(buf);
        WP_LAP(3);   // MFMA sweep
        select(acc, t, it);
      } else {       // waves 4..7: last tile's selection first, then this tile's MFMAs
        if (it > 0) select(acc_prev, t - nwg, it - 1);
        WP_LAP(2);
        acc_prev = sweep(buf);
        WP_LAP(3);
      }
    }
    WP_LAP(4);   // selection (waves 0..3)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WP_LAP(6);   // wait for the next tile
    __syncthreads();
    WP_LAP(7);   // barrier
    cur ^= 1;
    ++it;
    t += nwg;
  }

  if (wave_active && late && it > 0) select(acc_prev, t - nwg, it - 1);   // the deferred last tile
  if (wave_active && (it & 1)) insert(px, pr);   // odd tile count: the last (even) tile is still pending
  if (wave_active && q_valid) {   // [nq, nwg, kp = 2 K]: lane half h owns slots h K .. h K + K - 1
    const size_t o = ((size_t)qi * nwg + stream) * a.kp + (size_t)h * K;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      a.part_scores[o + j] = ts[j];
      a.part_rows[o + j] = tr[j];
    }
  }
  WP_LAP(10);   // final flush
  WP_STORE(NW);
}

template <int D, int NW, int K>
int launch_wide(const ScanArgs& a, hipStream_t stream) {
  using C = WCfg<D, NW>;
  static bool done = false;
  auto kernel = &scan_wide_kernel<D, NW, K>;
  if (!done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, C::kLds);
    if (e != hipSuccess) return (int)e;
    done = true;
  }
  hipLaunchKernelGGL(kernel, dim3(a.nqb * a.nwg), dim3(NW * 64), C::kLds, stream, a);
  return (int)hipGetLastError();
}

template <int D, int NW>
int launch_wide_k(const ScanArgs& a, hipStream_t stream) {
  const int kk = scan_wide_slots(a.k);
  if (kk == 4) return launch_wide<D, NW, 4>(a, stream);
  if (kk == 10) return launch_wide<D, NW, 10>(a, stream);
  return launch_wide<D, NW, 16>(a, stream);
}

template <int D>
int launch_wide_d(const ScanArgs& a, int nw, hipStream_t stream) {
  return nw == 8 ? launch_wide_k<D, 8>(a, stream) : launch_wide_k<D, 4>(a, stream);
}

}  // namespace

// Waves per workgroup the wide kernel would use for this launch, 0 = not applicable (use scan.hip).
// CRS_SCAN_WIDE=0 disables it (A/B runs).
int scan_wide_waves(int nq, int k, int pdim) {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("CRS_SCAN_WIDE");
    on = (e && e[0] == '0') ? 0 : 1;
  }
  if (!on || nq <= 64 || k > 16 || pdim > 512) return 0;
  return nq > 128 ? 8 : 4;
}
// rows per tile of the configuration scan_wide_waves() picks
int scan_wide_tile_rows(int nw, int pdim) { return (nw == 8 && pdim <= 384) ? 64 : 32; }
// list slots per lane the kernel is instantiated for (>= k); the partial lists are 2 * this wide
int scan_wide_slots(int k) { return k <= 4 ? 4 : k <= 10 ? 10 : 16; }
// resident workgroups per CU: the query fragments cost D/4 registers per lane -> two waves per SIMD
int scan_wide_wg_per_cu(int nw, int pdim) { return nw == 8 ? 1 : 2; }

int scan_launch_wide(const ScanArgs& a, int pdim, int nw, hipStream_t stream) {
  switch (pdim) {
    case 128: return launch_wide_d<128>(a, nw, stream);
    case 256: return launch_wide_d<256>(a, nw, stream);
    case 384: return launch_wide_d<384>(a, nw, stream);
    case 512: return launch_wide_d<512>(a, nw, stream);
    default: return -1;
  }
}

}  // namespace crs
