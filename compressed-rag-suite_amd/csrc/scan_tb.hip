// scan_tb.hip -- "tile best" form of the 16x16x32 scan (fp16 slabs; chain form k <= 16, dump form k <= 64): the default for 64-query
// batches, and for larger batches of rows wider than 512 elements (scan_wide.hip takes the others).
//
// scan.hip filters every score against a running per-query threshold, appends survivors to per-lane
// LDS lists and compacts them with in-register sorting networks.  Measured on C2 (100 k rows, 6 tiles
// per workgroup) that machinery -- bootstrap, two in-loop compactions, the final one -- was a third of
// every wave's 57 k cycles and the kernel sat at 0.34 of the HBM roofline; its 80 KB of LDS per workgroup
// also caps residency at two workgroups (two tiles in flight) per CU.
// Here a wave keeps, per query and tile, only the tile's best score (max tree over the lane's
// accumulators + two row swaps across the query's four lanes) filed under the tile's first row; merge.hip
// picks the k best tiles per query and scan_refine.hip re-opens them (exactness argument there):
//   * K = 0, "dump": streams of at most a few tiles write every representative straight to the
//     workgroup's partial list (slot = tile number in the stream);
//   * K > 0, "chain": the representative goes into a register-resident sorted list of the K best so far
//     (branch-free compare-exchange chain, 5 VALU per slot).  All four lanes of a query insert every tile and
//     so hold the same list: letting them take turns (a quarter of the chain work) left a fold of the four
//     lists for the exit -- 200 dependent compare-exchange steps, 37 k cycles during which the workgroup loads
//     nothing, a tenth of the C4 launch -- whereas the chain inside the loop hides behind the memory waits.
// No LDS lists, no data-dependent branch, 2 tiles of LDS per workgroup, two workgroups per CU (TbCfg).  NW = 4 waves serve 64 queries; NW = 8 serve 128 from one
// staged copy of the tile (rows wider than 512 elements, where scan_wide.hip's 32-query fragments no
// longer fit the register file).

#include "scan_common.h"

#include <stdlib.h>

namespace crs {
namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int D, int TR, int NW>
struct TbCfg {
  static constexpr int kT = NW * 64;
  static constexpr int kCpr = D / 8;
  static constexpr int kTileBytes = TR * D * 2;
  static constexpr int kLoads = kTileBytes / (kT * 16);
  static constexpr int kKsteps = D / 32;
  static constexpr int kRt = TR / 16;
  static constexpr int kLds = 2 * kTileBytes;
  // resident workgroups per CU the kernel is built for (LDS and a 512 / waves-per-SIMD register budget)
  // one look-ahead tile per workgroup and two workgroups per CU: 48 KB in flight per CU is where a plain sweep of
  // HBM peaks as well; a second look-ahead tile or a third workgroup only lengthen the memory queues (C4:
  // 5.6-5.8 TB/s against 6.05).  128-element rows (8 KB tiles) take three.
  static constexpr int kWgpc = NW == 8 ? (D <= 384 ? 2 : 1) : (D <= 128 ? 3 : 2);
  static_assert(D % 128 == 0 && TR % 16 == 0, "row length: multiple of 128 elements; tile rows: multiple of 16");
  static_assert(kTileBytes % (kT * 16) == 0, "tile must split into whole 16-byte loads");
};

template <int D, int TR, int NW, int K>
__global__ __launch_bounds__(NW * 64, (TbCfg<D, TR, NW>::kWgpc * NW) / 4) void scan_tb_kernel(const ScanArgs a) {
  using C = TbCfg<D, TR, NW>;
  constexpr int kT = C::kT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* tile_buf = smem;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwg = CRS_NSTREAMS;
  const int stream = CRS_STREAM;
  WP_DECL;
  const int qbase = CRS_QBLOCK * (NW * 16) + wave * 16;
  const bool wave_active = qbase < a.nq;

  // ---- tile transfer: global memory -> LDS directly (global_load_lds_dwordx4, 16 bytes per lane; no staging
  // registers, no LDS stores).  The LDS side of the transfer is linear -- position P = j * kT + tid of the tile
  // buffer -- so the XOR swizzle of the 16-byte chunks is applied on the SOURCE side: P receives chunk swz(P).
  // Written in asm: the compiler's own waitcnt insertion puts a vmcnt(0) in front of every LDS read that may alias
  // an in-flight transfer, i.e. in front of every fragment read.
  unsigned src_off[C::kLoads];
#pragma unroll
  for (int j = 0; j < C::kLoads; ++j) {
    const int P = j * kT + tid;
    const int r = P / C::kCpr, cp = P % C::kCpr;
    src_off[j] = (unsigned)(r * C::kCpr + ((cp & ~15) | ((cp ^ r) & 15))) * 16u;
  }
  const char* slab = reinterpret_cast<const char*>(a.slab);
  const size_t last_chunk = (size_t)a.n_rows * (D * 2) - 16;
  const int n_full = a.n_rows / TR;
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

#ifdef CRS_TB_CONTIG   /* timing experiment (tools/scan_tb_probe): every stream walks ONE contiguous run of tiles */
  const int tps_ = (a.n_tiles + nwg - 1) / nwg;
  const int tstep = 1, tend = min(a.n_tiles, (stream + 1) * tps_);
  int t = stream * tps_;
#else
  const int tstep = nwg, tend = a.n_tiles;
  int t = stream;
#endif
  // Tile order.  Static part: tiles stream, stream + nwg, ... below t_dyn.  Dynamic part (chain mode, long streams): granules of
  // dyn_mask + 1 consecutive tiles from t_dyn + ticket * (dyn_mask + 1), the ticket drawn from a device-wide counter (one address
  // takes ~90 M atomics/s on this part: a ticket per tile would be the bottleneck) -- a workgroup that was held up (an encoder kernel on its CU, a
  // slower HBM channel) then simply draws fewer tiles instead of making the whole launch wait for its fixed share.  The tile
  // AFTER the look-ahead tile is requested at the top of an iteration (thread 0, one atomic) and handed to the workgroup through
  // LDS behind the iteration's barrier: a ticket has a whole tile time (~2 us) to come back.  Any assignment gives the same lists
  // after the merge: a workgroup still sees its tiles in increasing order (ties keep the earlier tile, as before).
  __shared__ int sh_next[2];
  const int t_dyn = K > 0 ? a.t_dyn : a.n_tiles;
  int tn = t + tstep;     // the look-ahead tile (t_dyn >= 2 nwg: static for the first iteration)
  dma_tile(t, 0);   // before the query fragments are fetched: the two latencies overlap
  const int lr = lane & 15, kq = lane >> 4;
  const int qi = qbase + lr;
  const bool q_valid = qi < a.nq;
  f16x8 qf[C::kKsteps];
  {
    const _Float16* qrow = a.q + (size_t)(q_valid ? qi : 0) * D + kq * 8;
#pragma unroll
    for (int ks = 0; ks < C::kKsteps; ++ks) {
      const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      qf[ks] = q_valid ? *reinterpret_cast<const f16x8*>(qrow + ks * 32) : z;
    }
#pragma unroll
    for (int ks = 0; ks < C::kKsteps; ++ks) {
      f16x8 x = qf[ks];
      asm volatile("" : "+v"(x));
      qf[ks] = x;
    }
  }
  int a_off[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) a_off[m] = lr * (C::kCpr * 16) + (((m * 4 + kq) ^ lr) & 15) * 16;
  // the query's slots in the partial list [nq, nwg, kp]
  const size_t o = ((size_t)(q_valid ? qi : 0) * nwg + stream) * a.kp;
  float* out_s = a.part_scores + o;
  int* out_i = a.part_rows + o;

  // chain mode: this lane's K best tiles so far as (best score, first row), sorted; earlier tile first on ties
  constexpr int KK = K > 0 ? K : 1;
  float ts[KK];
  int tr[KK];
#pragma unroll
  for (int j = 0; j < KK; ++j) { ts[j] = kNegInf; tr[j] = -1; }
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

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  WP_LAP(0);   // prologue

  int cur = 0, it = 0;
  // one iteration: tile t sits in LDS buffer `cur`; tile t + nwg streams into the other buffer while t is multiplied
  while (t < tend) {
    const bool has_next = tn < tend;
    if (has_next) dma_tile(tn, cur ^ 1);
    // the tile after tn: static stride, the next tile of tn's granule, or -- tn is the last static tile / the last tile of its
    // granule -- the first tile of the granule the counter hands out (wave-uniform)
    const bool in_dyn = K > 0 && tn >= t_dyn;
    const bool draw = K > 0 && a.ticket != nullptr && (in_dyn ? ((tn - t_dyn) & a.dyn_mask) == a.dyn_mask : tn + tstep >= t_dyn);
    unsigned tk = 0;
    if constexpr (K > 0) {
      // one lane draws; written as asm under a hand-set exec mask: hipcc's atomicAdd waits for the returned value on the spot
      // (its wave-aggregation rewrite reads it at once), i.e. for the look-ahead transfer issued just above as well -- transfer
      // and multiplication of the dynamic tiles would no longer overlap.  The value is first read behind the vmcnt(0) below.
      const unsigned mask = __builtin_amdgcn_readfirstlane((draw && has_next && wave == 0) ? 1u : 0u);   // lane 0 of wave 0, or nobody
      unsigned long long keep;
      const unsigned zero = 0u, one = 1u;
      asm volatile("s_mov_b64 %1, exec\n\ts_mov_b32 exec_lo, %2\n\ts_mov_b32 exec_hi, 0\n\tglobal_atomic_add %0, %3, %4, %5 sc0\n\ts_mov_b64 exec, %1"
                   : "+v"(tk), "=&s"(keep) : "s"(mask), "v"(zero), "v"(one), "s"(a.ticket) : "memory");
    }
    WP_LAP(1);   // look-ahead issue
#if defined(CRS_TB_EXPERIMENT) && CRS_TB_EXPERIMENT <= 2   /* tools/scan_tb_probe.hip: timing-only builds (1: no MFMA / selection, 3: fragment reads without MFMA, 4: MFMA without fragment reads) */
    if (false) {
#else
    if (wave_active) {
#endif
      const char* buf = tile_buf + cur * C::kTileBytes;
      float best = kNegInf;
#pragma unroll
      for (int rt = 0; rt < C::kRt; ++rt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < C::kKsteps; ++ks) {
#if defined(CRS_TB_EXPERIMENT) && CRS_TB_EXPERIMENT == 3
          const f32x4 af = *reinterpret_cast<const f32x4*>(buf + a_off[ks & 3] + rt * 16 * (C::kCpr * 16) + (ks >> 2) * 256);
          acc[ks & 3] += af[ks & 3];
#elif defined(CRS_TB_EXPERIMENT) && CRS_TB_EXPERIMENT == 4
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(qf[(ks + rt) % C::kKsteps], qf[ks], acc, 0, 0, 0);
#else
          const f16x8 af = *reinterpret_cast<const f16x8*>(buf + a_off[ks & 3] + rt * 16 * (C::kCpr * 16) + (ks >> 2) * 256);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, qf[ks], acc, 0, 0, 0);
#endif
        }
        if (t < n_full) {
          best = fmaxf(best, fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3])));
        } else {   // ragged last tile: rows past the end must not win
          const int row0 = t * TR + rt * 16 + kq * 4;
#pragma unroll
          for (int i = 0; i < 4; ++i) best = (row0 + i < a.n_rows) ? fmaxf(best, acc[i]) : best;
        }
      }
      WP_LAP(3);   // fragment reads + MFMA
      best = quad_max(best);   // the query's four lanes: all rows of the tile
      if constexpr (K == 0) {
        if (q_valid && kq == 0) {
          out_s[it] = best;
          out_i[it] = t * TR;
        }
      } else {
        insert(best, t * TR);   // all four lanes of the query: they hold the same list, so nothing is left to fold at the exit
      }
    }
    WP_LAP(4);   // selection
    int tnn = in_dyn ? tn + 1 : tn + tstep;
    if (has_next) {
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(tk) : : "memory");
      WP_LAP(5);   // wait for the look-ahead tile (and for the ticket)
      if (K > 0 && draw && tid == 0) sh_next[it & 1] = t_dyn + (int)tk * (a.dyn_mask + 1);
      __syncthreads();
      WP_LAP(7);   // barrier
      if (K > 0 && draw) {   // two slots: iteration it + 1 writes the other one
        tnn = __builtin_amdgcn_readfirstlane(sh_next[it & 1]);
        tnn = (unsigned)tnn < (unsigned)tend ? tnn : tend;   // a poisoned counter must not become an address
      }
    }
    cur ^= 1;
    ++it;
    t = tn;
    tn = tnn;
  }
  if (wave_active) {
    if constexpr (K == 0) {
      if (q_valid) {   // slots of tiles this (shorter) stream does not have
        for (int p = it + kq; p < a.kp; p += 4) {
          out_s[p] = kNegInf;
          out_i[p] = -1;
        }
      }
    } else {
      if (q_valid && kq == 0) {    // kp = K
#pragma unroll
        for (int j = 0; j < K; ++j) {
          out_s[j] = ts[j];
          out_i[j] = tr[j];
        }
      }
    }
  }
  WP_LAP(10);   // final fold + write
  WP_STORE(NW);
}

__global__ void tb_zero_ticket_kernel(unsigned* ticket) {
  if (threadIdx.x < 4) ticket[threadIdx.x] = 0u;
}

template <int D, int TR, int NW, int K>
int launch_tb(const ScanArgs& a, hipStream_t stream) {
  using C = TbCfg<D, TR, NW>;
  static bool done = false;
  auto kernel = &scan_tb_kernel<D, TR, NW, K>;
  if (!done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, C::kLds);
    if (e != hipSuccess) return (int)e;
    done = true;
  }
  hipLaunchKernelGGL(kernel, dim3(a.nqb * a.nwg), dim3(NW * 64), C::kLds, stream, a);
  return (int)hipGetLastError();
}

template <int D, int TR, int NW>
int launch_tb_k(const ScanArgs& a, int slots, hipStream_t stream) {
  switch (slots) {
    case 0: return launch_tb<D, TR, NW, 0>(a, stream);
    case 4: return launch_tb<D, TR, NW, 4>(a, stream);
    case 10: return launch_tb<D, TR, NW, 10>(a, stream);
    case 16: return launch_tb<D, TR, NW, 16>(a, stream);
    // 16 < k <= 64 on long streams: a 32- or 64-slot chain (5 VALU per slot and tile, still under the tile's memory time)
    // instead of the threshold kernels.  32 slots (64 registers) fit beside the query fragments of every row length at two
    // waves per SIMD; 64 slots only for 256- / 384-element rows (they spill from 512 on).  10 M x 384, 64 queries, k = 17 / 32:
    // scan 2.36 / 2.13 ms on the threshold kernels -> 1.23 ms, whole search 2.09 / 2.16 -> 1.31 / 1.32.
    case 24: if constexpr (NW == 4) return launch_tb<D, TR, NW, 24>(a, stream); else return -1;   // the store's over-fetch on large shards
    case 32: if constexpr (NW == 4) return launch_tb<D, TR, NW, 32>(a, stream); else return -1;
    // register budgets checked by tools/check_resources.py (no plan-selectable instantiation may touch scratch): 64 slots
    // fit 256-element rows only (384: 20 bytes / lane of scratch), 56 / 48 slots 384-element rows, 48 slots 512 / 640, 40 slots
    // (the reference's 2 k = 40 with rerank on) 768
    case 64: if constexpr (NW == 4 && D == 256) return launch_tb<D, TR, NW, 64>(a, stream); else return -1;
    case 56: if constexpr (NW == 4 && D == 384) return launch_tb<D, TR, NW, 56>(a, stream); else return -1;
    case 48: if constexpr (NW == 4 && D >= 384 && D <= 640) return launch_tb<D, TR, NW, 48>(a, stream); else return -1;
    case 40: if constexpr (NW == 4 && D == 768) return launch_tb<D, TR, NW, 40>(a, stream); else return -1;
    default: return -1;
  }
}

template <int D, int TR>
int launch_tb_d(const ScanArgs& a, int nw, int slots, hipStream_t stream) {
  if constexpr ((TR * D * 2) % (8 * 64 * 16) == 0) {
    if (nw == 8) return launch_tb_k<D, TR, 8>(a, slots, stream);
  }
  return nw == 4 ? launch_tb_k<D, TR, 4>(a, slots, stream) : -1;
}

}  // namespace

// 8 waves need a tile that splits into whole 16-byte loads over 512 threads (not 640- / 896-element rows)
bool scan_tb_has_8_waves(int pdim) { return pdim != 640 && pdim != 896; }

int scan_tb_wg_per_cu(int pdim, int nw) {
  if (nw == 8 && !scan_tb_has_8_waves(pdim)) nw = 4;
  switch (pdim) {
    case 128: return nw == 8 ? TbCfg<128, 32, 8>::kWgpc : TbCfg<128, 32, 4>::kWgpc;
    case 256: return nw == 8 ? TbCfg<256, 32, 8>::kWgpc : TbCfg<256, 32, 4>::kWgpc;
    case 384: return nw == 8 ? TbCfg<384, 32, 8>::kWgpc : TbCfg<384, 32, 4>::kWgpc;
    case 512: return nw == 8 ? TbCfg<512, 32, 8>::kWgpc : TbCfg<512, 32, 4>::kWgpc;
    case 640: return TbCfg<640, 16, 4>::kWgpc;
    case 768: return nw == 8 ? TbCfg<768, 16, 8>::kWgpc : TbCfg<768, 16, 4>::kWgpc;
    case 896: return TbCfg<896, 16, 4>::kWgpc;
    case 1024: return nw == 8 ? TbCfg<1024, 16, 8>::kWgpc : TbCfg<1024, 16, 4>::kWgpc;
    default: return 2;
  }
}

// chain length for 16 < k <= 64 on long streams (0: none -- the threshold kernels take the search)
int scan_tb_long_chain_slots(int pdim, int nw, int k) {
  if (nw != 4 || k <= 16 || k > 64) return 0;
  if (k <= 24) return 24;
  if (k <= 32) return 32;
  if (pdim == 256) return 64;
  if (pdim == 384) return k <= 48 ? 48 : (k <= 56 ? 56 : 0);
  if (pdim == 512 || pdim == 640) return k <= 48 ? 48 : 0;
  if (pdim == 768) return k <= 40 ? 40 : 0;
  return 0;
}

// slots = 0: dump mode (kp = tiles per stream); else chain mode with that many slots (kp = slots)
// the ticket counter is zeroed by a kernel of our own, in stream order: a kernel node in a captured graph like the scan itself
// (a hipMemsetAsync node gave correct lists too, but split the graph's event timing)
int scan_ticket_zero(unsigned* ticket, hipStream_t stream) {
  hipLaunchKernelGGL(tb_zero_ticket_kernel, dim3(1), dim3(64), 0, stream, ticket);
  return (int)hipGetLastError();
}

int scan_launch_tb(const ScanArgs& a, int pdim, int nw, int slots, hipStream_t stream) {
  switch (pdim) {
    case 128: return launch_tb_d<128, 32>(a, nw, slots, stream);
    case 256: return launch_tb_d<256, 32>(a, nw, slots, stream);
    case 384: return launch_tb_d<384, 32>(a, nw, slots, stream);
    case 512: return launch_tb_d<512, 32>(a, nw, slots, stream);
    case 640: return launch_tb_d<640, 16>(a, nw, slots, stream);
    case 768: return launch_tb_d<768, 16>(a, nw, slots, stream);
    case 896: return launch_tb_d<896, 16>(a, nw, slots, stream);
    case 1024: return launch_tb_d<1024, 16>(a, nw, slots, stream);
    default: return -1;
  }
}

}  // namespace crs
