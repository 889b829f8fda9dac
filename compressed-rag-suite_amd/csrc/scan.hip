// scan.hip -- K9: exact cosine scan + in-kernel top-k over an HBM-resident vector slab (gfx950).
//
// Replaces the arithmetic behind collection.query() (reference rag/indexing.py:171-176; ChromaDB's
// HNSW walk) with a brute-force, exact, bandwidth-bound pass:
//
//   scores[q, r] = <Q[q, :], slab[r, :]>          (unit rows  =>  cosine)
//   per query keep the k best by (score desc, row asc)
//
// Structure (one workgroup = 4 wave64):
//   * the slab is streamed ONCE, in tiles of TR whole rows (a tile is one contiguous
//     TR*D*sizeof(elem) block of HBM, read with 16-byte-per-lane coalesced loads), staged through
//     LDS with an XOR swizzle so the MFMA A-fragment ds_read_b128 is bank-conflict free;
//   * each wave keeps the fragments of ITS 16 queries (full depth D) in VGPRs for the whole kernel,
//     so queries never touch LDS; all four waves read the same slab tile from LDS;
//   * v_mfma_f32_16x16x32_f16 with A = slab rows, B = queries: lane l ends up holding the scores
//     of query (l & 15) for rows 4*(l >> 4) .. +3 of the 16-row sub-tile;
//   * top-k: every lane filters its scores against a per-query running threshold tau and appends
//     survivors to a private LDS list (no atomics).  When a list is nearly full the wave compacts:
//     register sorting networks + two cross-lane bitonic merges give the exact k-th best of the
//     query's candidates, which becomes the new tau; survivors are spread back evenly.
//     Rows are visited in ascending order, so a later row that merely ties tau can never displace
//     an earlier one: the strict compare implements the (score desc, row asc) rule exactly.
//   * each wave finally writes <= k (score, row) pairs per query; merge.hip reduces the per-workgroup
//     lists to the final sorted top-k.
//
// HBM-bound by design: algorithmic bytes per launch = n_rows * D * sizeof(elem).

#include "scan_common.h"

#include <stdlib.h>

namespace crs {

int scan_variant();

namespace {

// ---------------------------------------------------------------- the scan kernels
template <int D, int TR, int L>
struct Cfg {
  static constexpr int kCpr = D / 8;                       // 16-byte chunks per row
  static constexpr int kTileBytes = TR * D * 2;            // fp16
  static constexpr int kLoads = kTileBytes / (kThreads * 16);
  static constexpr int kKsteps = D / 32;
  static constexpr int kRt = TR / 16;
  static constexpr int kListBytes = kWaves * L * 64 * 4;   // per array (scores / rows)
  static_assert(D % 128 == 0, "row length must be a multiple of 128 elements");
  static_assert(TR % 16 == 0, "tile rows must be a multiple of 16");
  static_assert(kTileBytes % (kThreads * 16) == 0, "tile must split into whole 16-byte loads");
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// What every variant shares: query fragments in VGPRs, the swizzled A-fragment read offsets, the
// MFMA sweep of one staged tile and the filter/append/compact epilogue.
template <int D, int TR, int L>
struct WaveState {
  using C = Cfg<D, TR, L>;
  f16x8 qf[C::kKsteps];
  int a_off[4];
  float* sbuf;
  int* ibuf;
  float tau;
  int cnt;
  int lane, lr, kq;
  bool q_valid;
  unsigned* tau_pub;   // this lane's query slot in the shared-threshold array (nullptr: sharing off)
#ifdef CRS_STAMPS
  unsigned long long n_compact = 0, cyc_compact = 0;
#endif

  __device__ __forceinline__ void init(const ScanArgs& a, int wave, int lane_, float* sbuf_all, int* ibuf_all) {
    lane = lane_;
    lr = lane & 15;   // A: row inside the 16-row sub-tile; B/D: query inside the wave's 16
    kq = lane >> 4;   // A/B: which 8-element k-quarter; D: which group of 4 rows
    sbuf = sbuf_all + wave * (L * 64);
    ibuf = ibuf_all + wave * (L * 64);
    const int qi = CRS_QBLOCK * 64 + wave * 16 + lr;
    q_valid = qi < a.nq;
    const _Float16* qrow = a.q + (size_t)(q_valid ? qi : 0) * D + kq * 8;
#pragma unroll
    for (int ks = 0; ks < C::kKsteps; ++ks) {
      f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      qf[ks] = q_valid ? *reinterpret_cast<const f16x8*>(qrow + ks * 32) : z;
    }
    // retire these ordinary loads here, so no compiler-placed vmcnt(0) for them lands in the loop
#pragma unroll
    for (int ks = 0; ks < C::kKsteps; ++ks) {
      f16x8 x = qf[ks];
      asm volatile("" : "+v"(x));
      qf[ks] = x;
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) a_off[m] = lr * (C::kCpr * 16) + (((m * 4 + kq) ^ lr) & 15) * 16;
    tau = q_valid ? kNegInf : __builtin_huge_valf();
    cnt = 0;
    tau_pub = (a.tau_shared && q_valid) ? a.tau_shared + qi : nullptr;
  }

  // CHECK = false: the caller keeps every list at <= L - 4*kRt entries at tile boundaries (workgroup-
  // synchronous compaction, variant A), so no list can overflow inside a tile and no check is needed.
  template <bool CHECK = true>
  __device__ __forceinline__ void tile(const char* buf, int t, const ScanArgs& a) {
#pragma unroll
    for (int rt = 0; rt < C::kRt; ++rt) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < C::kKsteps; ++ks) {
        const f16x8 af = *reinterpret_cast<const f16x8*>(
            buf + a_off[ks & 3] + rt * 16 * (C::kCpr * 16) + (ks >> 2) * 256);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, qf[ks], acc, 0, 0, 0);
      }
      const int row0 = t * TR + rt * 16 + kq * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float sc = acc[i];
        const int row = row0 + i;
        if (sc > tau && row < a.n_rows) {
          sbuf[cnt * 64 + lane] = sc;
          ibuf[cnt * 64 + lane] = row;
          ++cnt;
        }
      }
#ifdef CRS_EXPERIMENT_NO_COMPACT   /* timing-only build (tools/scan_probe): what would the kernel cost with selection for free? */
      if (__any(cnt > L - 4)) { cnt = 0; tau = fmaxf(tau, 0.03f); }
#else
      if (CHECK && __any(cnt > L - 4)) {
#ifdef CRS_STAMPS
        const unsigned long long t0_ = __builtin_amdgcn_s_memtime();
#endif
        compact<L, false>(sbuf, ibuf, lane, cnt, tau, a.k, nullptr, nullptr, q_valid, 0, tau_pub);
#ifdef CRS_STAMPS
        cyc_compact += __builtin_amdgcn_s_memtime() - t0_;
        ++n_compact;
#endif
      }
#endif
    }
  }

  // ---- compaction on demand (variant A calls it on a fixed schedule so that the four waves of a
  // workgroup compact in the SAME tile: with the per-tile barrier, four waves compacting in four
  // different tiles stall the workgroup four times -- the slowest workgroups of a C2 launch spent
  // 16-19 k of their 68 k cycles at barriers waiting for a sibling's compaction).
  __device__ __forceinline__ void compact_now(const ScanArgs& a) {
#ifdef CRS_STAMPS
    const unsigned long long t0_ = __builtin_amdgcn_s_memtime();
#endif
#ifdef CRS_EXPERIMENT_NO_COMPACT
    cnt = 0; tau = fmaxf(tau, 0.03f);
#else
    compact<L, false>(sbuf, ibuf, lane, cnt, tau, a.k, nullptr, nullptr, q_valid, 0, tau_pub);
#endif
#ifdef CRS_STAMPS
    cyc_compact += __builtin_amdgcn_s_memtime() - t0_;
    ++n_compact;
#endif
  }

  // ---- threshold bootstrap (k <= 16, TR == 32): the scores of a stream's first 64 rows stay in
  // registers (16 per lane = exactly the input of kth_of_quad), the exact k-th best of those 64 rows
  // becomes tau, and only rows >= tau are appended.  This replaces "append 64 unfiltered candidates
  // per query, then run a full compaction over them" at the start of every stream.
  float boot[16];

  template <int STAGE>
  __device__ __forceinline__ void tile_boot(const char* buf, int t, const ScanArgs& a) {
    static_assert(C::kRt == 2, "bootstrap is written for 32-row tiles");
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < C::kKsteps; ++ks) {
        const f16x8 af = *reinterpret_cast<const f16x8*>(
            buf + a_off[ks & 3] + rt * 16 * (C::kCpr * 16) + (ks >> 2) * 256);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, qf[ks], acc, 0, 0, 0);
      }
      const int row0 = t * TR + rt * 16 + kq * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) boot[STAGE * 8 + rt * 4 + i] = (row0 + i < a.n_rows) ? acc[i] : kNegInf;
    }
  }

  // t0 / t1: the two tiles whose scores sit in boot[0..7] / boot[8..15]
  __device__ __forceinline__ void boot_select(int t0, int t1, const ScanArgs& a) {
    float s[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) s[j] = boot[j];
    const float t = kth_of_quad(s, a.k);   // -inf while fewer than k valid rows were seen
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float sc = boot[j];
      if (q_valid && sc >= t && sc > kNegInf) {
        const int tile = (j < 8) ? t0 : t1;
        sbuf[cnt * 64 + lane] = sc;
        ibuf[cnt * 64 + lane] = tile * TR + ((j >> 2) & 1) * 16 + kq * 4 + (j & 3);
        ++cnt;
      }
    }
    if (q_valid) tau = fmaxf(tau, t);
    // many ties on the threshold (e.g. duplicate rows) can leave a list too full for the next tile
    if (__any(cnt > L - 4 * C::kRt)) compact_now(a);
  }

  __device__ __forceinline__ void finish(const ScanArgs& a, int wave) {
    const int qi = CRS_QBLOCK * 64 + wave * 16 + lr;
    const size_t o = ((size_t)(q_valid ? qi : 0) * CRS_NSTREAMS + CRS_STREAM) * a.kp;  // [nq, nwg, kp]
    flush_lists<L>(sbuf, ibuf, lane, cnt, tau, a.k, a.kp, a.part_scores + o, a.part_rows + o, q_valid);
  }
};

// ---- variant A: register-staged double buffer, 2 workgroups per CU.
// ASM_LOADS (name kept from round 1; there is no inline-asm register load left anywhere in the library): this
// instantiation also polls the shared threshold.  All global loads are compiler-visible -- an asm load whose
// destination the compiler could copy or spill before the counted wait produced one wrong answer in round 1.
// These threshold kernels now only serve 16 < k <= 64 on long streams; everything else runs the tile-best kernels.
// BOOT: peel the first two tiles of the stream and bootstrap the threshold from them in registers.
// A separate instantiation (not a runtime flag): with the peeled code present hipcc schedules the
// steady-state loop ~10 % slower (C4 177 -> 200 us), while short streams (C2, 6 tiles per workgroup)
// gain 5 % from it; the launcher picks per launch.
template <int D, int TR, int L, bool ASM_LOADS, bool BOOT>
__global__ __launch_bounds__(kThreads, 2) void scan_f16_kernel(const ScanArgs a) {
  using C = Cfg<D, TR, L>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* tile_buf = smem;
  float* sbuf_all = reinterpret_cast<float*>(smem + 2 * C::kTileBytes);
  int* ibuf_all = reinterpret_cast<int*>(smem + 2 * C::kTileBytes + C::kListBytes);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwg = CRS_NSTREAMS;
  const bool wave_active = (CRS_QBLOCK * 64 + wave * 16) < a.nq;  // wave-uniform

  CRS_STAMP_REAL(62);
  CRS_STAMP(0);

  // staging geometry: load j of this thread covers 16-byte chunk P = j*256 + tid of the tile
  int lds_dst[C::kLoads];
#pragma unroll
  for (int j = 0; j < C::kLoads; ++j) {
    const int P = j * kThreads + tid;
    const int r = P / C::kCpr, c = P % C::kCpr;
    lds_dst[j] = (r * C::kCpr + ((c & ~15) | ((c ^ r) & 15))) * 16;
  }
  const char* slab = reinterpret_cast<const char*>(a.slab);
  const size_t last_chunk = (size_t)a.n_rows * (D * 2) - 16;
  const int n_full = a.n_rows / TR;  // tiles that need no clamping

  u32x4 st[C::kLoads];
  unsigned tg = 0;   // shared threshold fetched along with the next tile (ASM_LOADS only), applied one tile late
  auto load_tile = [&](int tile_) {
    const int tile = __builtin_amdgcn_readfirstlane(tile_);   // uniform by construction; make it provable
    if (tile < n_full) {
      const char* base = uniform_ptr(slab + (size_t)tile * C::kTileBytes);  // SGPR base + 32-bit lane offset
#pragma unroll
      for (int j = 0; j < C::kLoads; ++j) {
        const unsigned off = (unsigned)(j * kThreads + tid) * 16u;
        st[j] = *reinterpret_cast<const u32x4*>(base + off);   // compiler-visible: it places the waits itself
      }
    } else {  // ragged last tile, or past the end: clamp every lane to the slab's last 16 bytes
#pragma unroll
      for (int j = 0; j < C::kLoads; ++j) {
        size_t off = (size_t)tile * C::kTileBytes + (size_t)(j * kThreads + tid) * 16;
        off = off > last_chunk ? last_chunk : off;
        const char* p = slab + off;
        st[j] = *reinterpret_cast<const u32x4*>(p);
      }
    }
  };
  auto fetch_tau = [&](const unsigned* p) {
    if (ASM_LOADS && p) {
      tg = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  auto park_tile = [&](char* dst) {

#pragma unroll
    for (int j = 0; j < C::kLoads; ++j) *reinterpret_cast<u32x4*>(dst + lds_dst[j]) = st[j];
  };

  // first tile's loads go out before the query fragments are fetched, so the two latencies overlap
  int t = CRS_STREAM;
  load_tile(t);
  WaveState<D, TR, L> w;
  w.init(a, wave, lane, sbuf_all, ibuf_all);
  CRS_STAMP(1);
  park_tile(tile_buf);
  __syncthreads();
  CRS_STAMP(2);

  int cur = 0;
  int it = 0;
  // Workgroup-synchronous compaction by SCHEDULE: every wave compacts at the same tile counts
  // (3, 4, 6, 8, 12, 16, 24, 32, ... : between two of them a query gains ~k new candidates), so the
  // four compactions of a workgroup overlap in time instead of stalling the per-tile barrier one
  // after the other.  No communication is needed (the LDS budget of 2 workgroups / CU is used to the
  // byte); the in-tile overflow check stays as a safety net for outlier lanes.
  const int sched = a.sched;
  auto scheduled = [sched](int n) {
    if (sched == 0 || n < 3) return false;
    const bool pow2 = (n & (n - 1)) == 0;
    if (sched == 2) return pow2;
    return pow2 || ((n % 3) == 0 && ((n / 3) & (n / 3 - 1)) == 0);
  };
  if constexpr (BOOT && L == 16 && C::kRt == 2) {
    if (t + nwg < a.n_tiles) {   // uniform: this stream has at least two tiles
      const int t0 = t, t1 = t + nwg;
      load_tile(t1);
      if (wave_active) w.template tile_boot<0>(tile_buf, t0, a);
      park_tile(tile_buf + C::kTileBytes);
      __syncthreads();
      load_tile(t1 + nwg);
      if (wave_active) {
        w.template tile_boot<1>(tile_buf + C::kTileBytes, t1, a);
        w.boot_select(t0, t1, a);
      }
      park_tile(tile_buf);
      __syncthreads();
      t = t1 + nwg;
      it = 2;
    }
  }
  for (; t < a.n_tiles; t += nwg) {
    load_tile(t + nwg);
    fetch_tau(w.tau_pub);
    if (wave_active) {
      if (scheduled(it)) w.compact_now(a);
      w.template tile<true>(tile_buf + cur * C::kTileBytes, t, a);
    }
    if (it < 14) CRS_STAMP(3 + 3 * it);
    park_tile(tile_buf + (cur ^ 1) * C::kTileBytes);
    if (ASM_LOADS && w.tau_pub) w.tau = fmaxf(w.tau, foreign_tau(tg));
    if (it < 14) CRS_STAMP(4 + 3 * it);
    __syncthreads();
    if (it < 14) CRS_STAMP(5 + 3 * it);
    cur ^= 1;
    ++it;
  }
  CRS_STAMP(58);
  if (wave_active) w.finish(a, wave);
  CRS_STAMP(59);
  CRS_STAMP_REAL(63);
#ifdef CRS_STAMPS
  if (a.stamps && (threadIdx.x & 63) == 0) {
    a.stamps[((size_t)CRS_STREAM * 4 + (threadIdx.x >> 6)) * 64 + 60] = w.n_compact;
    a.stamps[((size_t)CRS_STREAM * 4 + (threadIdx.x >> 6)) * 64 + 61] = w.cyc_compact;
  }
#endif
}

// ---- variant B: LDS-DMA ring.  Tiles go global -> LDS directly (global_load_lds_dwordx4, no VGPR
// staging) into a ring of NS stages, NS-1 tiles ahead of the math.  The DMA writes LDS linearly
// (wave base + lane*16), so the bank swizzle is applied to the per-lane SOURCE address instead;
// readers use the same involution.  Ordering uses counted vmcnt + a raw s_barrier:
//     wait  vmcnt((NS-2)*LOADS)   -> this thread's share of tile i has landed
//     barrier                     -> everyone's share has, and everyone is done reading stage (i-1)%NS
//     issue tile i+NS-1 into stage (i-1)%NS, then do tile i's math
// Past the end the clamped addresses turn the prefetch into harmless re-reads, which keeps the
// vmcnt arithmetic uniform.
template <int D, int TR, int L, int NS, int WGPC>
__global__ __launch_bounds__(kThreads, WGPC) void scan_f16_ring_kernel(const ScanArgs a) {
  using C = Cfg<D, TR, L>;
  constexpr int kWaitN = (NS - 2) * C::kLoads;
  static_assert(kWaitN <= 63, "vmcnt field is 6 bits");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* tile_buf = smem;
  float* sbuf_all = reinterpret_cast<float*>(smem + NS * C::kTileBytes);
  int* ibuf_all = reinterpret_cast<int*>(smem + NS * C::kTileBytes + C::kListBytes);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwg = CRS_NSTREAMS;
  const bool wave_active = (CRS_QBLOCK * 64 + wave * 16) < a.nq;

  // source offset (tile relative) of the 16-byte chunk that lands at LDS position P = j*256 + tid
  unsigned src_off[C::kLoads];
#pragma unroll
  for (int j = 0; j < C::kLoads; ++j) {
    const int P = j * kThreads + tid;
    const int r = P / C::kCpr, cp = P % C::kCpr;
    const int c = (cp & ~15) | ((cp ^ r) & 15);
    src_off[j] = (unsigned)(r * C::kCpr + c) * 16u;
  }
  const char* slab = reinterpret_cast<const char*>(a.slab);
  const size_t last_chunk = (size_t)a.n_rows * (D * 2) - 16;
  const int n_full = a.n_rows / TR;

  auto issue = [&](int tile, int stage) {
    char* sb = tile_buf + stage * C::kTileBytes + wave * 1024;
    if (tile < n_full) {
      const char* base = slab + (size_t)tile * C::kTileBytes;
#pragma unroll
      for (int j = 0; j < C::kLoads; ++j)
        __builtin_amdgcn_global_load_lds((gptr_t)(base + src_off[j]), (lptr_t)(sb + j * (kThreads * 16)), 16, 0, 0);
    } else {
#pragma unroll
      for (int j = 0; j < C::kLoads; ++j) {
        size_t off = (size_t)tile * C::kTileBytes + (size_t)src_off[j];
        off = off > last_chunk ? last_chunk : off;
        __builtin_amdgcn_global_load_lds((gptr_t)(slab + off), (lptr_t)(sb + j * (kThreads * 16)), 16, 0, 0);
      }
    }
  };

  int t = CRS_STREAM;
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) issue(t + s * nwg, s);
  WaveState<D, TR, L> w;
  w.init(a, wave, lane, sbuf_all, ibuf_all);  // its vmcnt(0) also retires the prologue DMAs

  int stage = 0;            // stage holding tile t
  int free_stage = NS - 1;  // stage (i-1) % NS: the one the next issue may overwrite
  for (; t < a.n_tiles; t += nwg) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kWaitN) : "memory");
    __builtin_amdgcn_s_barrier();
    issue(t + (NS - 1) * nwg, free_stage);
    if (wave_active) w.tile(tile_buf + stage * C::kTileBytes, t, a);
    free_stage = stage;
    stage = (stage + 1 == NS) ? 0 : stage + 1;
  }
  // drain the tail prefetches before the workgroup's LDS can be handed to another one
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (wave_active) w.finish(a, wave);
}

template <typename K>
int launch_kernel(K kernel, int lds, const ScanArgs& a, int nwg, hipStream_t stream, bool* attr_done) {
  if (!*attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    *attr_done = true;
  }
  dim3 grid(a.nqb * a.nwg);           // 1-D; (query block, tile stream) from scan_common.h's grid mapping
  hipLaunchKernelGGL(kernel, grid, dim3(kThreads), lds, stream, a);
  return (int)hipGetLastError();
}

template <int D, int TR, int L>
int launch_l(const ScanArgs& a, int nwg, hipStream_t stream) {
  using C = Cfg<D, TR, L>;
  constexpr int lists = 2 * C::kListBytes;
  static bool done[5] = {false, false, false, false, false};
  switch (scan_variant()) {
    case 0: return launch_kernel(&scan_f16_kernel<D, TR, L, false, false>, 2 * C::kTileBytes + lists, a, nwg, stream, &done[0]);
    case 3:
      if (a.boot && L == 16 && TR == 32)
        return launch_kernel(&scan_f16_kernel<D, TR, L, true, (L == 16 && TR == 32)>, 2 * C::kTileBytes + lists, a, nwg, stream, &done[4]);
      // 32-slot lists (k > 16) on rows of >= 512 elements run out of registers and spill a few; an inline-asm
      // load whose destination the compiler then copies or spills before the data has landed would hand
      // garbage on, so those instantiations keep compiler-visible loads
      if constexpr (L == 32 && D >= 512)
        return launch_kernel(&scan_f16_kernel<D, TR, L, false, false>, 2 * C::kTileBytes + lists, a, nwg, stream, &done[0]);
      else
        return launch_kernel(&scan_f16_kernel<D, TR, L, true, false>, 2 * C::kTileBytes + lists, a, nwg, stream, &done[3]);
    case 2: return launch_kernel(&scan_f16_ring_kernel<D, TR, L, 2, 2>, 2 * C::kTileBytes + lists, a, nwg, stream, &done[2]);
    default: {
      constexpr int NS = (L == 16) ? 4 : 3;
      return launch_kernel(&scan_f16_ring_kernel<D, TR, L, NS, 1>, NS * C::kTileBytes + lists, a, nwg, stream, &done[1]);
    }
  }
}

template <int D, int TR>
int launch_d(const ScanArgs& a, int nwg, hipStream_t stream) {
  if (a.k <= 16) return launch_l<D, TR, 16>(a, nwg, stream);
  return launch_l<D, TR, 32>(a, nwg, stream);
}

}  // namespace

// 0 = register-staged double buffer (plain loads), 3 = same with asm early loads (2 workgroups/CU);
// 1 = LDS-DMA ring, 1 workgroup/CU; 2 = LDS-DMA double buffer, 2 workgroups/CU.
// CRS_SCAN_VARIANT overrides the default; read once.
int scan_variant() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("CRS_SCAN_VARIANT");
    v = (e && e[0] >= '0' && e[0] <= '3') ? (e[0] - '0') : 3;
  }
  return v;
}
int scan_wg_per_cu() { return scan_variant() == 1 ? 1 : 2; }
// Cross-workgroup threshold sharing (scan_common.h) is OFF by default: measured on MI355X it cut the
// in-loop compactions per wave from 2 to 1 at C2, but the per-tile L1-bypassing poll of the shared
// word sits in the same in-order vmcnt queue as the tile loads and cost more than it saved
// (C2 28.8 -> 38.0 us, C4 174 -> 257 us).  CRS_SCAN_SHARE_TAU=1 re-enables it for experiments.
bool scan_share_tau() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("CRS_SCAN_SHARE_TAU");
    v = (e && e[0] == '1') ? 1 : 0;
  }
  return v == 1;
}

int scan_tile_rows(int pdim) { return pdim <= 512 ? 32 : 16; }

int scan_launch_f16(const ScanArgs& a, int pdim, int nwg, hipStream_t stream) {
  switch (pdim) {
    case 128: return launch_d<128, 32>(a, nwg, stream);
    case 256: return launch_d<256, 32>(a, nwg, stream);
    case 384: return launch_d<384, 32>(a, nwg, stream);
    case 512: return launch_d<512, 32>(a, nwg, stream);
    case 640: return launch_d<640, 16>(a, nwg, stream);
    case 768: return launch_d<768, 16>(a, nwg, stream);
    case 896: return launch_d<896, 16>(a, nwg, stream);
    case 1024: return launch_d<1024, 16>(a, nwg, stream);
    default: return -1;
  }
}

}  // namespace crs
