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
//   * result layout: lane l holds query (l & 31), rows (reg & 3) + 8 (reg >> 2) + 4 (l >> 5) of the tile,
//     so a query's candidates live in a lane PAIR (l, l ^ 32): private LDS lists of 16 slots per lane,
//     compaction = one in-register sorting network + ONE cross-lane merge (scan_common.h's quad version
//     needs two);
//   * tile staging, XOR swizzle (conflict-free for the 32-row fragment reads too: every 16-lane group of
//     a ds_read_b128 covers 16 distinct rows mod 16), early asm loads and the one-barrier double buffer
//     are those of scan.hip.
// k <= 16 only (the launcher falls back to scan.hip otherwise).  Exactness argument unchanged: rows are
// visited in ascending order inside a stream, the filter is a strict compare against the running k-th
// best, ties on the threshold are resolved by row inside the compaction.

#include "scan_common.h"

#include <stdlib.h>

namespace crs {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// tools/scan_wide_probe.hip: per-wave cycle accumulators (diagnostic build only)
#ifdef CRS_STAMPS
#define WP_DECL unsigned long long wp_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long wp_t_ = __builtin_amdgcn_s_memtime(); const unsigned long long wp_t0_ = wp_t_
#define WP_LAP(slot) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); wp_[slot] += n_ - wp_t_; wp_t_ = n_; } while (0)
#define WP_COUNT(slot) do { ++wp_[slot]; } while (0)
#define WP_STORE(nw_) do { if (a.stamps && lane == 0) { wp_[11] = __builtin_amdgcn_s_memtime() - wp_t0_; for (int i_ = 0; i_ < 12; ++i_) a.stamps[((size_t)blockIdx.x * (nw_) + wave) * 12 + i_] = wp_[i_]; } } while (0)
#else
#define WP_DECL do {} while (0)
#define WP_LAP(slot) do {} while (0)
#define WP_COUNT(slot) do {} while (0)
#define WP_STORE(nw_) do {} while (0)
#endif

constexpr int WTR = 32;   // tile rows = one 32x32 MFMA row block
constexpr int WL = 16;    // list slots per lane

template <int D, int NW>
struct WCfg {
  static constexpr int kThreadsW = NW * 64;
  static constexpr int kCpr = D / 8;
  static constexpr int kTileBytes = WTR * D * 2;
  static constexpr int kLoads = kTileBytes / (kThreadsW * 16);
  static constexpr int kKsteps = D / 16;
  static constexpr int kListBytes = NW * WL * 64 * 4;
  static constexpr int kLds = 2 * kTileBytes + 2 * kListBytes;
  static_assert(D % 128 == 0, "row length must be a multiple of 128 elements");
  static_assert(kTileBytes % (kThreadsW * 16) == 0, "tile must split into whole 16-byte loads");
};

__device__ __forceinline__ int pair_sum(int x) { return x + __shfl_xor(x, 32); }
__device__ __forceinline__ int pair_min(int x) { return min(x, __shfl_xor(x, 32)); }

// k-th best (1-based) of the 32 candidates of this lane's query (16 here, 16 in lane ^ 32)
__device__ __forceinline__ float kth_of_pair(float (&s)[16], int k) {
  bitonic_sort_desc<16>(s);
  float m[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) m[j] = fmaxf(s[j], __shfl_xor(s[15 - j], 32));
  bitonic_clean_desc<16>(m);
  return pick<16>(m, k - 1);
}

// As scan_common.h's compact(), for a query spread over the lane pair (l, l ^ 32).
template <bool FINAL>
__device__ __forceinline__ void compact2(float* __restrict__ sbuf, int* __restrict__ ibuf, int lane, int& cnt,
                                         float& tau, int k, float* out_s, int* out_i, bool q_valid, int kp) {
  constexpr int L = WL;
  float v[L], s[L];
  int id[L];
#pragma unroll
  for (int j = 0; j < L; ++j) {
    const bool in = j < cnt;
    const float x = sbuf[j * 64 + lane];
    v[j] = in ? x : kNegInf;
    s[j] = v[j];
    id[j] = ibuf[j * 64 + lane];
  }
  const int total = pair_sum(cnt);
  const float tnew = kth_of_pair(s, k);   // -inf while the pair holds fewer than k candidates
  int n_gt = 0, n_eq = 0;
#pragma unroll
  for (int j = 0; j < L; ++j) {
    n_gt += (j < cnt && v[j] > tnew) ? 1 : 0;
    n_eq += (j < cnt && v[j] == tnew) ? 1 : 0;
  }
  n_gt = pair_sum(n_gt);
  n_eq = pair_sum(n_eq);
  const int need = (total >= k) ? (k - n_gt) : n_eq;
  int idthr = 0x7fffffff;
  if (__any(n_eq > need)) {   // ties on the threshold score: keep the `need` smallest rows among them
    int thr = -1;
    const int rounds = (n_eq > need) ? need : 0;
    for (int it = 0; __any(it < rounds); ++it) {
      int c = 0x7fffffff;
#pragma unroll
      for (int j = 0; j < L; ++j)
        if (j < cnt && v[j] == tnew && id[j] > thr) c = min(c, id[j]);
      c = pair_min(c);
      if (it < rounds) thr = c;
    }
    if (n_eq > need) idthr = thr;
  }
  unsigned km = 0;
#pragma unroll
  for (int j = 0; j < L; ++j) {
    const bool keep = (j < cnt) && (v[j] > tnew || (v[j] == tnew && id[j] <= idthr));
    km |= (keep ? 1u : 0u) << j;
  }
  const int c = __popc(km);
  const int qb = lane & 31, g = lane >> 5;
  const int c0 = __shfl(c, qb), c1 = __shfl(c, qb + 32);
  const int prefix = g ? c0 : 0;
  const int kept = c0 + c1;
  if (FINAL) {
    if (q_valid) {
#pragma unroll
      for (int j = 0; j < L; ++j) {
        if ((km >> j) & 1u) {
          const int p = prefix + __popc(km & ((1u << j) - 1u));
          out_s[p] = v[j];
          out_i[p] = id[j];
        }
      }
      for (int p = kept + g; p < kp; p += 2) {
        out_s[p] = kNegInf;
        out_i[p] = -1;
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < L; ++j) {
      if ((km >> j) & 1u) {
        const int p = prefix + __popc(km & ((1u << j) - 1u));
        const int dl = qb + ((p & 1) << 5);
        sbuf[(p >> 1) * 64 + dl] = v[j];
        ibuf[(p >> 1) * 64 + dl] = id[j];
      }
    }
    cnt = (kept - g + 1) >> 1;
    if (total >= k) tau = fmaxf(tau, tnew);
  }
}

// PF = tiles in flight per workgroup: 2 for the 8-wave configuration (one workgroup per CU) where the
// registers allow it, else 1 (two workgroups per CU, or D = 512 whose query fragments fill the file).
template <int D, int NW>
constexpr int wide_pf() { return 1; }   // 2 measured no faster (the kernel was barrier-bound, below) and doubles the code

template <int D, int NW>
__global__ __launch_bounds__(NW * 64, 2) void scan_wide_kernel(const ScanArgs a) {
  using C = WCfg<D, NW>;
  constexpr int PF = wide_pf<D, NW>();
  constexpr int kT = C::kThreadsW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* tile_buf = smem;
  float* sbuf_all = reinterpret_cast<float*>(smem + 2 * C::kTileBytes);
  int* ibuf_all = reinterpret_cast<int*>(smem + 2 * C::kTileBytes + C::kListBytes);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwg = CRS_NSTREAMS;
  const int qblock = CRS_QBLOCK, stream = CRS_STREAM;
  WP_DECL;
  const bool wave_active = (qblock * (NW * 32) + wave * 32) < a.nq;   // wave-uniform

  // staging geometry: load j of this thread covers 16-byte chunk P = j * kT + tid of the tile
  int lds_dst[C::kLoads];
#pragma unroll
  for (int j = 0; j < C::kLoads; ++j) {
    const int P = j * kT + tid;
    const int r = P / C::kCpr, c = P % C::kCpr;
    lds_dst[j] = (r * C::kCpr + ((c & ~15) | ((c ^ r) & 15))) * 16;
  }
  const char* slab = reinterpret_cast<const char*>(a.slab);
  const size_t last_chunk = (size_t)a.n_rows * (D * 2) - 16;
  const int n_full = a.n_rows / WTR;

  // Two register staging sets: while tile i is being multiplied out of LDS, tiles i+1 AND i+2 are in
  // flight (one tile ahead leaves 24 KB per CU in flight at one workgroup per CU, and the kernel then
  // runs at the memory latency: measured 2.4 TB/s on C4 with 256 queries).
  u32x4 st0[C::kLoads], st1[PF == 2 ? C::kLoads : 1];
  auto load_tile = [&](auto& st, int tile_) {
    const int tile = __builtin_amdgcn_readfirstlane(tile_);
    if (tile < n_full) {
      const char* base = uniform_ptr(slab + (size_t)tile * C::kTileBytes);
#pragma unroll
      for (int j = 0; j < C::kLoads; ++j) {
        const unsigned off = (unsigned)(j * kT + tid) * 16u;
        u32x4 x;
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(x) : "v"(off), "s"(base) : "memory");
        st[j] = x;
      }
    } else {   // ragged last tile, or past the end: clamp every lane to the slab's last 16 bytes
#pragma unroll
      for (int j = 0; j < C::kLoads; ++j) {
        size_t off = (size_t)tile * C::kTileBytes + (size_t)(j * kT + tid) * 16;
        off = off > last_chunk ? last_chunk : off;
        const char* p = slab + off;
        u32x4 x;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(x) : "v"(p) : "memory");
        st[j] = x;
      }
    }
  };
  // wait until all but the youngest tile's loads have landed (vmcnt counts in issue order), then move
  // this set into LDS
  auto park_tile = [&](auto& st, char* dst) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PF == 2 ? C::kLoads : 0) : "memory");
#pragma unroll
    for (int j = 0; j < C::kLoads; ++j) {
      u32x4 x = st[j];
      asm volatile("" : "+v"(x));
      st[j] = x;
    }
#pragma unroll
    for (int j = 0; j < C::kLoads; ++j) *reinterpret_cast<u32x4*>(dst + lds_dst[j]) = st[j];
  };

  int t = stream;
  load_tile(st0, t);   // goes out before the query fragments are fetched, so the two latencies overlap

  // ---- this wave's 32 queries, full depth, as B fragments: lane (n = l & 31, h = l >> 5) holds
  // Q[n][16 ks + 8 h .. + 8] for every k-step
  const int qn = lane & 31, h = lane >> 5;
  const int qi = qblock * (NW * 32) + wave * 32 + qn;
  const bool q_valid = qi < a.nq;
  f16x8 qf[C::kKsteps];
  {
    const _Float16* qrow = a.q + (size_t)(q_valid ? qi : 0) * D + h * 8;
#pragma unroll
    for (int ks = 0; ks < C::kKsteps; ++ks) {
      const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      qf[ks] = q_valid ? *reinterpret_cast<const f16x8*>(qrow + ks * 16) : z;
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

  float* sbuf = sbuf_all + wave * (WL * 64);
  int* ibuf = ibuf_all + wave * (WL * 64);
  float tau = q_valid ? kNegInf : __builtin_huge_valf();
  int cnt = 0;

  if constexpr (PF == 2) load_tile(st1, t + nwg);   // after the query loads: park_tile's counted wait then covers st0 + queries
  park_tile(st0, tile_buf);
  __syncthreads();
  WP_LAP(0);   // prologue

  int cur = 0, it = 0;
  // one iteration: tile t sits in LDS buffer `cur`, tile t + nwg is in flight in `sx`, `sy` is free
  auto body = [&](auto& sx, auto& sy) {
    if constexpr (PF == 2) load_tile(sy, t + 2 * nwg); else load_tile(sx, t + nwg);
    if (wave_active) {
      WP_LAP(1);   // tile-load issue
      WP_LAP(2);
      const char* buf = tile_buf + cur * C::kTileBytes;
      f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < C::kKsteps; ++ks) {
        const f16x8 af = *reinterpret_cast<const f16x8*>(buf + a_off[ks & 7] + (ks >> 3) * 256);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, qf[ks], acc, 0, 0, 0);
      }
      WP_LAP(3);   // MFMA sweep
      bool hit = false;
#pragma unroll
      for (int r = 0; r < 16; ++r) hit |= acc[r] > tau;
      if (__any(hit)) {
        const int row_base = t * WTR + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float sc = acc[4 * g + i];
            const int row = row_base + 8 * g + i;
            if (sc > tau && row < a.n_rows) {
              sbuf[cnt * 64 + lane] = sc;
              ibuf[cnt * 64 + lane] = row;
              ++cnt;
            }
          }
          if (__any(cnt > WL - 4)) {
            WP_LAP(4);
            compact2<false>(sbuf, ibuf, lane, cnt, tau, a.k, nullptr, nullptr, q_valid, 0);
            WP_COUNT(9);
            WP_LAP(5);   // on-demand compaction
          }
        }
      }
    }
    WP_LAP(4);   // filter + append
    park_tile(sx, tile_buf + (cur ^ 1) * C::kTileBytes);
    WP_LAP(6);   // wait for the next tile + LDS store
    __syncthreads();
    WP_LAP(7);   // barrier
    cur ^= 1;
    ++it;
    t += nwg;
  };
  if constexpr (PF == 2) {
    while (t < a.n_tiles) {
      body(st1, st0);
      if (t >= a.n_tiles) break;
      body(st0, st1);
    }
  } else {
    while (t < a.n_tiles) body(st0, st0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tail prefetches (clamped re-reads) must not outlive the kernel's registers

  if (wave_active) {
    const size_t o = ((size_t)(q_valid ? qi : 0) * nwg + stream) * a.kp;   // [nq, nwg, kp]
    float* out_s = a.part_scores + o;
    int* out_i = a.part_rows + o;
    const int total = pair_sum(cnt);
    if (__any(total > a.kp)) {
      compact2<true>(sbuf, ibuf, lane, cnt, tau, a.k, out_s, out_i, q_valid, a.kp);
    } else {   // every query holds <= kp candidates: a superset of its top-k, dump as is
      const int c0 = __shfl(cnt, qn);
      const int prefix = h ? c0 : 0;
      if (q_valid) {
#pragma unroll
        for (int j = 0; j < WL; ++j) {
          if (j < cnt) {
            out_s[prefix + j] = sbuf[j * 64 + lane];
            out_i[prefix + j] = ibuf[j * 64 + lane];
          }
        }
        for (int p = total + h; p < a.kp; p += 2) {
          out_s[p] = kNegInf;
          out_i[p] = -1;
        }
      }
    }
  }
  WP_LAP(10);   // final flush
  WP_STORE(NW);
}

template <int D, int NW>
int launch_wide(const ScanArgs& a, hipStream_t stream) {
  using C = WCfg<D, NW>;
  static bool done = false;
  auto kernel = &scan_wide_kernel<D, NW>;
  if (!done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, C::kLds);
    if (e != hipSuccess) return (int)e;
    done = true;
  }
  hipLaunchKernelGGL(kernel, dim3(a.nqb * a.nwg), dim3(NW * 64), C::kLds, stream, a);
  return (int)hipGetLastError();
}

template <int D>
int launch_wide_d(const ScanArgs& a, int nw, hipStream_t stream) {
  if constexpr (D <= 384) {
    if (nw == 4) return launch_wide<D, 4>(a, stream);
  }
  return nw == 8 ? launch_wide<D, 8>(a, stream) : -1;
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
  return (nq > 128 || pdim > 384) ? 8 : 4;   // <512, 4 waves> would spill
}
// resident workgroups per CU: 8 waves = 112..128 KB of LDS -> 1; 4 waves (<= 96 KB) -> 1 or 2
int scan_wide_wg_per_cu(int nw, int pdim) {
  if (nw == 8) return 1;
  const int lds = 2 * WTR * pdim * 2 + 2 * nw * WL * 64 * 4;
  return lds <= 80 * 1024 ? 2 : 1;
}

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
