// scan_common.h -- device code shared by the fp16 and int8 scan kernels (scan.hip, scan_i8.hip):
// the register sorting networks, the per-wave candidate lists and their compaction.
// Header-only; everything lives in an anonymous namespace of the including translation unit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "scan.h"
#include "lds_dma.h"

// Grid mapping (1-D grid of nqb * nwg workgroups): every tile stream is walked by nqb workgroups, one per
// 64-query block.  Workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share one), and each
// XCD has its own L2: the nqb workgroups of a stream are therefore placed on linear ids b, b+8, b+16, ...
// so that they share an XCD, start together and run in lockstep -- the stream's tiles are then fetched
// from HBM once and re-read by the other query blocks from that XCD's L2 (with the natural order
// b = stream * nqb + qblock the re-reads come from four different XCDs and are served by the Infinity
// Cache at ~8 TB/s, which bounded the 256-query C3 launch).  Placement is only a speed matter.
#define CRS_NSTREAMS (a.nwg)
#define CRS_QBLOCK (crs::grid_qblock(a))
#define CRS_STREAM (crs::grid_stream(a))

namespace crs {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));  // native vector: usable as an asm operand

__device__ __forceinline__ int grid_qblock(const ScanArgs& a) {
  const int b = (int)blockIdx.x;
  if (a.nqb == 1) return 0;
  if (a.nwg & 7) return b % a.nqb;
  return (b >> 3) % a.nqb;
}
__device__ __forceinline__ int grid_stream(const ScanArgs& a) {
  const int b = (int)blockIdx.x;
  if (a.nqb == 1) return b;
  if (a.nwg & 7) return b / a.nqb;
  return ((b >> 3) / a.nqb) * 8 + (b & 7);
}


// In-kernel timeline stamps for tools/scan_probe.hip (a separate diagnostic build); no code in the product build.
#ifdef CRS_STAMPS
#define CRS_STAMP(slot)                                                                          \
  do {                                                                                           \
    if (a.stamps && (threadIdx.x & 63) == 0) {                                                   \
      const unsigned long long t_ = __builtin_amdgcn_s_memtime();                               \
      a.stamps[((size_t)CRS_STREAM * 4 + (threadIdx.x >> 6)) * 64 + (slot)] = t_;                \
    }                                                                                            \
  } while (0)
#define CRS_STAMP_REAL(slot)                                                                     \
  do {                                                                                           \
    if (a.stamps && (threadIdx.x & 63) == 0)                                                     \
      a.stamps[((size_t)CRS_STREAM * 4 + (threadIdx.x >> 6)) * 64 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define CRS_STAMP(slot) do {} while (0)
#define CRS_STAMP_REAL(slot) do {} while (0)

#endif

// tools/scan_wide_probe.hip, tools/scan_tb_probe.hip: per-wave cycle accumulators (diagnostic build only)
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

constexpr int kThreads = 256;
constexpr int kWaves = 4;
constexpr float kNegInf = -__builtin_huge_valf();

// ---------------------------------------------------------------- register sorting networks
template <int L>
__device__ __forceinline__ void bitonic_sort_desc(float (&v)[L]) {
#pragma unroll
  for (int k = 2; k <= L; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
      for (int i = 0; i < L; ++i) {
        const int l = i ^ j;
        if (l > i) {
          const bool desc = ((i & k) == 0);
          const float a = v[i], b = v[l];
          const float hi = fmaxf(a, b), lo = fminf(a, b);
          v[i] = desc ? hi : lo;
          v[l] = desc ? lo : hi;
        }
      }
    }
  }
}

// v is bitonic -> sorted descending
template <int L>
__device__ __forceinline__ void bitonic_clean_desc(float (&v)[L]) {
#pragma unroll
  for (int j = L >> 1; j > 0; j >>= 1) {
#pragma unroll
    for (int i = 0; i < L; ++i) {
      const int l = i ^ j;
      if (l > i) {
        const float a = v[i], b = v[l];
        v[i] = fmaxf(a, b);
        v[l] = fminf(a, b);
      }
    }
  }
}

__device__ __forceinline__ int quad_sum(int x) {
  x += __shfl_xor(x, 16);
  x += __shfl_xor(x, 32);
  return x;
}
__device__ __forceinline__ int quad_min(int x) {
  x = min(x, __shfl_xor(x, 16));
  x = min(x, __shfl_xor(x, 32));
  return x;
}

// m is sorted descending: m[idx] = min over j <= idx (written as a min chain so the compiler keeps
// the array in registers instead of indexing it through scratch)
template <int L>
__device__ __forceinline__ float pick(const float (&m)[L], int idx) {
  float t = m[0];
#pragma unroll
  for (int j = 1; j < L; ++j) t = fminf(t, (j <= idx) ? m[j] : __builtin_huge_valf());
  return t;
}

// The k-th best (1-based k) score among the 4*L candidates of this lane's query, L = 16, k <= 16.
// `s` = this lane's L scores (unsorted, -inf padded); destroyed.
__device__ __forceinline__ float kth_of_quad(float (&s)[16], int k) {
  bitonic_sort_desc<16>(s);
  float m[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) m[j] = fmaxf(s[j], __shfl_xor(s[15 - j], 16));
  bitonic_clean_desc<16>(m);  // top-16 of the lane pair, both partners hold the same list
#pragma unroll
  for (int j = 0; j < 16; ++j) s[j] = fmaxf(m[j], __shfl_xor(m[15 - j], 32));
  bitonic_clean_desc<16>(s);  // top-16 of the quad
  return pick<16>(s, k - 1);
}

// L = 32, k <= 64: the sorted top-64 of the quad's 128 candidates lives in a lane PAIR
// (lane g=0/2 holds ranks 0..31, lane g=1/3 ranks 32..63).
__device__ __forceinline__ float kth_of_quad(float (&s)[32], int k) {
  const int lane = threadIdx.x & 63;
  const bool upper = (lane >> 4) & 1;  // g odd: holds the lower-ranked half of its pair
  bitonic_sort_desc<32>(s);
  float m[32];
  // full merge inside the pair (g, g^1): lower g keeps the 32 largest, upper the 32 smallest
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    const float o = __shfl_xor(s[31 - j], 16);
    m[j] = upper ? fminf(s[j], o) : fmaxf(s[j], o);
  }
  bitonic_clean_desc<32>(m);  // pair now holds 64 sorted: [lower lane | upper lane]
  // top-64 of the two pairs: element i of this pair against element 63-i of the other pair,
  // which sits in lane^48 at index 31-j.
#pragma unroll
  for (int j = 0; j < 32; ++j) s[j] = fmaxf(m[j], __shfl_xor(m[31 - j], 48));
  // (lower | upper) is a bitonic sequence of 64: first the distance-32 exchange across the pair
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    const float o = __shfl_xor(s[j], 16);
    m[j] = upper ? fminf(s[j], o) : fmaxf(s[j], o);
  }
  bitonic_clean_desc<32>(m);
  const float t = pick<32>(m, (k - 1) & 31);
  // rank k-1 sits in pair-lane ((k-1) >> 5)
  const int src = (lane & 47 & ~16) | (((k - 1) >> 5) << 4);
  return __shfl(t, src);
}

// max over the lane pair (l, l ^ 32) / the lane quad (l, l ^ 16, l ^ 32, l ^ 48) with the gfx950 row
// swaps (VALU, no LDS round trip: a ds_bpermute queues behind the tile reads of every wave of the CU).
// v_permlane32_swap(D, S) exchanges lanes 32..63 of D with lanes 0..31 of S; fed the same register
// twice it returns (low half broadcast, high half broadcast), whose max is the pair's max in all lanes.
__device__ __forceinline__ float pair_max(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float quad_max(float x) {   // v_permlane16_swap: odd 16-lane rows of D <-> even rows of S
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return pair_max(fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1])));
}

// ---------------------------------------------------------------- threshold sharing across workgroups
// Every wave's tau is the k-th best of a SUBSET of the rows, hence a lower bound on the global k-th
// best.  Waves publish it with an agent-scope atomic max on an order-preserving integer image of
// the float; readers (write-through / L1-bypassing loads, one tile late, staleness is harmless) may
// then drop anything strictly below it.  A foreign tau is applied one ulp lower, so a row that
// TIES the foreign k-th score still passes (it may precede that entry in row order).
__device__ __forceinline__ unsigned ord_of(float x) {
  const unsigned u = __float_as_uint(x);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float float_of_ord(unsigned o) {
  return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}
__device__ __forceinline__ float foreign_tau(unsigned o) {   // strict threshold equivalent to ">= published"
  return o > 1u ? float_of_ord(o - 1u) : kNegInf;
}

// ---------------------------------------------------------------- per-wave candidate lists
// sbuf/ibuf: [L][64] (slot-major, lane-minor => conflict-free 4-byte accesses).
// On exit (not FINAL) every query keeps exactly min(k, total) candidates, spread round-robin over
// its 4 lanes, and tau is the k-th best seen so far.  FINAL writes them to out_s/out_i instead
// (k slots per query, unused slots get (-inf, -1)); out pointers are per-lane (this lane's query).
template <int L, bool FINAL>
__device__ __forceinline__ void compact(float* __restrict__ sbuf, int* __restrict__ ibuf, int lane,
                                        int& cnt, float& tau, int k, float* out_s, int* out_i,
                                        bool q_valid, int kp = 0, unsigned* tau_pub = nullptr) {
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
  const int total = quad_sum(cnt);
  float tnew = kth_of_quad(s, k);  // -inf when total < k (padding)
  int n_gt = 0, n_eq = 0;
#pragma unroll
  for (int j = 0; j < L; ++j) {
    n_gt += (j < cnt && v[j] > tnew) ? 1 : 0;
    n_eq += (j < cnt && v[j] == tnew) ? 1 : 0;
  }
  n_gt = quad_sum(n_gt);
  n_eq = quad_sum(n_eq);
  // ties on the threshold score: keep only the `need` smallest rows among the equal ones
  int need = (total >= k) ? (k - n_gt) : n_eq;
  int idthr = 0x7fffffff;
  if (__any(n_eq > need)) {
    int thr = -1;
    const int rounds = (n_eq > need) ? need : 0;
    for (int it = 0; __any(it < rounds); ++it) {
      int c = 0x7fffffff;
#pragma unroll
      for (int j = 0; j < L; ++j)
        if (j < cnt && v[j] == tnew && id[j] > thr) c = min(c, id[j]);
      c = quad_min(c);
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
  const int qb = lane & 15;
  const int c0 = __shfl(c, qb), c1 = __shfl(c, qb + 16), c2 = __shfl(c, qb + 32),
            c3 = __shfl(c, qb + 48);
  const int g = lane >> 4;
  const int prefix = (g > 0 ? c0 : 0) + (g > 1 ? c1 : 0) + (g > 2 ? c2 : 0);
  const int kept = c0 + c1 + c2 + c3;
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
      for (int p = kept + g; p < kp; p += 4) {
        out_s[p] = kNegInf;
        out_i[p] = -1;
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < L; ++j) {
      if ((km >> j) & 1u) {
        const int p = prefix + __popc(km & ((1u << j) - 1u));
        const int dl = qb + ((p & 3) << 4);
        sbuf[(p >> 2) * 64 + dl] = v[j];
        ibuf[(p >> 2) * 64 + dl] = id[j];
      }
    }
    cnt = (kept - g + 3) >> 2;
    if (total >= k) {
      tau = fmaxf(tau, tnew);
      if (tau_pub && g == 0 && q_valid)
        __hip_atomic_fetch_max(tau_pub, ord_of(tnew), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// End of a wave's stream: hand its candidates to the merge kernel.  When every query of the wave
// holds <= kp candidates they are dumped as they are (a superset of the wave's top-k: nothing was
// ever dropped that could rank), skipping the final selection; otherwise one last compaction.
template <int L>
__device__ __forceinline__ void flush_lists(float* __restrict__ sbuf, int* __restrict__ ibuf, int lane, int& cnt,
                                            float& tau, int k, int kp, float* out_s, int* out_i, bool q_valid) {
  const int total = quad_sum(cnt);
  if (__any(total > kp)) {
    compact<L, true>(sbuf, ibuf, lane, cnt, tau, k, out_s, out_i, q_valid, kp);
    return;
  }
  const int qb = lane & 15, g = lane >> 4;
  const int c0 = __shfl(cnt, qb), c1 = __shfl(cnt, qb + 16), c2 = __shfl(cnt, qb + 32);
  const int prefix = (g > 0 ? c0 : 0) + (g > 1 ? c1 : 0) + (g > 2 ? c2 : 0);
  if (!q_valid) return;
#pragma unroll
  for (int j = 0; j < L; ++j) {
    if (j < cnt) {
      out_s[prefix + j] = sbuf[j * 64 + lane];
      out_i[prefix + j] = ibuf[j * 64 + lane];
    }
  }
  for (int p = total + g; p < kp; p += 4) {
    out_s[p] = kNegInf;
    out_i[p] = -1;
  }
}

}  // namespace
}  // namespace crs
