// enc_gemm8.hip -- 256 x 256 x 64 MFMA GEMM with an eight-barrier-per-k-tile phase schedule (gfx950).
//
//   C[M, N] = epilogue( A[M, K] x W[N, K]^T + bias[N] )        mode 0: fp16; 1: erf-GELU, fp16; 2: + residual fp32, fp32
//
// The structure the CDNA4 guide measures at ~1.3 PFLOP/s on random fp16/bf16 data (cdna_hip_programming.md section 5,
// "256^2 8-phase template"), written for this library's operand layout (W stored [N, K] like torch.nn.Linear):
//   * one 512-thread workgroup per CU owns a 256 x 256 output tile; eight waves as 2 (rows) x 4 (columns), wave tile
//     128 x 64 = 8 x 4 accumulator tiles of v_mfma_f32_16x16x32_f16 (the shape the chip holds the higher clock on);
//   * LDS: two k-tile buffers of four 16 KB HALF tiles (A rows 0-127 | A rows 128-255 | W rows 0-127 | W rows 128-255,
//     128-byte rows, 16-byte chunks XOR-swizzled by (row >> 1) & 7: conflict-free ds_read_b128), filled by LDS-DMA
//     (global_load_lds_dwordx4; the swizzle is applied to the SOURCE address, the transfer writes LDS linearly);
//   * a k-tile is four PHASES of 16 MFMAs (one 64 x 32 quadrant of the wave tile x K = 64), each between two barriers;
//     the two waves of a SIMD (wave w and w + 4 = the two row groups) run half a phase apart, so one multiplies while
//     the other reads fragments and issues transfers;
//   * every phase issues ONE half tile, three to six phases ahead of its first use, behind a counted vmcnt(6) once per
//     k-tile (three half tiles stay in flight across the barriers; the loop never drains to vmcnt(0)).
// Hazards (why this order is safe; barrier numbers per k-tile: group 0 passes X_p = 2p, Y_p = 2p + 1, group 1 X_p = 2p + 1,
// Y_p = 2p + 2):
//   RAW  a half tile is read one phase after the barrier that follows EVERY wave's vmcnt for it: the wait sits before X_3
//        of k-tile t (numbers 6 / 7), the first reads of k-tile t + 1 come after Y_3 (7 / 8);
//   WAR  fragment reads are retired (lgkmcnt(0)) BEFORE the wave's X barrier, and a half tile is re-filled only by waves
//        that have passed a barrier after that X: W halves (read in phase 0) from phase 1 on, the A half of group 0
//        (last read phase 2, retired before number 4) in phase 3 (after 5 / 6), the A half of group 1 (retired before 5)
//        in phase 0 of the next k-tile (after 7 / 8).
// The epilogue goes through small wave-private LDS staging tiles (16 rows of 144 bytes) so that global stores are 16 bytes
// per lane and 128 bytes contiguous per row.
#include "enc.h"
#include "enc_gelu.h"
#include "lds_dma.h"

#include <stdlib.h>

namespace crs {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int PM = 256, PN = 256, PK = 64;
constexpr int kP8Threads = 512;
constexpr int kHalf = 128 * PK * 2;            // one half tile: 128 rows x 128 bytes = 16 KB
constexpr int kBuf = 4 * kHalf;                // one k-tile: [A0 | A1 | W0 | W1] = 64 KB
constexpr int kLdsMain = 2 * kBuf;             // 128 KB
constexpr int kEpiRow16 = 144;                 // epilogue staging rows: 128 bytes of payload, 16-byte aligned, bank-spread
constexpr int kEpiStage = 16 * kEpiRow16;      // one wave's staging tile: 16 rows (2304 bytes)
constexpr int kLds8 = kLdsMain + 8 * kEpiStage;   // 146 KB: the staging tiles sit BEHIND the k-tile buffers

template <int N>
__device__ __forceinline__ void wait_vm8() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_lds8() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// PERSIST: one workgroup per CU walks work items w, w + G, w + 2G, ... (G = gridDim.x) as ONE continuous stream of k-tiles:
// the half tiles of the next item's first k-tiles are issued by the phases of the current item's last k-tiles, exactly as
// inside an item, so only a workgroup's FIRST item pays the fill of the pipeline; the epilogue of an item goes through a
// small wave-private staging tile OUTSIDE the two k-tile buffers (those are already receiving the next item), and because the
// row groups run a barrier apart, one group's epilogue sits beside the other group's MFMA phases.  PERSIST = false launches
// one workgroup per item (same code, has_next = false): items <= CUs, and the A/B reference.
// A work item = (output tile, K slab): MODE 3 (split-K) items contract columns [s Ksplit, (s + 1) Ksplit) and leave their fp32
// partial tile in slab s of out[splits][M][N]; bias / residual / LayerNorm belong to the kernel that sums the slabs.
// NOSTAGGER / NOPRIO: A/B switches (CRS_GEMM8_VAR = 3 / 2): the half-phase stagger of the row groups is worth 12-20 %, the
// s_setprio pair around the MFMA clusters 12-17 % (it keeps hipcc from moving MFMAs across the barriers).
template <int MODE, bool PERSIST, int VAR>
__global__ __launch_bounds__(kP8Threads, 2) void gemm8_kernel(const _Float16* __restrict__ A, const _Float16* __restrict__ W,
                                                             const float* __restrict__ bias, const float* __restrict__ residual,
                                                             void* __restrict__ out, int M, int N, int K, int Ksplit, int n_items) {
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  extern __shared__ __attribute__((aligned(16))) char sm8[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int ncb = (N + PN - 1) / PN;     // N a multiple of 128: the last column block may be half a tile (round 3)
  const int n_tiles = ncb * (M / PM);
  const int G = (int)gridDim.x;
  // item order: column blocks of a row block are neighbours, and each XCD (hardware id % 8) walks one contiguous range of
  // every round's items, so they share the A panel through ONE L2
  const int me = xcd_chunked_id((int)blockIdx.x, G);

  // ---- transfers.  LDS position P = j * 512 + tid (16-byte units) of a half tile = (row P >> 3, slot P & 7) receives
  // source chunk slot ^ ((row >> 1) & 7) of that row; the same (row, chunk) pattern serves all four half tiles.
  unsigned voff[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int id = j * kP8Threads + tid, row = id >> 3, slot = id & 7;
    voff[j] = (unsigned)row * (unsigned)K * 2u + (unsigned)((slot ^ ((row >> 1) & 7)) << 4);
  }
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr_t)sm8 + (unsigned)wave * 1024u);
  const int nk = ((MODE == 3) ? Ksplit : K) / PK;   // k-tiles per item; even (a multiple of 128 columns: the dispatch checks)
  // bases of the four half tiles of an item: [0] / [1] = A rows 0-127 / 128-255, [2] / [3] = W rows 0-127 / 128-255
  struct Bases { const _Float16* p[4]; int m0, n0, slab; };
  auto bases_of = [&](int item) {
    Bases bs;
    const int tile = item % n_tiles;
    bs.slab = item / n_tiles;
    bs.m0 = (tile / ncb) * PM;
    bs.n0 = (tile % ncb) * PN;
    const size_t kb = (MODE == 3) ? (size_t)bs.slab * Ksplit : 0;
    bs.p[0] = uniform_ptr(A + (size_t)bs.m0 * K + kb);
    bs.p[1] = uniform_ptr(A + (size_t)(bs.m0 + 128) * K + kb);
    bs.p[2] = uniform_ptr(W + (size_t)bs.n0 * K + kb);
    // (half a column block: W rows n0 + 128 .. do not exist -- the second half tile re-reads the first; the waves that would
    // multiply it skip their MFMAs and stores)
    bs.p[3] = (bs.n0 + 128 < N) ? uniform_ptr(W + (size_t)(bs.n0 + 128) * K + kb) : bs.p[2];
    return bs;
  };
  int item = me;
  Bases cur = bases_of(item < n_items ? item : 0), nxt = cur;
  bool wv = cur.n0 + wn * 64 < N;      // this wave's 64 output columns exist (wave-uniform)
  bool has_next = PERSIST && (item + G < n_items);
  if (has_next) nxt = bases_of(item + G);
  // k-tile kt of the current item (kt < nk) or k-tile kt - nk of the next one
  auto issue = [&](int buf, int half, int kt) {
    const _Float16* base = (kt < nk ? cur.p[half] + (size_t)kt * PK : nxt.p[half] + (size_t)(kt - nk) * PK);
    const unsigned d = lds0 + (unsigned)(buf * kBuf + half * kHalf);
    lds_dma16(d, voff[0], base);
    lds_dma16(d + 8192u, voff[1], base);
  };

  // ---- fragment addresses: lane (r = lane & 15, h = lane >> 4) reads row r of a 16-row tile, 16-byte chunk 4 s + h
  const int r = lane & 15, h = lane >> 4;
  const int sw = (r >> 1) & 7;
  const int lo0 = r * 128 + ((h ^ sw) << 4);            // k sub-step 0
  const int lo1 = r * 128 + (((4 + h) ^ sw) << 4);      // k sub-step 1
  const char* a_base = sm8 + wm * kHalf;                                            // this row group's A half
  const char* w_base = sm8 + (2 + (wn >> 1)) * kHalf + (wn & 1) * 64 * 128;         // this wave's 64 W rows
  char* stage = sm8 + kLdsMain + wave * kEpiStage;                                  // this wave's epilogue staging tile

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f16x8 af[4][2], bf[4][2];
  if (item >= n_items) return;   // (grid <= items: never taken; keeps a mis-sized launch from reading past the operands)

  // fill: k-tile 0 whole, k-tile 1 up to its A half 0 (phase 0 of the loop issues A half 1 of the following k-tile)
  issue(0, 2, 0); issue(0, 3, 0); issue(0, 0, 0); issue(0, 1, 0);
  issue(1, 2, 1); issue(1, 3, 1); issue(1, 0, 1);
  wait_vm8<6>();
  __builtin_amdgcn_s_barrier();
  if (VAR != 3 && wm == 1) __builtin_amdgcn_s_barrier();   // the row groups run one barrier apart from here on

#define CRS_MFMA_QUAD(RT0, CT0)                                                                                         \
  if (VAR != 2) __builtin_amdgcn_s_setprio(1);                                                                          \
  if (wv)                                                                                                               \
  _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)                                                                      \
  _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                                      \
  _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                                                      \
      acc[(RT0) + i_][(CT0) + j_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[(CT0) + j_][s_], af[i_][s_], acc[(RT0) + i_][(CT0) + j_], 0, 0, 0); \
  if (VAR != 2) __builtin_amdgcn_s_setprio(0);

  auto ktile = [&](int t, const int b) {
    const char* ab = a_base + b * kBuf;
    const char* wb = w_base + b * kBuf;
    const bool more1 = (t + 1 < nk) || has_next, more2 = (t + 2 < nk) || has_next;
    // ---- phase 0: A rows 0-63 of the wave tile, all of its W rows; quadrant (0, 0)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      af[i][0] = *reinterpret_cast<const f16x8*>(ab + i * 2048 + lo0);
      af[i][1] = *reinterpret_cast<const f16x8*>(ab + i * 2048 + lo1);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bf[j][0] = *reinterpret_cast<const f16x8*>(wb + j * 2048 + lo0);
      bf[j][1] = *reinterpret_cast<const f16x8*>(wb + j * 2048 + lo1);
    }
    if (more1) issue(b ^ 1, 1, t + 1);
    wait_lds8();
    __builtin_amdgcn_s_barrier();
    CRS_MFMA_QUAD(0, 0)
    __builtin_amdgcn_s_barrier();
    // ---- phase 1: quadrant (0, 1)
    if (more2) issue(b, 2, t + 2);
    __builtin_amdgcn_s_barrier();
    CRS_MFMA_QUAD(0, 2)
    __builtin_amdgcn_s_barrier();
    // ---- phase 2: A rows 64-127; quadrant (1, 1)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      af[i][0] = *reinterpret_cast<const f16x8*>(ab + (4 + i) * 2048 + lo0);
      af[i][1] = *reinterpret_cast<const f16x8*>(ab + (4 + i) * 2048 + lo1);
    }
    if (more2) issue(b, 3, t + 2);
    wait_lds8();
    __builtin_amdgcn_s_barrier();
    CRS_MFMA_QUAD(4, 2)
    __builtin_amdgcn_s_barrier();
    // ---- phase 3: quadrant (1, 0); the next k-tile must have landed (the three younger half tiles may stay in flight)
    if (more2) { issue(b, 0, t + 2); wait_vm8<6>(); } else { wait_vm8<0>(); }
    __builtin_amdgcn_s_barrier();
    CRS_MFMA_QUAD(4, 0)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- epilogue of one item.  Lane (c = lane & 15, q = lane >> 4) holds, per accumulator tile (rt, ct), token row rt * 16 + c
  // and the four consecutive output columns ct * 16 + 4 q .. + 3 (the product is formed transposed for exactly this).  One
  // 16-row block of the wave tile at a time goes through the wave's staging tile (144-byte rows) and leaves as 16-byte
  // stores, 128 bytes contiguous per row.  No workgroup barrier: the staging tiles are wave-private.
  const int c = lane & 15, q = lane >> 4;
  auto epilogue = [&](const Bases& bs) {
    if (!wv) return;          // (its accumulators were never touched: still zero)
    f32x4 bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      bv[j] = (bias && MODE != 3) ? *reinterpret_cast<const f32x4*>(bias + bs.n0 + wn * 64 + j * 16 + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    const int row_g0 = bs.m0 + wm * 128, col_g0 = bs.n0 + wn * 64;
    if (MODE < 2) {
      _Float16* o = reinterpret_cast<_Float16*>(out);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 v = acc[i][j] + bv[j];
          gelu_f32x2 x0 = {v[0], v[1]}, x1 = {v[2], v[3]};
          if (MODE == 1) { x0 = gelu_erf2(x0); x1 = gelu_erf2(x1); }
          const f16x4 hv = {(_Float16)x0[0], (_Float16)x0[1], (_Float16)x1[0], (_Float16)x1[1]};
          *reinterpret_cast<f16x4*>(stage + c * kEpiRow16 + (j * 16 + 4 * q) * 2) = hv;
          acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int lrow = it * 8 + (lane >> 3), ch = lane & 7;
          const f16x8 v = *reinterpret_cast<const f16x8*>(stage + lrow * kEpiRow16 + ch * 16);
          *reinterpret_cast<f16x8*>(o + (size_t)(row_g0 + i * 16 + lrow) * N + col_g0 + ch * 8) = v;
        }
        __builtin_amdgcn_wave_barrier();
      }
    } else {
      float* o = reinterpret_cast<float*>(out) + (MODE == 3 ? (size_t)bs.slab * M * N : (size_t)0);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {        // 16 rows x 32 fp32 columns (128-byte rows) per pass
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            *reinterpret_cast<f32x4*>(stage + c * kEpiRow16 + (jj * 16 + 4 * q) * 4) = acc[i][jp * 2 + jj] + bv[jp * 2 + jj];
            acc[i][jp * 2 + jj] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int it = 0; it < 2; ++it) {
            const int lrow = it * 8 + (lane >> 3), ch = lane & 7;
            f32x4 v = *reinterpret_cast<const f32x4*>(stage + lrow * kEpiRow16 + ch * 16);
            const size_t at = (size_t)(row_g0 + i * 16 + lrow) * N + col_g0 + jp * 32 + ch * 4;
            if (MODE == 2) v += *reinterpret_cast<const f32x4*>(residual + at);
            *reinterpret_cast<f32x4*>(o + at) = v;
          }
          __builtin_amdgcn_wave_barrier();
        }
      }
    }
  };

  for (;;) {
    for (int t = 0; t < nk; t += 2) {
      ktile(t, 0);
      ktile(t + 1, 1);
    }
    epilogue(cur);
    if (!has_next) break;
    item += G;
    cur = nxt;
    wv = cur.n0 + wn * 64 < N;
    has_next = item + G < n_items;
    if (has_next) nxt = bases_of(item + G);
  }
#undef CRS_MFMA_QUAD
  if (VAR != 3 && wm == 0) __builtin_amdgcn_s_barrier();   // the barrier row group 1 executed first: both groups end even
}

template <int MODE, int VAR>
int launch8v(const _Float16* a, const _Float16* w, const float* bias, const float* residual, void* out, int m, int n, int k,
             int splits, int cus, hipStream_t stream) {
  const int items = ((n + PN - 1) / PN) * (m / PM) * (MODE == 3 ? splits : 1);
  // the stream pays for the fp16 epilogues (bias / GELU; + 8 % at K = 768 and 384: their VALU and stores sit beside the next item's
  // first phases) and measured 4 % slower for the fp32 + residual ones (the residual loads of the 16-row passes drain behind the
  // transfers in flight): those keep one workgroup per item.  CRS_GEMM8_VAR=1: one workgroup per item everywhere (A/B)
  const bool persist = VAR != 1 && items > cus && MODE < 2;
  const void* kernel = persist ? reinterpret_cast<const void*>(&gemm8_kernel<MODE, true, VAR>) : reinterpret_cast<const void*>(&gemm8_kernel<MODE, false, VAR>);
  static bool done[2] = {false, false};
  if (!done[persist]) {
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLds8);
    if (e != hipSuccess) return (int)e;
    done[persist] = true;
  }
  const int ksplit = k / (MODE == 3 ? splits : 1);
  if (persist)
    hipLaunchKernelGGL((gemm8_kernel<MODE, true, VAR>), dim3(cus), dim3(kP8Threads), kLds8, stream, a, w, bias, residual, out, m, n, k, ksplit, items);
  else
    hipLaunchKernelGGL((gemm8_kernel<MODE, false, VAR>), dim3(items), dim3(kP8Threads), kLds8, stream, a, w, bias, residual, out, m, n, k, ksplit, items);
  return (int)hipGetLastError();
}

int gemm8_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) cus = p.multiProcessorCount;
    else cus = 256;
  }
  return cus;
}

template <int MODE>
int launch8(const _Float16* a, const _Float16* w, const float* bias, const float* residual, void* out, int m, int n, int k,
            int splits, hipStream_t stream) {
  static int var = -1;
  if (var < 0) { const char* e = getenv("CRS_GEMM8_VAR"); var = (e && e[0] >= '0' && e[0] <= '3') ? e[0] - '0' : 0; }
  const int cus = gemm8_cus();
  switch (var) {
    case 1: return launch8v<MODE, 1>(a, w, bias, residual, out, m, n, k, splits, cus, stream);
    case 2: return launch8v<MODE, 2>(a, w, bias, residual, out, m, n, k, splits, cus, stream);
    case 3: return launch8v<MODE, 3>(a, w, bias, residual, out, m, n, k, splits, cus, stream);
    default: return launch8v<MODE, 0>(a, w, bias, residual, out, m, n, k, splits, cus, stream);
  }
}

}  // namespace

// Shapes the phase-scheduled kernel takes: whole 256 x 256 tiles, K a multiple of 128 and >= 256 (two k-tiles in the
// prologue), 16-byte aligned rows.  CRS_GEMM8=0 disables it (A/B runs); CRS_GEMM8_MIN_WGS moves the lower limit.
bool gemm8_applies(int m, int n, int k, int mode) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("CRS_GEMM8"); on = (e && e[0] == '0') ? 0 : 1; }
  // (N: whole 256-column blocks, or -- round 3, CRS_GEMM8_HALF=0 off -- a last block of 128: MiniLM's 384 / 1152)
  static int half = -1;
  if (half < 0) { const char* e = getenv("CRS_GEMM8_HALF"); half = (e && e[0] == '0') ? 0 : 1; }
  const bool n_ok = n % PN == 0 || (half && n % 128 == 0 && n > PN);
  if (!on || m % PM || !n_ok || k % 128 || k < 256) return false;
  static long min_wgs = -1;
  if (min_wgs < 0) { const char* e = getenv("CRS_GEMM8_MIN_WGS"); min_wgs = e ? atol(e) : 128; }
  return (long)(m / PM) * ((n + PN - 1) / PN) >= min_wgs;
}

int gemm8_launch(const _Float16* a, const _Float16* w, const float* bias, const float* residual, void* out, int m, int n, int k, int mode,
                 hipStream_t stream) {
  switch (mode) {
    case 0: return launch8<0>(a, w, bias, residual, out, m, n, k, 1, stream);
    case 1: return launch8<1>(a, w, bias, residual, out, m, n, k, 1, stream);
    case 2: return launch8<2>(a, w, bias, residual, out, m, n, k, 1, stream);
    default: return -1;
  }
}

// Split-K form for projections whose output has too few 256 x 256 tiles to fill the chip (bge-base's N = 768 at a few
// thousand tokens): the number of K slabs (0 = not applicable) such that tiles x slabs >= 128 workgroups, every slab a
// multiple of 128 columns and >= 256; the LayerNorm kernel that follows sums the fp32 slabs (it is instantiated for
// 2 / 3 / 4 / 6 / 8 of them).
int gemm8_splitk(int m, int n, int k) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("CRS_GEMM8"); on = (e && e[0] == '0') ? 0 : 1; }
  if (!on || m % PM || n % PN || m < 2048) return 0;
  const long tiles = (long)(m / PM) * (n / PN);
  if (tiles >= 128) return 0;
  const int cand[5] = {2, 3, 4, 6, 8};
  for (int s : cand)
    if (k % s == 0 && (k / s) % 128 == 0 && k / s >= 256 && tiles * s >= 128) return s;
  return 0;
}

int gemm8_splitk_launch(const _Float16* a, const _Float16* w, float* partials, int m, int n, int k, int splits, hipStream_t stream) {
  return launch8<3>(a, w, nullptr, nullptr, partials, m, n, k, splits, stream);
}

}  // namespace crs
