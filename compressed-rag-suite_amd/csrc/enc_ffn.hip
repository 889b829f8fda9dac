// enc_ffn.hip -- the feed-forward block's two GEMMs in one kernel (query batches, H <= 384):
//
//   y[s] = gelu( x W_up[f0:f0+96, :]^T + b_up[f0:f0+96] ) W_down[:, f0:f0+96]^T         s = f0 / 96
//
// i.e. a workgroup takes 64 token rows and ONE 96-wide slice of the intermediate dimension through both
// projections and leaves an fp32 partial of the block's output; the LayerNorm kernel that follows sums the
// F/96 partials together with bias and residual (it already sums split-K partials, enc_misc.hip).
// Same motive as enc_qkvattn.hip: on the retrieve path the encoder is bound by its chain of ~6 us launches,
// and a slice of the intermediate activations never has to leave the CU -- the up projection's output
// columns ARE the down projection's contraction range, so no other workgroup's data is needed.
// STATUS: parity-green, but measured SLOWER than the two launches it replaces (forward 0.295 vs 0.274 ms at
// 64 x 16 tokens) and therefore off by default (enc_capi.hip, CRS_ENC_FFN=1): the two operand fetches are
// serialised (W_down's slice re-uses W_up's LDS) and the LayerNorm reads 16 partials instead of 4.
//   * 8 waves, grid = (token blocks of 64) x (F / 96): 256 workgroups for 1024 tokens of MiniLM;
//   * phase 1 = the one-shot panel GEMM of enc_gemm.hip: x panel [64, H] + the slice's 96 rows of W_up by
//     LDS-DMA (source-side swizzle), 2 x 3 tiles of v_mfma_f32_32x32x16_f16, + bias, erf-GELU, fp16 -> LDS;
//   * phase 2: the slice's 96 COLUMNS of W_down ([H, 96], 192-byte rows, chunk index XOR-swizzled by
//     (row >> 2) & 3) replace W_up in LDS by a second one-shot DMA; [64, H] = 2 x H/32 tiles over 8 waves,
//     6 k-steps each; the accumulators go through an fp32 LDS tile so that the partial leaves as
//     row-contiguous 16-byte stores (a 4-byte-per-lane store instruction costs a wave the same ~100 cycles).

#include "enc.h"

namespace crs {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int kFfThreads = 512;
constexpr int FM = 64;      // token rows per workgroup
constexpr int FS = 96;      // slice of the intermediate dimension
constexpr int FROW = FS + 8;   // halves per row of the activated slice in LDS (208-byte rows: conflict-free fragment reads)
constexpr int DCPR = FS / 8;   // 16-byte chunks per W_down slice row (12)

__device__ __forceinline__ float gelu_erf_f(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erf_abs = 1.0f - poly * __expf(-z * z);
  return 0.5f * x * (1.0f + (x < 0.f ? -erf_abs : erf_abs));
}

__global__ __launch_bounds__(kFfThreads, 1) void ffn_slice_kernel(const _Float16* __restrict__ x16,
                                                                 const _Float16* __restrict__ Wup,    // [F, H]
                                                                 const float* __restrict__ bup,       // [F]
                                                                 const _Float16* __restrict__ Wdown,  // [H, F]
                                                                 float* __restrict__ y32,             // [F/96][T][H]
                                                                 int T, int H, int F) {
  extern __shared__ __attribute__((aligned(16))) char fsm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * FM, slice = blockIdx.y, f0 = slice * FS;
  const int cpr = H >> 3;
  char* sa = fsm;                                   // x16 panel [FM][H]
  char* sw = fsm + FM * cpr * 16;                   // W_up slice [FS][H]; later W_down slice [H][FS]
  const int w_bytes = (FS * cpr > H * DCPR ? FS * cpr : H * DCPR) * 16;
  _Float16* sact = reinterpret_cast<_Float16*>(sw + w_bytes);   // gelu(up) [FM][FROW]
  const int fr = lane & 31, fh = lane >> 5;

  // ---- phase 1a: x panel + W_up slice, one shot
  {
    const int total = (FM + FS) * cpr;
    for (int base = wave * 64; base < total; base += kFfThreads) {
      const int p = base + lane;
      const int row = p / cpr, cp = p - row * cpr;
      const int c = (cp & ~15) | ((cp ^ row) & 15);
      const _Float16* g = (row < FM) ? x16 + (size_t)min(m0 + row, T - 1) * H + c * 8
                                     : Wup + (size_t)(f0 + row - FM) * H + c * 8;
      __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(fsm + base * 16), 16, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- phase 1b: act[FM, FS] = gelu(x W_up^T + b): 2 x 3 tiles, waves 0..5
  if (wave < 6) {
    const int rb = wave % 2, cb = wave / 2;
    const int arow = rb * 32 + fr, wrow = cb * 32 + fr;
    const char* pa = sa + arow * (cpr * 16);
    const char* pw = sw + wrow * (cpr * 16);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int ksteps = H >> 4;
#pragma unroll 4
    for (int ks = 0; ks < ksteps; ++ks) {
      const int c = ks * 2 + fh;
      const f16x8 af = *reinterpret_cast<const f16x8*>(pa + (((c & ~15) | ((c ^ arow) & 15)) << 4));
      const f16x8 bf = *reinterpret_cast<const f16x8*>(pw + (((c & ~15) | ((c ^ wrow) & 15)) << 4));
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc, 0, 0, 0);
    }
    const int col = cb * 32 + fr;
    const float bv = bup[f0 + col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
      sact[row * FROW + col] = (_Float16)gelu_erf_f(acc[r] + bv);
    }
  }
  __syncthreads();   // W_up no longer needed, activated slice complete

  // ---- phase 2a: W_down[:, f0:f0+96] -> LDS rows of 12 chunks, chunk index ^ ((row >> 2) & 3)
  {
    const int total = H * DCPR;                     // a multiple of 64 (H multiple of 128)
    for (int base = wave * 64; base < total; base += kFfThreads) {
      const int p = base + lane;
      const int row = p / DCPR, cp = p - row * DCPR;
      const int c = cp ^ ((row >> 2) & 3);
      const _Float16* g = Wdown + (size_t)row * F + f0 + c * 8;
      __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(sw + base * 16), 16, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- phase 2b: y[FM, H] = act W_down_slice^T: wave w -> row block w & 1, column blocks ncb_w * (w >> 1) ..
  const int ncb = H >> 5;                            // 32-column blocks (12 at H = 384)
  const int per_wave = (ncb + 3) / 4;                // column blocks per wave (3)
  const int rb = wave & 1, cb0 = (wave >> 1) * per_wave;
  f32x16 acc[3];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  {
    const int arow = rb * 32 + fr;
#pragma unroll
    for (int ks = 0; ks < FS / 16; ++ks) {
      const int c = ks * 2 + fh;
      const f16x8 af = *reinterpret_cast<const f16x8*>(&sact[arow * FROW + c * 8]);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        if (j < per_wave && cb0 + j < ncb) {
          const int wrow = (cb0 + j) * 32 + fr;
          const f16x8 bf = *reinterpret_cast<const f16x8*>(sw + (wrow * DCPR + (c ^ ((wrow >> 2) & 3))) * 16);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc[j], 0, 0, 0);
        }
      }
    }
  }
  __syncthreads();   // every fragment read done: the operand buffers become the fp32 output tile

  // ---- accumulators -> LDS tile [FM][H + 4] -> row-contiguous 16-byte stores
  float* tile = reinterpret_cast<float*>(fsm);
  const int ts = H + 4;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    if (j < per_wave && cb0 + j < ncb) {
      const int col = (cb0 + j) * 32 + fr;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        tile[row * ts + col] = acc[j][r];
      }
    }
  }
  __syncthreads();
  float* out = y32 + (size_t)slice * T * H;
  const int q4 = H >> 2;                             // float4 pieces per row
  for (int id = tid; id < FM * q4; id += kFfThreads) {
    const int row = id / q4, c4 = id - row * q4;
    const int gr = m0 + row;
    if (gr < T) *reinterpret_cast<f32x4*>(out + (size_t)gr * H + c4 * 4) = *reinterpret_cast<const f32x4*>(&tile[row * ts + c4 * 4]);
  }
}

int ffn_lds_bytes(int H) {
  const int cpr = H >> 3;
  const int w_bytes = (FS * cpr > H * DCPR ? FS * cpr : H * DCPR) * 16;
  const int phase = FM * cpr * 16 + w_bytes + FM * FROW * 2;
  const int tile = FM * (H + 4) * 4;
  return phase > tile ? phase : tile;
}

}  // namespace

// number of fp32 partials the fused block leaves (0: shape not supported)
int ffn_fused_slices(int hidden, int ffn) {
  if (hidden > 384 || hidden % 128 || ffn % FS) return 0;
  if (ffn_lds_bytes(hidden) > 160 * 1024) return 0;
  const int ns = ffn / FS;
  return (ns == 16 || ns == 8 || ns == 4) ? ns : 0;   // split counts the LayerNorm kernel is instantiated for
}

int ffn_fused_launch(const _Float16* x16, const _Float16* w_up, const float* b_up, const _Float16* w_down, float* y32,
                     int tokens, int hidden, int ffn, hipStream_t stream) {
  const int ns = ffn_fused_slices(hidden, ffn);
  if (ns == 0) return -1;
  const int lds = ffn_lds_bytes(hidden);
  static int attr_lds = 0;
  if (lds > attr_lds) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_slice_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    attr_lds = lds;
  }
  hipLaunchKernelGGL(ffn_slice_kernel, dim3((tokens + FM - 1) / FM, ns), dim3(kFfThreads), lds, stream, x16, w_up, b_up,
                     w_down, y32, tokens, hidden, ffn);
  return (int)hipGetLastError();
}

}  // namespace crs
