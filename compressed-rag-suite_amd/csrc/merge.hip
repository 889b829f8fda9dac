// merge.hip -- reduce partial top-k lists to the final, sorted top-k per query (gfx950).
//
// Two users, one kernel:
//   * stage 2 of crs_cosine_topk: the per-workgroup lists scan.hip leaves behind
//     ([nwg, nq, k] fp32 score + int32 local row);
//   * crs_merge_topk: the per-shard results an RCCL all-gather delivers ([G, nq, k] fp32 + int64
//     global id) -- K10 of SURVEY.md section 2.3, new relative to the single-process reference.
//
// One 256-thread workgroup per query.  All keys (score, id) are distinct, so the result is
// produced by k_out rounds of "largest key strictly below the previous winner": every thread
// scans its strided share of the (L2-resident, KiB-sized) candidate set, a wave64 shuffle
// reduction and one LDS exchange pick the round's winner.  Order: score descending, id ascending.

#include "scan.h"

namespace crs {
namespace {

constexpr int kThreads = 256;
constexpr float kNegInf = -__builtin_huge_valf();

template <typename IdT>
struct Key {
  float s;
  IdT id;
};

template <typename IdT>
__device__ __forceinline__ bool better(float s, IdT id, float s2, IdT id2) {
  return s > s2 || (s == s2 && id < id2);
}

template <typename IdT>
__global__ __launch_bounds__(kThreads) void merge_kernel(const float* __restrict__ scores,
                                                        const IdT* __restrict__ ids, int nlists,
                                                        int nq, int k_in, int k_out,
                                                        int64_t id_base, float* __restrict__ out_s,
                                                        int64_t* __restrict__ out_i) {
  __shared__ float sh_s[2][4];
  __shared__ IdT sh_i[2][4];
  const int q = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = nlists * k_in;
  const IdT kWorstId = (IdT)0x7fffffff;  // only compared against when s == -inf

  float last_s = __builtin_huge_valf();
  IdT last_i = (IdT)-1;
  for (int r = 0; r < k_out; ++r) {
    float bs = kNegInf;
    IdT bi = kWorstId;
    bool have = false;
    for (int e = tid; e < m; e += kThreads) {
      const int list = e / k_in, j = e - list * k_in;
      const size_t at = ((size_t)list * nq + q) * k_in + j;
      const IdT id = ids[at];
      const float s = scores[at];
      if (id < 0) continue;
      // strictly after the previous winner in the total order
      const bool after = (s < last_s) || (s == last_s && id > last_i);
      if (after && (!have || better<IdT>(s, id, bs, bi))) {
        bs = s; bi = id; have = true;
      }
    }
    if (!have) { bs = kNegInf; bi = (IdT)-1; }
    // wave reduction; "absent" candidates carry id -1 and lose to any present one
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float os = __shfl_xor(bs, off);
      const IdT oi = __shfl_xor(bi, off);
      const bool take = (oi >= 0) && (bi < 0 || better<IdT>(os, oi, bs, bi));
      bs = take ? os : bs;
      bi = take ? oi : bi;
    }
    const int pp = r & 1;
    if (lane == 0) { sh_s[pp][wave] = bs; sh_i[pp][wave] = bi; }
    __syncthreads();
    bs = sh_s[pp][0]; bi = sh_i[pp][0];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float os = sh_s[pp][w];
      const IdT oi = sh_i[pp][w];
      const bool take = (oi >= 0) && (bi < 0 || better<IdT>(os, oi, bs, bi));
      bs = take ? os : bs;
      bi = take ? oi : bi;
    }
    if (tid == 0) {
      out_s[(size_t)q * k_out + r] = (bi >= 0) ? bs : kNegInf;
      out_i[(size_t)q * k_out + r] = (bi >= 0) ? (int64_t)bi + id_base : (int64_t)-1;
    }
    if (bi < 0) {
      // exhausted: fill the tail and stop (uniform across the workgroup)
      if (tid == 0)
        for (int rr = r + 1; rr < k_out; ++rr) {
          out_s[(size_t)q * k_out + rr] = kNegInf;
          out_i[(size_t)q * k_out + rr] = -1;
        }
      break;
    }
    last_s = bs; last_i = bi;
  }
}

}  // namespace

int merge_launch_i32(const float* scores, const int* rows, int nlists, int nq, int k_in, int k_out,
                     int64_t id_base, float* out_scores, int64_t* out_ids, hipStream_t stream) {
  hipLaunchKernelGGL((merge_kernel<int>), dim3(nq), dim3(kThreads), 0, stream, scores, rows, nlists,
                     nq, k_in, k_out, id_base, out_scores, out_ids);
  return (int)hipGetLastError();
}

int merge_launch_i64(const float* scores, const int64_t* ids, int nlists, int nq, int k_in,
                     int k_out, float* out_scores, int64_t* out_ids, hipStream_t stream) {
  hipLaunchKernelGGL((merge_kernel<int64_t>), dim3(nq), dim3(kThreads), 0, stream, scores, ids,
                     nlists, nq, k_in, k_out, (int64_t)0, out_scores, out_ids);
  return (int)hipGetLastError();
}

}  // namespace crs
