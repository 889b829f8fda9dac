// merge.hip -- reduce partial top-k lists to the final, sorted top-k per query (gfx950).
//
// Two users, one kernel:
//   * stage 2 of crs_cosine_topk: the per-workgroup lists scan.hip leaves behind
//     ([nq, nwg, k] fp32 score + int32 local row);
//   * crs_merge_topk: the per-shard results an RCCL all-gather delivers ([G, nq, k] fp32 + int64
//     global id) -- K10 of SURVEY.md section 2.3, new relative to the single-process reference.
//
// One 256-thread workgroup per query, m = nlists * k_in candidates (L2-resident, KiB-sized):
//   1. every thread streams its strided share and keeps its best score ("bucket maximum");
//   2. the k-th largest of the 256 bucket maxima is a lower bound tau on the true k-th best
//      (those k maxima are k distinct candidates).  It is found without atomics: a 64-lane
//      bitonic sort per wave (shuffles), then three "top-64 of two sorted lists" merges via LDS;
//   3. candidates with score >= tau (typically k .. 2k of them) are appended to an LDS list;
//   4. each candidate's rank among the list = its output slot (order: score desc, id asc).
// If the list overflows (only when thousands of candidates tie), the kernel falls back to
// k_out rounds of workgroup-wide arg-max, which is slow but exact for any input.

#include "scan.h"

namespace crs {
namespace {

constexpr int kThreads = 256;
constexpr int kCap = 1024;  // LDS candidate list capacity
constexpr float kNegInf = -__builtin_huge_valf();

template <typename IdT>
__device__ __forceinline__ bool better(float s, IdT id, float s2, IdT id2) {
  return s > s2 || (s == s2 && id < id2);
}

// full bitonic sort (descending by lane) of one value per lane across a wave64
__device__ __forceinline__ float wave_sort_desc(float v, int lane) {
#pragma unroll
  for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      const float o = __shfl_xor(v, j);
      const bool lower = (lane & j) == 0;
      const bool desc = (lane & k) == 0;       // k == 64: always descending
      const bool want_max = (lower == desc);
      v = want_max ? fmaxf(v, o) : fminf(v, o);
    }
  }
  return v;
}
// v is a bitonic sequence across the wave -> sorted descending
__device__ __forceinline__ float wave_clean_desc(float v, int lane) {
#pragma unroll
  for (int j = 32; j > 0; j >>= 1) {
    const float o = __shfl_xor(v, j);
    v = ((lane & j) == 0) ? fmaxf(v, o) : fminf(v, o);
  }
  return v;
}

// REGE: candidates a thread keeps in registers on the single-pass path (32: 8192 per workgroup, 173 VGPRs; 64: 16384, 244 VGPRs)
template <typename IdT, int REGE = 32>
__global__ __launch_bounds__(kThreads) void merge_kernel(const float* __restrict__ scores,
                                                        const IdT* __restrict__ ids, int nlists,
                                                        int k_in, int k_out, size_t list_stride,
                                                        size_t id_list_stride, size_t q_stride, int64_t id_base,
                                                        float* __restrict__ out_s,
                                                        int64_t* __restrict__ out_i, int lists_per_slice) {
  __shared__ float sh_sorted[4][64];
  __shared__ float sh_cs[kCap];
  __shared__ IdT sh_ci[kCap];
  __shared__ int sh_cnt;
  __shared__ float sh_rs[2][4];
  __shared__ IdT sh_ri[2][4];

  const int q = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Two-level form (merge_launch_i32, > 8192 candidates per query): workgroup (q, y) ranks only lists
  // [y L, (y + 1) L) of the query and leaves their k_out best at out[(q * gridDim.y + y) * k_out]; a second launch merges
  // those gridDim.y short lists.  One-level launches have gridDim.y == 1 and lists_per_slice == nlists.
  const int first_list = blockIdx.y * lists_per_slice;
  nlists = (nlists - first_list < lists_per_slice) ? nlists - first_list : lists_per_slice;
  scores += (size_t)first_list * list_stride;
  ids += (size_t)first_list * id_list_stride;
  const int m = nlists * k_in;
  // candidate e = (list e / k_in, slot e % k_in) lives at list*list_stride + q*q_stride + slot
  // (== e when a query's lists are contiguous).  Threads walk the entries in passes (below), eight passes at a
  // time with all loads issued before any use, so L2 latency is paid once per batch of 8.
  const float* qs = scores + (size_t)q * q_stride;
  const IdT* qi = ids + (size_t)q * q_stride;
  const bool contig = (list_stride == (size_t)k_in) && (id_list_stride == (size_t)k_in);
  // Which entry a thread visits in pass p.  Plain striding (entry = tid + 256 p) gives thread t the SAME slot t % k_in of
  // every list whenever k_in divides 256 (a few slots when it shares a factor) -- and the lists arrive sorted, so a few
  // threads own every list's best entries, the k-th largest "bucket maximum" below is then the maximum of a bucket of
  // 4th-best entries, far too low a bar, the LDS list overflows and the exact-but-slow fallback runs (k = 32 over 512
  // lists: 210-250 us instead of ~ 15).  So a pass covers WHOLE lists (C = the largest multiple of k_in <= 256 entries;
  // threads >= C sit the pass out) and the slot is rotated by the pass number -- a bijection inside each list: every
  // thread meets all slots in turn.
  const bool rotate = k_in > 1 && k_in <= kThreads;
  const int C = rotate ? (kThreads / k_in) * k_in : kThreads;          // entries per pass
  // A pass advances a thread by C / k_in whole lists, so (list, slot) need ONE division per thread, not two per entry
  // (k_in is a run-time value: at 123 slots per list the divisions were most of a 31 488-candidate merge, 53 us)
  const int lpp = rotate ? kThreads / k_in : 0;                          // lists per pass
  const int list0 = rotate ? tid / k_in : 0, slot0 = rotate ? tid - list0 * k_in : 0;
  auto entry_at = [&](int p, int slot) {                                 // slot = (slot0 + p) % k_in, kept by the caller
    if (tid >= C) return -1;
    const int e = rotate ? (list0 + p * lpp) * k_in + slot : p * kThreads + tid;
    return e < m && (!rotate || list0 + p * lpp < nlists) ? e : -1;
  };
  const int n_pass = (m + C - 1) / C;
#define CRS_FOR_EACH_ENTRY(BODY)                                                         \
  for (int p0 = 0, sl_ = slot0; p0 < n_pass; p0 += 8) {                                  \
    float s_[8];                                                                         \
    IdT id_[8];                                                                          \
    _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                      \
      const int e = (p0 + u < n_pass) ? entry_at(p0 + u, sl_) : -1;                      \
      sl_ = (sl_ + 1 == k_in) ? 0 : sl_ + 1;                                             \
      const bool in = e >= 0;                                                            \
      size_t at = 0, ati = 0;                                                            \
      if (in) {                                                                          \
        at = contig ? (size_t)e : (size_t)(e / k_in) * list_stride + (e % k_in);         \
        ati = contig ? (size_t)e : (size_t)(e / k_in) * id_list_stride + (e % k_in);     \
      }                                                                                  \
      s_[u] = qs[at];                                                                    \
      id_[u] = in ? qi[ati] : (IdT)-1;                                                   \
    }                                                                                    \
    _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                      \
      const float s = s_[u];                                                             \
      const IdT id = id_[u];                                                             \
      BODY;                                                                              \
    }                                                                                    \
  }
  float* os = out_s + ((size_t)q * gridDim.y + blockIdx.y) * k_out;
  int64_t* oi = out_i + ((size_t)q * gridDim.y + blockIdx.y) * k_out;

  if (tid == 0) sh_cnt = 0;
  for (int r = tid; r < k_out; r += kThreads) { os[r] = kNegInf; oi[r] = -1; }

  // Fast path (a query's lists contiguous, m <= 16384: every stage-2 merge of the scan): each thread
  // pulls its <= 64 strided candidates into registers with all loads in flight at once, so the
  // candidate set crosses the memory system exactly once (the generic path below walks it twice in
  // batches of 8 and was bound by those serial round trips).
  constexpr int kRegE = REGE;
  const bool cached = contig && n_pass <= kRegE;
  float cs[kRegE];
  IdT ci[kRegE];
  float best = kNegInf;
  if (cached) {
    int sl = slot0;
#pragma unroll
    for (int u = 0; u < kRegE; ++u) {
      const int e = (u < n_pass) ? entry_at(u, sl) : -1;
      sl = (sl + 1 == k_in) ? 0 : sl + 1;
      cs[u] = e >= 0 ? qs[e] : kNegInf;
      ci[u] = e >= 0 ? qi[e] : (IdT)-1;
    }
#pragma unroll
    for (int u = 0; u < kRegE; ++u) best = fmaxf(best, (ci[u] >= 0) ? cs[u] : kNegInf);
  } else {
    // 1. bucket maxima
    CRS_FOR_EACH_ENTRY({ best = fmaxf(best, (id >= 0) ? s : kNegInf); })
  }
  // 2. k-th largest of the 256 maxima (k_out <= 64)
  const float sorted = wave_sort_desc(best, lane);
  sh_sorted[wave][lane] = sorted;
  __syncthreads();
  float a = fmaxf(sh_sorted[0][lane], sh_sorted[1][63 - lane]);
  float b = fmaxf(sh_sorted[2][lane], sh_sorted[3][63 - lane]);
  a = wave_clean_desc(a, lane);
  b = wave_clean_desc(b, lane);
  float t = fmaxf(a, __shfl(b, 63 - lane));
  t = wave_clean_desc(t, lane);  // top-64 of all maxima, identical in every wave
  const float tau = __shfl(t, k_out - 1);

  // 3. candidates >= tau
  if (cached) {
#pragma unroll
    for (int u = 0; u < kRegE; ++u) {
      if (ci[u] >= 0 && cs[u] >= tau) {
        const int p = atomicAdd(&sh_cnt, 1);
        if (p < kCap) { sh_cs[p] = cs[u]; sh_ci[p] = ci[u]; }
      }
    }
  } else {
    CRS_FOR_EACH_ENTRY({
      if (id >= 0 && s >= tau) {
        const int p = atomicAdd(&sh_cnt, 1);
        if (p < kCap) { sh_cs[p] = s; sh_ci[p] = id; }
      }
    })
  }
  __syncthreads();
  const int cnt = sh_cnt;
  if (cnt <= kCap) {
    // 4. rank = output slot
    for (int c = tid; c < cnt; c += kThreads) {
      const float s = sh_cs[c];
      const IdT id = sh_ci[c];
      int rank = 0;
      for (int o = 0; o < cnt; ++o) rank += better<IdT>(sh_cs[o], sh_ci[o], s, id) ? 1 : 0;
      if (rank < k_out) { os[rank] = s; oi[rank] = (int64_t)id + id_base; }
    }
    return;
  }

  // ---- fallback: k_out rounds of "largest key strictly below the previous winner"
  float last_s = __builtin_huge_valf();
  IdT last_i = (IdT)-1;
  for (int r = 0; r < k_out; ++r) {
    float bs = kNegInf;
    IdT bi = (IdT)-1;
    CRS_FOR_EACH_ENTRY({
      const bool after = (s < last_s) || (s == last_s && id > last_i);
      if (id >= 0 && after && (bi < 0 || better<IdT>(s, id, bs, bi))) { bs = s; bi = id; }
    })
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float os2 = __shfl_xor(bs, off);
      const IdT oi2 = __shfl_xor(bi, off);
      const bool take = (oi2 >= 0) && (bi < 0 || better<IdT>(os2, oi2, bs, bi));
      bs = take ? os2 : bs;
      bi = take ? oi2 : bi;
    }
    const int pp = r & 1;
    if (lane == 0) { sh_rs[pp][wave] = bs; sh_ri[pp][wave] = bi; }
    __syncthreads();
    bs = sh_rs[pp][0]; bi = sh_ri[pp][0];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float os2 = sh_rs[pp][w];
      const IdT oi2 = sh_ri[pp][w];
      const bool take = (oi2 >= 0) && (bi < 0 || better<IdT>(os2, oi2, bs, bi));
      bs = take ? os2 : bs;
      bi = take ? oi2 : bi;
    }
    if (bi < 0) break;  // exhausted (uniform); the tail already holds (-inf, -1)
    if (tid == 0) { os[r] = bs; oi[r] = (int64_t)bi + id_base; }
    last_s = bs; last_i = bi;
  }
#undef CRS_FOR_EACH_ENTRY
}

}  // namespace

// stage-2 layout: [nq, nlists, k_in] (a query's candidates are contiguous)
// Stage-2 merges of more than 8192 candidates per query.  Few queries (C4: 64 x 512 streams x 24 / 32 chain slots): ONE launch of
// the 64-per-thread form (16384 candidates per workgroup) -- a second launch would cost more than the lower occupancy.  Many
// queries (C3: 256 x 31 488 dumped representatives): two levels, slices of <= 8192 candidates on the 32-per-thread form (four
// workgroups per CU; the 64-per-thread form measured 73 against 53 us there).
int merge_slice_cand(int nq, int nlists, int k_in) {
  const long m = (long)nlists * k_in;
  return (m <= 16384 && (long)nq * ((m + 8191) / 8192) < 256) ? 16384 : 8192;
}
int merge_slices_for(int nq, int nlists, int k_in) {
  const int cand = merge_slice_cand(nq, nlists, k_in);
  if ((long)nlists * k_in <= cand || k_in > cand) return 1;
  const int per = cand / k_in;
  return (nlists + per - 1) / per;
}
int merge_slices(int nlists, int k_in) {   // upper bound over nq (workspace sizing): the 8192-candidate slicing
  if ((long)nlists * k_in <= 8192 || k_in > 8192) return 1;
  const int per = 8192 / k_in;
  return (nlists + per - 1) / per;
}

int merge_launch_i32(const float* scores, const int* rows, int nlists, int nq, int k_in, int k_out,
                     int64_t id_base, float* out_scores, int64_t* out_ids, float* inter_s, int64_t* inter_i, hipStream_t stream) {
  const int slices = (inter_s && inter_i) ? merge_slices_for(nq, nlists, k_in) : 1;
  if (slices > 1) {
    // level 1: every slice's k_out best (local rows, id_base 0) -> [nq, slices, k_out]; level 2: those short lists
    const int per = merge_slice_cand(nq, nlists, k_in) / k_in;
    hipLaunchKernelGGL((merge_kernel<int>), dim3(nq, slices), dim3(kThreads), 0, stream, scores, rows, nlists, k_in, k_out,
                       (size_t)k_in, (size_t)k_in, (size_t)nlists * k_in, (int64_t)0, inter_s, inter_i, per);
    hipLaunchKernelGGL((merge_kernel<int64_t>), dim3(nq), dim3(kThreads), 0, stream, inter_s, inter_i, slices, k_out, k_out,
                       (size_t)k_out, (size_t)k_out, (size_t)slices * k_out, id_base, out_scores, out_ids, slices);
    return (int)hipGetLastError();
  }
  if ((long)nlists * k_in > 8192 && merge_slice_cand(nq, nlists, k_in) == 16384)
    hipLaunchKernelGGL((merge_kernel<int, 64>), dim3(nq), dim3(kThreads), 0, stream, scores, rows, nlists,
                       k_in, k_out, (size_t)k_in, (size_t)k_in, (size_t)nlists * k_in, id_base, out_scores, out_ids, nlists);
  else
    hipLaunchKernelGGL((merge_kernel<int>), dim3(nq), dim3(kThreads), 0, stream, scores, rows, nlists,
                       k_in, k_out, (size_t)k_in, (size_t)k_in, (size_t)nlists * k_in, id_base, out_scores, out_ids, nlists);
  return (int)hipGetLastError();
}

int merge_launch_i64(const float* scores, const int64_t* ids, int nlists, int nq, int k_in,
                     int k_out, float* out_scores, int64_t* out_ids, hipStream_t stream) {
  // all-gather layout: [nlists, nq, k_in]
  hipLaunchKernelGGL((merge_kernel<int64_t>), dim3(nq), dim3(kThreads), 0, stream, scores, ids,
                     nlists, k_in, k_out, (size_t)nq * k_in, (size_t)nq * k_in, (size_t)k_in, (int64_t)0, out_scores,
                     out_ids, nlists);
  return (int)hipGetLastError();
}

// wire layout (crs_hip.h): nlists blocks of `block_bytes`, each [ids int64 [nq, k_in] | scores fp32 [nq, k_in] | pad]
int merge_launch_wire(const void* wire, size_t block_bytes, size_t scores_off, int nlists, int nq, int k_in, int k_out,
                      float* out_scores, int64_t* out_ids, hipStream_t stream) {
  const int64_t* ids = reinterpret_cast<const int64_t*>(wire);
  const float* scores = reinterpret_cast<const float*>(reinterpret_cast<const char*>(wire) + scores_off);
  hipLaunchKernelGGL((merge_kernel<int64_t>), dim3(nq), dim3(kThreads), 0, stream, scores, ids, nlists, k_in, k_out,
                     block_bytes / 4, block_bytes / 8, (size_t)k_in, (int64_t)0, out_scores, out_ids, nlists);
  return (int)hipGetLastError();
}

}  // namespace crs
