// capi.hip -- the extern "C" surface declared in include/crs_hip.h.
#include "../../include/crs_hip.h"

#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "scan.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, const char* detail = "") {
  snprintf(g_err, sizeof g_err, fmt, detail);
  return code;
}
int hip_fail(hipError_t e, const char* where) {
  snprintf(g_err, sizeof g_err, "%s: %s", where, hipGetErrorString(e));
  return CRS_EHIP;
}

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
size_t align_up16(size_t x) { return (x + 255) / 256 * 256; }

int device_cus() {
  static int cus[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (dev < 0 || dev >= 64) return 256;
  if (cus[dev] == 0) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
    cus[dev] = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
  }
  return cus[dev];
}


constexpr int kW1MaxDump = 256;
// streams shorter than this many tiles keep the static stride (1.25 M x 384 rows = 76 tiles per stream measured no gain)
constexpr int kDynMinRounds = 96;

struct Plan {
  int pdim, tile_rows, n_tiles, nwg, kp;   // nwg = tile streams (workgroups per query block)
  int nqb;                                 // query blocks (64 queries each; 32 * wide_nw for the wide kernel)
  int wide_nw;                             // > 0: scan_wide.hip with this many waves per workgroup
  int w1_qg;                               // > 0: scan_w1.hip, this many queries per workgroup (dump selection)
  int i8_tb;                               // 1: scan_i8.hip in a tile-best mode (tb_slots: 0 dump, else chain)
  int tb_nw;                               // > 0: scan_tb.hip with this many waves per workgroup
  int tb_slots;                            //   its chain length (0: dump mode)
  int group_best;                          // 1: the scan leaves tile representatives (scan_refine.hip finishes)
  size_t part_elems;  // nwg * nq * kp
};

// slots per (query, workgroup) partial list: k, or 16 when threshold sharing is on (lists may then be
// dumped unselected)
int partial_width(int k) { return (crs::scan_share_tau() && k <= 16) ? 16 : k; }
size_t tau_bytes(int nq) { return align_up16((size_t)nq * 4); }

bool tb_long_chain() {   // CRS_SCAN_LONG_CHAIN=0: 16 < k <= 64 on long streams back on the threshold kernels (A/B, tests)
  static int v = -1;
  if (v < 0) { const char* e = getenv("CRS_SCAN_LONG_CHAIN"); v = (e && e[0] == '0') ? 0 : 1; }
  return v == 1;
}
// Dynamic tile schedule of the chain-mode tile-best scan (scan_tb.hip): percent of the tiles handed out through the counter and
// tiles per ticket.  Defaults 85 % in granules of 8 (C4 beside the encoder lanes: 44.9 -> 47.2 k q/s on one box; 20 / 50 / 95 / 100 %
// and granules of 4 / 16 measured within 1 % of that or worse; a ticket per tile is bound by the ~90 M atomics/s one address
// takes).  CRS_TB_DYN=0 restores the static stride.  Read per call: tools/tb_dyn_check.py switches it inside one process.
int tb_dyn_percent() {
  const char* e = getenv("CRS_TB_DYN");
  const int v = e ? atoi(e) : 85;
  return v < 0 ? 0 : v > 100 ? 100 : v;
}
int tb_dyn_granule() {
  const char* e = getenv("CRS_TB_DYN_G");
  const int v = e ? atoi(e) : 8;
  return v >= 16 ? 16 : v >= 8 ? 8 : v >= 4 ? 4 : v >= 2 ? 2 : 1;
}
bool tb_enabled() {   // CRS_SCAN_TB=0: always use the threshold/compaction kernel for <= 64 queries (A/B runs, tests)
  static int v = -1;
  if (v < 0) { const char* e = getenv("CRS_SCAN_TB"); v = (e && e[0] == '0') ? 0 : 1; }
  return v == 1;
}

// workspace: [shared thresholds | partial scores | partial rows | stage-1 winners (scores, ids) | two-level merge scratch
// (scores, ids): one k-entry list per 8192 candidates of a query (merge.hip)]
size_t inter_lists(size_t part_elems, int nq) { return part_elems / ((size_t)nq * 4096) + 2; }   // >= merge_slices(nwg, kp)
size_t ws_bytes(size_t part_elems, int nq, int k) {
  const size_t inter = (size_t)nq * inter_lists(part_elems, nq) * k;
  return tau_bytes(nq) + 2 * align_up(part_elems * 4, 256) + align_up((size_t)nq * k * 4, 256) + align_up((size_t)nq * k * 8, 256) +
         align_up(inter * 4, 256) + align_up(inter * 8, 256);
}

int make_plan(int nq, int dim, int k, int64_t n_rows, int slab_type, Plan* p) {
  if (nq <= 0 || dim <= 0 || dim > 1024) return fail(CRS_EINVAL, "nq must be > 0 and 0 < dim <= 1024");
  if (k <= 0 || k > CRS_MAX_K) return fail(CRS_EINVAL, "k must be in 1..CRS_MAX_K");
  if (n_rows <= 0 || n_rows > 0x7fffffffLL - 64) return fail(CRS_EINVAL, "n_rows must be in 1..2^31-65");
  const int cus = device_cus();
  if (cus <= 0) return fail(CRS_EHIP, "no HIP device available%s");
  p->pdim = crs_row_elems(dim, slab_type);
  // kernel family: one of the tile-best kernels, or the classic threshold/compaction scan.  The register-chain forms
  // hold k <= 16; the dump form (short streams) has no such limit and serves k <= 64.  For k > 16 the family is
  // therefore only known once the stream length is: plan for tile-best first, fall back to classic if it has to chain.
  bool allow_w1 = true;
  for (int attempt = 0; attempt < 3; ++attempt) {
    const bool allow_tb = attempt < 2 && tb_enabled();
    p->w1_qg = (allow_tb && allow_w1 && slab_type == CRS_SLAB_F16) ? crs::scan_w1_queries_per_wg(nq, k, p->pdim) : 0;
    p->wide_nw = (allow_tb && !p->w1_qg && slab_type == CRS_SLAB_F16 && k <= 16) ? crs::scan_wide_waves(nq, k, p->pdim) : 0;
    p->tb_nw = 0;
    if (allow_tb && !p->w1_qg && !p->wide_nw && slab_type == CRS_SLAB_F16)
      p->tb_nw = (nq > 64 && k <= 16 && crs::scan_tb_has_8_waves(p->pdim)) ? 8 : 4;
    p->i8_tb = (allow_tb && slab_type == CRS_SLAB_I8) ? 1 : 0;   // scan_i8.hip's tile-best modes
    p->tile_rows = p->w1_qg ? 32 : p->wide_nw ? crs::scan_wide_tile_rows(p->wide_nw, p->pdim) : slab_type == CRS_SLAB_I8 ? crs::scan_i8_tile_rows() : crs::scan_tile_rows(p->pdim);
    p->n_tiles = (int)((n_rows + p->tile_rows - 1) / p->tile_rows);
    const int cap = cus * (p->w1_qg ? 1 : p->wide_nw ? crs::scan_wide_wg_per_cu(p->wide_nw, p->pdim)
                           : p->tb_nw ? crs::scan_tb_wg_per_cu(p->pdim, p->tb_nw) : crs::scan_wg_per_cu());
    // all query blocks of a tile stream must be co-resident: streams = resident slots / query blocks
    const int qpb = p->w1_qg ? p->w1_qg : p->wide_nw ? 32 * p->wide_nw : p->tb_nw ? 16 * p->tb_nw : 64;
    const int nqb = (nq + qpb - 1) / qpb;
    p->nqb = nqb;
    int streams = cap / nqb;
    if (streams < 1) streams = 1;
    if (nqb > 1 && streams >= 8) streams &= ~7;   // whole rounds over the 8 XCDs (scan_common.h: grid mapping)
    p->nwg = p->n_tiles < streams ? p->n_tiles : streams;
    if (nqb > 1 && p->nwg >= 8) p->nwg &= ~7;
    p->kp = partial_width(k);
    p->group_best = 0;
    p->tb_slots = 0;
    if (p->w1_qg) {                          // dump: every tile's representative goes to the partial list
      p->kp = (p->n_tiles + p->nwg - 1) / p->nwg;
      p->group_best = 1;
      // the dump's workspace and merge input grow with the shard (nq * n_rows / 32 entries): past kW1MaxDump entries per
      // (query, stream) -- 1 M rows x 768 at 256 queries is 123 -- the bounded 8-wave chain kernels take over
      // (10 M x 768 at 256 queries would otherwise be 640 MB of workspace and 312 k candidates per query)
      if (p->kp > kW1MaxDump) { allow_w1 = false; continue; }
    } else if (p->wide_nw) {   // register chain of the K best tile representatives per lane
      p->kp = 2 * crs::scan_wide_slots(k);
      p->group_best = 1;
    } else if (p->tb_nw || p->i8_tb) {
      // short streams: every tile's representative goes straight to the partial list ("dump"; merge.hip's
      // single-pass path takes <= 8192 candidates per query); longer ones keep the K best in registers
      const int tps = (p->n_tiles + p->nwg - 1) / p->nwg;
      p->group_best = 1;
      if ((size_t)tps * p->nwg <= 8192 && tps <= 2 * crs::scan_wide_slots(k)) {
        p->kp = tps;
      } else if (k <= 16) {
        p->tb_slots = crs::scan_wide_slots(k);
        p->kp = p->tb_slots;
      } else if (tb_long_chain() && (p->i8_tb ? crs::scan_i8_long_chain_slots(p->pdim, k) : crs::scan_tb_long_chain_slots(p->pdim, p->tb_nw, k)) > 0) {
        // 16 < k <= 64 on a long stream: the same kernels with a 32- / 64-slot chain instead of the threshold kernels,
        // where that chain fits the register file (CRS_SCAN_LONG_CHAIN=0 restores the threshold kernels)
        p->tb_slots = p->i8_tb ? crs::scan_i8_long_chain_slots(p->pdim, k) : crs::scan_tb_long_chain_slots(p->pdim, p->tb_nw, k);
        p->kp = p->tb_slots;
      } else {
        continue;                            // the threshold kernels
      }
    }
    break;
  }
  p->part_elems = (size_t)p->nwg * nq * p->kp;
  return CRS_OK;
}



}  // namespace

namespace crs {
int set_error(int code, const char* msg) {  // shared with enc_capi.hip
  snprintf(g_err, sizeof g_err, "%s", msg);
  return code;
}
}  // namespace crs

extern "C" {

const char* crs_last_error(void) { return g_err; }
int crs_abi_version(void) { return 3; }
int crs_row_elems(int dim, int slab_type) {
  if (dim <= 0) return 0;
  const int g = slab_type == CRS_SLAB_I8 ? 256 : 128;
  return (dim + g - 1) / g * g;
}
int crs_padded_dim(int dim) { return crs_row_elems(dim, CRS_SLAB_F16); }

int crs_slab_append_f32(const float* emb_dev, int64_t n, int dim, int slab_type, void* slab_dev,
                        float* scales_dev, float* shadow_f32_dev, int64_t row0, float* row_err_max_dev, void* stream) {
  if (n < 0 || dim <= 0 || dim > 1024 || row0 < 0) return fail(CRS_EINVAL, "bad n/dim/row0");
  if (slab_type != CRS_SLAB_F16 && slab_type != CRS_SLAB_I8) return fail(CRS_EINVAL, "bad slab_type");
  if (n == 0) return CRS_OK;
  if (!emb_dev || !slab_dev) return fail(CRS_EINVAL, "null pointer");
  if (slab_type == CRS_SLAB_I8 && !scales_dev) return fail(CRS_EINVAL, "int8 slab needs scales");
  const int e = crs::slab_append_launch(emb_dev, n, dim, crs_row_elems(dim, slab_type), slab_type, slab_dev,
                                        scales_dev, shadow_f32_dev, row0, row_err_max_dev, (hipStream_t)stream);
  return e ? hip_fail((hipError_t)e, "slab_append") : CRS_OK;
}

int crs_queries_to_f16(const float* q_dev, int nq, int dim, int slab_type, void* q16_dev, void* stream) {
  if (nq < 0 || dim <= 0 || dim > 1024) return fail(CRS_EINVAL, "bad nq/dim");
  if (slab_type != CRS_SLAB_F16 && slab_type != CRS_SLAB_I8) return fail(CRS_EINVAL, "bad slab_type");
  if (nq == 0) return CRS_OK;
  if (!q_dev || !q16_dev) return fail(CRS_EINVAL, "null pointer");
  const int e = crs::queries_to_f16_launch(q_dev, nq, dim, crs_row_elems(dim, slab_type),
                                           reinterpret_cast<_Float16*>(q16_dev), (hipStream_t)stream);
  return e ? hip_fail((hipError_t)e, "queries_to_f16") : CRS_OK;
}

int crs_scan_workspace_bytes(int nq, int dim, int k, int64_t n_rows, size_t* bytes) {
  if (!bytes) return fail(CRS_EINVAL, "null pointer");
  Plan p, p8;
  int rc = make_plan(nq, dim, k, n_rows, CRS_SLAB_F16, &p);
  if (rc) return rc;
  rc = make_plan(nq, dim, k, n_rows, CRS_SLAB_I8, &p8);
  if (rc) return rc;
  // the call does not say which slab type will be searched: cover the plans of both, and the largest grid the
  // threshold kernels can use (resident workgroups), which does not depend on n_rows
  const size_t cap = (size_t)device_cus() * crs::scan_wg_per_cu();
  const size_t classic = ws_bytes(cap * nq * partial_width(k), nq, k);
  size_t planned = ws_bytes(p.part_elems, nq, k);
  const size_t planned8 = ws_bytes(p8.part_elems, nq, k);
  if (planned8 > planned) planned = planned8;
  *bytes = classic > planned ? classic : planned;
  return CRS_OK;
}

static int run_scan(const Plan& p, const void* q16, int nq, int slab_type, const void* slab,
                    const float* scales, int64_t n_rows, int k, void* ws, hipStream_t st,
                    float** ps_out, int** pr_out) {
  unsigned* tau = reinterpret_cast<unsigned*>(ws);
  float* ps = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + tau_bytes(nq));
  int* pr = reinterpret_cast<int*>(reinterpret_cast<char*>(ps) + align_up(p.part_elems * 4, 256));
  const bool share = crs::scan_share_tau();
  if (share) {
    const hipError_t me = hipMemsetAsync(tau, 0, tau_bytes(nq), st);
    if (me != hipSuccess) return (int)me;
  }
  crs::ScanArgs a;
  a.q = reinterpret_cast<const _Float16*>(q16);
  a.slab = slab;
  a.scales = scales;
  a.part_scores = ps;
  a.part_rows = pr;
  a.stamps = nullptr;
  a.tau_shared = share ? tau : nullptr;
  a.kp = p.kp;
  {
    static int boot = -1;
    if (boot < 0) { const char* e = getenv("CRS_SCAN_BOOT"); boot = (e && e[0] == '0') ? 0 : 1; }
    {
      static int sched = -1;
      if (sched < 0) { const char* e = getenv("CRS_SCAN_SCHED"); sched = (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 2; }
      a.sched = sched;
    }
    // short streams only: the bootstrap pays when a workgroup sees few tiles (see scan.hip)
    a.boot = (boot && p.n_tiles / (p.nwg > 0 ? p.nwg : 1) < 24) ? 1 : 0;
  }
  a.n_rows = (int)n_rows;
  a.n_tiles = p.n_tiles;
  a.nq = nq;
  a.k = k;
  a.nwg = p.nwg;
  a.nqb = p.nqb;
  a.ticket = nullptr;
  a.t_dyn = p.n_tiles;
  a.dyn_mask = 0;
  if (((slab_type == CRS_SLAB_F16 && p.tb_nw) || (slab_type == CRS_SLAB_I8 && p.i8_tb && p.pdim <= 768)) && p.tb_slots > 0 && p.nqb == 1 && !share) {
    // (int8 rows of 1024 elements stay static: those instantiations spill, and the ticket's register must not travel through scratch
    // while its value is in flight)
    // long chain-mode streams: the last tb_dyn_percent() of the tiles are drawn from a counter (scan_tb.hip, scan_i8.hip), which lives in the
    // (otherwise unused) shared-threshold words at the head of the workspace and is zeroed in stream order ahead of the scan
    const int rounds = p.n_tiles / p.nwg, pct = tb_dyn_percent();
    const char* me = getenv("CRS_TB_DYN_MIN");   // tests: dynamic schedule on short streams too
    const int min_rounds = me ? atoi(me) : kDynMinRounds;
    if (pct > 0 && rounds >= (min_rounds < 4 ? 4 : min_rounds)) {
      int stat = (int)((int64_t)rounds * (100 - pct) / 100);
      if (stat < 2) stat = 2;
      a.t_dyn = stat * p.nwg;
      a.ticket = tau;
      a.dyn_mask = tb_dyn_granule() - 1;
      const int ze = crs::scan_ticket_zero(a.ticket, st);
      if (ze) return ze;
    }
  }
  const int e = p.w1_qg ? crs::scan_launch_w1(a, p.pdim, st)
                : p.wide_nw ? crs::scan_launch_wide(a, p.pdim, p.wide_nw, st)
                
                : (slab_type == CRS_SLAB_I8) ? crs::scan_launch_i8(a, p.pdim, p.i8_tb ? p.tb_slots : -1, st)
                : p.tb_nw ? crs::scan_launch_tb(a, p.pdim, p.tb_nw, p.tb_slots, st)
                               : crs::scan_launch_f16(a, p.pdim, p.nwg, st);
  *ps_out = ps;
  *pr_out = pr;
  return e;
}

int crs_cosine_topk(const void* q16_dev, int nq, int dim, int slab_type, const void* slab_dev,
                    const float* scales_dev, int64_t n_rows, int k, int64_t id_base,
                    void* workspace_dev, size_t workspace_bytes, float* out_scores_dev,
                    int64_t* out_ids_dev, void* stream) {
  if (slab_type != CRS_SLAB_F16 && slab_type != CRS_SLAB_I8) return fail(CRS_EINVAL, "bad slab_type");
  Plan p;
  int rc = make_plan(nq, dim, k, n_rows, slab_type, &p);
  if (rc) return rc;
  if (!q16_dev || !slab_dev || !workspace_dev || !out_scores_dev || !out_ids_dev)
    return fail(CRS_EINVAL, "null pointer");
  if (slab_type == CRS_SLAB_I8 && !scales_dev) return fail(CRS_EINVAL, "int8 slab needs scales");
  if (((uintptr_t)q16_dev | (uintptr_t)slab_dev) & 15) return fail(CRS_EINVAL, "q/slab must be 16-byte aligned");
  if (workspace_bytes < ws_bytes(p.part_elems, nq, k)) return fail(CRS_ENOSPC, "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  float* ps;
  int* pr;
  int e = run_scan(p, q16_dev, nq, slab_type, slab_dev, scales_dev, n_rows, k, workspace_dev, st, &ps, &pr);
  if (e == -1) return fail(CRS_EINVAL, "unsupported padded dimension");
  if (e) return hip_fail((hipError_t)e, "scan launch");
  float* win_s = reinterpret_cast<float*>(reinterpret_cast<char*>(pr) + align_up(p.part_elems * 4, 256));
  int64_t* win_i = reinterpret_cast<int64_t*>(reinterpret_cast<char*>(win_s) + align_up((size_t)nq * k * 4, 256));
  float* inter_s = reinterpret_cast<float*>(reinterpret_cast<char*>(win_i) + align_up((size_t)nq * k * 8, 256));
  int64_t* inter_i = reinterpret_cast<int64_t*>(reinterpret_cast<char*>(inter_s) + align_up((size_t)nq * inter_lists(p.part_elems, nq) * k * 4, 256));
  if ((size_t)crs::merge_slices(p.nwg, p.kp) > inter_lists(p.part_elems, nq)) inter_s = nullptr;   // (cannot happen: see inter_lists)
  if (!p.group_best) {
    e = crs::merge_launch_i32(ps, pr, p.nwg, nq, p.kp, k, id_base, out_scores_dev, out_ids_dev, inter_s, inter_i, st);
    if (e) return hip_fail((hipError_t)e, "merge launch");
    return CRS_OK;
  }
  // group-best variants: k best representatives (local rows) -> their row groups re-scored and ranked
  e = crs::merge_launch_i32(ps, pr, p.nwg, nq, p.kp, k, 0, win_s, win_i, inter_s, inter_i, st);
  if (e) return hip_fail((hipError_t)e, "merge launch");
  e = (slab_type == CRS_SLAB_I8)
          ? crs::refine_i8_launch(reinterpret_cast<const _Float16*>(q16_dev), nq, p.pdim, slab_dev, scales_dev, (int)n_rows, win_s,
                                  win_i, k, p.tile_rows, id_base, out_scores_dev, out_ids_dev, st)
          : crs::refine_launch(reinterpret_cast<const _Float16*>(q16_dev), nq, p.pdim, reinterpret_cast<const _Float16*>(slab_dev),
                               (int)n_rows, win_s, win_i, k, p.tile_rows, id_base, out_scores_dev, out_ids_dev, st);
  if (e == -1) return fail(CRS_EINVAL, "unsupported padded dimension");
  if (e) return hip_fail((hipError_t)e, "refine launch");
  return CRS_OK;
}

int crs_merge_topk(const float* scores_dev, const int64_t* ids_dev, int nlists, int nq, int k_in,
                   int k_out, float* out_scores_dev, int64_t* out_ids_dev, void* stream) {
  if (nlists <= 0 || nq <= 0 || k_in <= 0 || k_out <= 0 || k_out > CRS_MAX_K) return fail(CRS_EINVAL, "bad sizes (k_out <= CRS_MAX_K)");
  if (!scores_dev || !ids_dev || !out_scores_dev || !out_ids_dev) return fail(CRS_EINVAL, "null pointer");
  const int e = crs::merge_launch_i64(scores_dev, ids_dev, nlists, nq, k_in, k_out, out_scores_dev,
                                      out_ids_dev, (hipStream_t)stream);
  return e ? hip_fail((hipError_t)e, "merge launch") : CRS_OK;
}

int crs_rescore_f32(const float* q32_dev, int nq, int dim, const float* shadow_dev, int64_t n_rows,
                    int64_t id_base, int k, float* scores_dev, int64_t* ids_dev, void* stream) {
  if (nq <= 0 || dim <= 0 || k <= 0 || k > 64 || n_rows <= 0) return fail(CRS_EINVAL, "bad sizes (k <= 64)");
  if (!q32_dev || !shadow_dev || !scores_dev || !ids_dev) return fail(CRS_EINVAL, "null pointer");
  const int e = crs::rescore_launch(q32_dev, nq, dim, shadow_dev, n_rows, id_base, k, scores_dev,
                                    ids_dev, (hipStream_t)stream);
  return e ? hip_fail((hipError_t)e, "rescore launch") : CRS_OK;
}

int crs_score_rows_f32(const float* q32_dev, int nq, int dim, const float* shadow_dev, int64_t n_rows, int64_t id_base, int k,
                       const int64_t* ids_dev, float* scores_dev, void* stream) {
  if (nq <= 0 || dim <= 0 || k <= 0 || n_rows <= 0) return fail(CRS_EINVAL, "bad sizes");
  if (!q32_dev || !shadow_dev || !ids_dev || !scores_dev) return fail(CRS_EINVAL, "null pointer");
  const int e = crs::score_rows_launch(q32_dev, nq, dim, shadow_dev, n_rows, id_base, k, ids_dev, scores_dev, (hipStream_t)stream);
  return e ? hip_fail((hipError_t)e, "score_rows launch") : CRS_OK;
}

int crs_refine_f32(const float* q32_dev, int nq, int dim, const float* shadow_dev, int64_t n_rows,
                   int64_t id_base, const int64_t* cand_ids_dev, int k_in, int k_out,
                   float* out_scores_dev, int64_t* out_ids_dev, void* stream) {
  if (nq <= 0 || dim <= 0 || n_rows <= 0 || k_out <= 0 || k_in < k_out || k_in > CRS_MAX_K)
    return fail(CRS_EINVAL, "bad sizes (1 <= k_out <= k_in <= CRS_MAX_K)");
  if (!q32_dev || !shadow_dev || !cand_ids_dev || !out_scores_dev || !out_ids_dev) return fail(CRS_EINVAL, "null pointer");
  const int e = crs::refine_f32_launch(q32_dev, nq, dim, shadow_dev, n_rows, id_base, cand_ids_dev, k_in, k_out,
                                       out_scores_dev, out_ids_dev, (hipStream_t)stream);
  return e ? hip_fail((hipError_t)e, "refine_f32 launch") : CRS_OK;
}

// exactness workspace: [threshold f32 [nq] | counter i32 [nq] | row lists i64 [nq, cap]]
// exactness workspace: [thresholds | counters | the escalation kernel's blocks-through counter (256 B) | lists]
static size_t exact_done_off(int nq) { return 2 * align_up((size_t)nq * 4, 256); }
static size_t exact_lists_off(int nq) { return exact_done_off(nq) + 256; }
size_t crs_exact_workspace_bytes(int nq, int cap) {
  return (nq > 0 && cap > 0) ? exact_lists_off(nq) + (size_t)nq * cap * 8 : 0;
}
float crs_exact_row_error_bound(int dim, int slab_type) { return crs::exact_err_rows_bound(dim, slab_type); }

static int exact_args_ok(int nq, int dim, int slab_type, int cap, size_t ws_bytes, const void* ws) {
  if (slab_type != CRS_SLAB_F16 && slab_type != CRS_SLAB_I8) return fail(CRS_EINVAL, "bad slab_type");
  if (nq <= 0 || dim <= 0 || dim > 1024) return fail(CRS_EINVAL, "bad nq/dim");
  if (cap < 64 || cap > CRS_EXACT_MAX_CAP) return fail(CRS_EINVAL, "cap must be in 64..CRS_EXACT_MAX_CAP");
  if (!ws || ((uintptr_t)ws & 15)) return fail(CRS_EINVAL, "exactness workspace must be a 16-byte aligned device pointer");
  if (ws_bytes < crs_exact_workspace_bytes(nq, cap)) return fail(CRS_ENOSPC, "exactness workspace too small");
  return CRS_OK;
}

int crs_refine_f32_cert(const float* q32_dev, const void* q16_dev, int nq, int dim, int slab_type, const float* shadow_dev,
                        int64_t n_rows, int64_t id_base, const int64_t* cand_ids_dev, const float* cand_scores_dev, int k_in,
                        int k_out, float row_err_max, float* out_scores_dev, int64_t* out_ids_dev, int32_t* status_dev,
                        void* exact_ws_dev, size_t exact_ws_bytes, int cap, void* stream) {
  int rc = exact_args_ok(nq, dim, slab_type, cap, exact_ws_bytes, exact_ws_dev);
  if (rc) return rc;
  if (n_rows <= 0 || k_out <= 0 || k_in < k_out || k_in > CRS_MAX_K) return fail(CRS_EINVAL, "bad sizes (1 <= k_out <= k_in <= CRS_MAX_K)");
  if (!q32_dev || !q16_dev || !shadow_dev || !cand_ids_dev || !cand_scores_dev || !out_scores_dev || !out_ids_dev || !status_dev)
    return fail(CRS_EINVAL, "null pointer");
  if (!(row_err_max >= 0.f)) row_err_max = crs::exact_err_rows_bound(dim, slab_type);   // untracked (or NaN): the analytic worst case
  char* ws = reinterpret_cast<char*>(exact_ws_dev);
  const int e = crs::refine_cert_launch(q32_dev, reinterpret_cast<const _Float16*>(q16_dev), nq, dim, crs_row_elems(dim, slab_type),
                                        slab_type, shadow_dev, n_rows, id_base, cand_ids_dev, cand_scores_dev, k_in, k_out, row_err_max,
                                        out_scores_dev, out_ids_dev, status_dev, reinterpret_cast<float*>(ws),
                                        reinterpret_cast<int*>(ws + align_up((size_t)nq * 4, 256)), reinterpret_cast<int*>(ws + exact_done_off(nq)),
                                        (hipStream_t)stream);
  return e ? hip_fail((hipError_t)e, "refine_f32_cert launch") : CRS_OK;
}

int crs_escalate_exact(const float* q32_dev, const void* q16_dev, int nq, int dim, int slab_type, const void* slab_dev,
                       const float* scales_dev, const float* shadow_dev, int64_t n_rows, int64_t id_base, int k_out,
                       float* out_scores_dev, int64_t* out_ids_dev, int32_t* status_dev, void* exact_ws_dev,
                       size_t exact_ws_bytes, int cap, void* stream) {
  int rc = exact_args_ok(nq, dim, slab_type, cap, exact_ws_bytes, exact_ws_dev);
  if (rc) return rc;
  if (n_rows <= 0 || n_rows > 0x7fffffffLL - 64 || k_out <= 0 || k_out > CRS_MAX_K) return fail(CRS_EINVAL, "bad sizes");
  if (!q32_dev || !q16_dev || !slab_dev || !shadow_dev || !out_scores_dev || !out_ids_dev || !status_dev) return fail(CRS_EINVAL, "null pointer");
  if (slab_type == CRS_SLAB_I8 && !scales_dev) return fail(CRS_EINVAL, "int8 slab needs scales");
  if (((uintptr_t)q16_dev | (uintptr_t)slab_dev) & 15) return fail(CRS_EINVAL, "q/slab must be 16-byte aligned");
  char* ws = reinterpret_cast<char*>(exact_ws_dev);
  const int e = crs::escalate_launch(q32_dev, reinterpret_cast<const _Float16*>(q16_dev), nq, dim, crs_row_elems(dim, slab_type), slab_type,
                                     slab_dev, scales_dev, shadow_dev, n_rows, id_base, k_out, out_scores_dev, out_ids_dev, status_dev,
                                     reinterpret_cast<const float*>(ws), reinterpret_cast<int*>(ws + align_up((size_t)nq * 4, 256)),
                                     reinterpret_cast<int*>(ws + exact_done_off(nq)),
                                     reinterpret_cast<int64_t*>(ws + exact_lists_off(nq)), cap, device_cus(), (hipStream_t)stream);
  if (e == -1) return fail(CRS_EINVAL, "unsupported padded dimension");
  return e ? hip_fail((hipError_t)e, "escalate launch") : CRS_OK;
}

size_t crs_wire_scores_offset(int nq, int k) { return (nq > 0 && k > 0) ? (size_t)nq * k * 8 : 0; }
size_t crs_wire_bytes(int nq, int k) {
  return (nq > 0 && k > 0) ? (size_t)nq * k * 8 + align_up((size_t)nq * k * 4, 8) : 0;
}
int crs_merge_topk_wire(const void* wire_dev, int nlists, int nq, int k_in, int k_out,
                        float* out_scores_dev, int64_t* out_ids_dev, void* stream) {
  if (nlists <= 0 || nq <= 0 || k_in <= 0 || k_out <= 0 || k_out > CRS_MAX_K) return fail(CRS_EINVAL, "bad sizes (k_out <= CRS_MAX_K)");
  if (!wire_dev || !out_scores_dev || !out_ids_dev) return fail(CRS_EINVAL, "null pointer");
  if ((uintptr_t)wire_dev & 7) return fail(CRS_EINVAL, "wire buffer must be 8-byte aligned");
  const int e = crs::merge_launch_wire(wire_dev, crs_wire_bytes(nq, k_in), crs_wire_scores_offset(nq, k_in), nlists, nq,
                                       k_in, k_out, out_scores_dev, out_ids_dev, (hipStream_t)stream);
  return e ? hip_fail((hipError_t)e, "merge launch") : CRS_OK;
}

int crs_scan_plan_describe(int nq, int dim, int k, int64_t n_rows, int slab_type, char* buf, size_t cap) {
  if (!buf || cap == 0) return fail(CRS_EINVAL, "null buffer");
  if (slab_type != CRS_SLAB_F16 && slab_type != CRS_SLAB_I8) return fail(CRS_EINVAL, "bad slab_type");
  Plan p;
  const int rc = make_plan(nq, dim, k, n_rows, slab_type, &p);
  if (rc) return rc;
  char name[96];
  if (p.w1_qg) snprintf(name, sizeof name, "scan_w2_kernel<%d> (%d queries/workgroup, dump)", p.pdim, p.w1_qg);
  else if (p.wide_nw) snprintf(name, sizeof name, "scan_wide_kernel<%d,%d,%d>", p.pdim, p.wide_nw, crs::scan_wide_slots(k));
  else if (p.tb_nw) snprintf(name, sizeof name, "scan_tb_kernel<%d,%d,%d,%d>", p.pdim, p.tile_rows, p.tb_nw, p.tb_slots);
  else if (slab_type == CRS_SLAB_I8) snprintf(name, sizeof name, "scan_i8_kernel<%d,%d,%d,%d>", p.pdim, p.tile_rows, (p.i8_tb || k <= 16) ? 16 : 32, p.i8_tb ? p.tb_slots : -1);
  else snprintf(name, sizeof name, "scan_f16_kernel<%d,%d,%d>", p.pdim, p.tile_rows, k <= 16 ? 16 : 32);
  snprintf(buf, cap, "%s streams=%d qblocks=%d kp=%d + merge%s", name, p.nwg, p.nqb, p.kp, p.group_best ? " + refine" : "");
  return CRS_OK;
}

int crs_stream_create_cu_masked(int first_cu, int n_cus, void** stream_out) {
  if (!stream_out || n_cus <= 0 || first_cu < 0) return fail(CRS_EINVAL, "bad CU range / null output");
  const int cus = device_cus();
  if (cus <= 0) return fail(CRS_EHIP, "no HIP device available%s");
  if (first_cu + n_cus > cus) return fail(CRS_EINVAL, "CU range exceeds the device");
  const int words = (cus + 31) / 32;
  uint32_t mask[16] = {0};
  if (words > 16) return fail(CRS_EINVAL, "device has more than 512 CUs");
  for (int c = first_cu; c < first_cu + n_cus; ++c) mask[c >> 5] |= 1u << (c & 31);
  hipStream_t st = nullptr;
  const hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask);
  if (e != hipSuccess) return hip_fail(e, "hipExtStreamCreateWithCUMask");
  *stream_out = (void*)st;
  return CRS_OK;
}

int crs_stream_destroy(void* stream) {
  if (!stream) return CRS_OK;
  const hipError_t e = hipStreamDestroy((hipStream_t)stream);
  return e == hipSuccess ? CRS_OK : hip_fail(e, "hipStreamDestroy");
}

int crs_time_cosine_topk(const void* q16_dev, int nq, int dim, int slab_type, const void* slab_dev,
                         const float* scales_dev, int64_t n_rows, int k, void* workspace_dev,
                         size_t workspace_bytes, float* out_scores_dev, int64_t* out_ids_dev,
                         void* stream, int iters, float* ms_total, float* ms_scan) {
  if (slab_type != CRS_SLAB_F16 && slab_type != CRS_SLAB_I8) return fail(CRS_EINVAL, "bad slab_type");
  Plan p;
  int rc = make_plan(nq, dim, k, n_rows, slab_type, &p);
  if (rc) return rc;
  if (iters <= 0 || !ms_total || !ms_scan) return fail(CRS_EINVAL, "bad iters / null outputs");
  if (workspace_bytes < ws_bytes(p.part_elems, nq, k)) return fail(CRS_ENOSPC, "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  hipEvent_t e0, e1;
  hipError_t he;
  if ((he = hipEventCreate(&e0)) != hipSuccess) return hip_fail(he, "hipEventCreate");
  if ((he = hipEventCreate(&e1)) != hipSuccess) return hip_fail(he, "hipEventCreate");
  float* ps;
  int* pr;
  // scan kernel alone
  hipEventRecord(e0, st);
  for (int i = 0; i < iters; ++i) {
    const int e = run_scan(p, q16_dev, nq, slab_type, slab_dev, scales_dev, n_rows, k, workspace_dev, st, &ps, &pr);
    if (e) { hipEventDestroy(e0); hipEventDestroy(e1); return e == -1 ? fail(CRS_EINVAL, "unsupported padded dimension") : hip_fail((hipError_t)e, "scan launch"); }
  }
  hipEventRecord(e1, st);
  if ((he = hipEventSynchronize(e1)) != hipSuccess) { hipEventDestroy(e0); hipEventDestroy(e1); return hip_fail(he, "scan timing"); }
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  *ms_scan = ms / iters;
  // scan + merge (what one crs_cosine_topk call costs on the device)
  hipEventRecord(e0, st);
  for (int i = 0; i < iters; ++i) {
    rc = crs_cosine_topk(q16_dev, nq, dim, slab_type, slab_dev, scales_dev, n_rows, k, 0, workspace_dev,
                         workspace_bytes, out_scores_dev, out_ids_dev, stream);
    if (rc) { hipEventDestroy(e0); hipEventDestroy(e1); return rc; }
  }
  hipEventRecord(e1, st);
  if ((he = hipEventSynchronize(e1)) != hipSuccess) { hipEventDestroy(e0); hipEventDestroy(e1); return hip_fail(he, "total timing"); }
  hipEventElapsedTime(&ms, e0, e1);
  *ms_total = ms / iters;
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return CRS_OK;
}

}  // extern "C"
