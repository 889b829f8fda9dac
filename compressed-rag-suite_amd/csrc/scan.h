// Internal (C++) declarations shared by the HIP translation units of libcrs_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace crs {

struct ScanArgs {
  const _Float16* q;      // [nq, D] fp16, zero padded to D
  const void* slab;       // [n_rows, D] fp16 (or int8)
  const float* scales;    // int8 slabs: one fp32 per row, else nullptr
  float* part_scores;     // [nq, nwg, kp]
  int* part_rows;         // [nq, nwg, kp] local row index (tile-best kernels: first row of the tile), -1 = empty
  unsigned* tau_shared;   // [nq] order-preserving uint of the best published k-th score (0 = none); may be null
  unsigned long long* stamps;  // diagnostics only (tools/scan_probe); nullptr in the product path
  int n_rows;
  int n_tiles;
  int nq;
  int k;
  int sched;              // synchronous-compaction schedule: 0 none, 1 {3,4,6,8,12,...}, 2 {4,8,16,...}
  int boot;               // 1: bootstrap the threshold from the first 64 rows in registers (scan.hip)
  int kp;                 // slots per (query, workgroup) partial list (capi.hip make_plan: k, tiles per stream, or chain slots)
  int nwg;                // tile streams (workgroups per query block)
  int nqb;                // query blocks (64, 128 or 256 queries each); the grid is nqb * nwg workgroups (scan_common.h: grid mapping)
  unsigned* ticket;       // scan_tb.hip chain mode: tiles >= t_dyn are handed out through this counter (zero at launch); else nullptr
  int t_dyn;              //   first dynamically scheduled tile (a multiple of nwg, >= 2 nwg); n_tiles when the schedule is static
  int dyn_mask;           //   a ticket stands for dyn_mask + 1 consecutive tiles (a power of two)
};

// scan_refine.hip: re-open the k winning tiles (16, 32 or 64 rows each) per query, re-score, rank
int refine_launch(const _Float16* q16, int nq, int pdim, const _Float16* slab, int n_rows, const float* win_s,
                  const int64_t* win, int k, int tile_rows, int64_t id_base, float* out_s, int64_t* out_i, hipStream_t stream);

int refine_i8_launch(const _Float16* q16, int nq, int pdim, const void* slab, const float* scales, int n_rows, const float* win_s,
                     const int64_t* win, int k, int tile_rows, int64_t id_base, float* out_s, int64_t* out_i, hipStream_t stream);

int scan_tile_rows(int pdim);
int scan_i8_tile_rows();
int scan_wg_per_cu();    // resident workgroups per CU the active scan.hip variant is launched with
bool scan_share_tau();  // CRS_SCAN_SHARE_TAU=1 enables cross-workgroup threshold sharing (scan.hip; measured slower, off)
// returns hipError_t as int, -1 for an unsupported padded dimension
int scan_launch_f16(const ScanArgs& a, int pdim, int nwg, hipStream_t stream);
// slots: -1 threshold kernel; 0 tile-best dump; 4 / 10 / 16 tile-best chain (finished by merge + refine_i8_launch)
int scan_launch_i8(const ScanArgs& a, int pdim, int slots, hipStream_t stream);
// scan_tb.hip: tile-best 16x16x32 scan, k <= 16; nw = 4 (64 queries / workgroup) or 8 (128);
// slots = 0: dump mode (kp = tiles per stream), else chain mode (kp = slots); finished by merge + refine_launch
int scan_tb_wg_per_cu(int pdim, int nw);
bool scan_tb_has_8_waves(int pdim);
int scan_launch_tb(const ScanArgs& a, int pdim, int nw, int slots, hipStream_t stream);
int scan_ticket_zero(unsigned* ticket, hipStream_t stream);   // ScanArgs::ticket := 0, in stream order (scan_tb.hip / scan_i8.hip chain modes)
int scan_tb_long_chain_slots(int pdim, int nw, int k);   // chain length for 16 < k <= 64 on long streams (0: none)
int scan_i8_long_chain_slots(int pdim, int k);
// scan_wide.hip: 65+ queries per launch, k <= 16, fp16 slabs
int scan_wide_waves(int nq, int k, int pdim);
int scan_wide_wg_per_cu(int nw, int pdim);
int scan_wide_slots(int k);
int scan_wide_tile_rows(int nw, int pdim);
int scan_launch_wide(const ScanArgs& a, int pdim, int nw, hipStream_t stream);

// scan_w1.hip: > 64 queries per launch on 768-element fp16 rows, k <= 64: 256 queries per workgroup, dump selection
int scan_w1_queries_per_wg(int nq, int k, int pdim);   // 0: not applicable; else queries per workgroup (128 or 256)
int scan_launch_w1(const ScanArgs& a, int pdim, hipStream_t stream);

// merge.hip
// inter_s / inter_i: [nq, merge_slices(nlists, k_in), k_out] scratch of the two-level form (> 8192 candidates per query);
// null = always one level
int merge_slices(int nlists, int k_in);
int merge_launch_i32(const float* scores, const int* rows, int nlists, int nq, int k_in, int k_out,
                     int64_t id_base, float* out_scores, int64_t* out_ids, float* inter_s, int64_t* inter_i, hipStream_t stream);
int merge_launch_i64(const float* scores, const int64_t* ids, int nlists, int nq, int k_in, int k_out,
                     float* out_scores, int64_t* out_ids, hipStream_t stream);

int merge_launch_wire(const void* wire, size_t block_bytes, size_t scores_off, int nlists, int nq, int k_in, int k_out,
                      float* out_scores, int64_t* out_ids, hipStream_t stream);

// convert.hip
int refine_f32_launch(const float* q32, int nq, int dim, const float* shadow, int64_t n_rows, int64_t id_base,
                      const int64_t* cand, int k_in, int k_out, float* out_s, int64_t* out_i, hipStream_t stream);
int slab_append_launch(const float* emb, int64_t n, int dim, int pdim, int slab_type, void* slab,
                       float* scales, float* shadow, int64_t row0, float* row_err_max, hipStream_t stream);
int queries_to_f16_launch(const float* q, int nq, int dim, int pdim, _Float16* out, hipStream_t stream);
int score_rows_launch(const float* q32, int nq, int dim, const float* shadow, int64_t n_rows, int64_t id_base, int k, const int64_t* ids,
                      float* scores, hipStream_t stream);
int rescore_launch(const float* q32, int nq, int dim, const float* shadow, int64_t n_rows,
                   int64_t id_base, int k, float* scores, int64_t* ids, hipStream_t stream);

// exact.hip: exactness certificate of the over-fetch re-rank + in-stream escalation of uncertified queries
float exact_err_arith(int dim, int pdim);
float exact_err_rows_bound(int dim, int slab_type);
int refine_cert_launch(const float* q32, const _Float16* q16, int nq, int dim, int pdim, int slab_type, const float* shadow,
                       int64_t n_rows, int64_t id_base, const int64_t* cand, const float* cand_s, int k_in, int k_out,
                       float err_rows, float* out_s, int64_t* out_i, int* status, float* ws_thr, int* ws_cnt, int* ws_done, hipStream_t stream);
int escalate_launch(const float* q32, const _Float16* q16, int nq, int dim, int pdim, int slab_type, const void* slab,
                    const float* scales, const float* shadow, int64_t n_rows, int64_t id_base, int k_out, float* out_s,
                    int64_t* out_i, int* status, const float* ws_thr, int* ws_cnt, int* ws_done, int64_t* ws_lists, int cap, int cus,
                    hipStream_t stream);

}  // namespace crs
