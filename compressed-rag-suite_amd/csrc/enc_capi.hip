// enc_capi.hip -- extern "C" surface of include/crs_encoder.h: the layer loop of the encoder.
//
// Per layer (7 launches on one stream, no host sync):
//   qkv  = x16 Wqkv^T + b                     gemm mode 0      [T, 3H] fp16
//   ctx  = attention(qkv, lens)               enc_attn.hip     [T, H]  fp16
//   y32  = ctx Wo^T + b + x32                 gemm mode 2      fp32
//   x    = LayerNorm(y32)                     -> x32 (fp32 residual stream), x16 (next GEMM input)
//   ffn  = gelu(x16 Wup^T + b)                gemm mode 1      [T, F] fp16
//   y32  = ffn Wdown^T + b + x32              gemm mode 2
//   x    = LayerNorm(y32)
// then pooling + L2 normalise.  The launch function allocates nothing and never synchronises, so
// a caller may capture it into a hipGraph for the launch-bound single-query case.
#include "../../include/crs_encoder.h"
#include "../../include/crs_hip.h"

#include <stdio.h>
#include <stdlib.h>

#include "enc.h"

namespace crs {
int set_error(int code, const char* msg);  // capi.hip
}

namespace {

size_t up256(size_t x) { return (x + 255) / 256 * 256; }

bool bigln_enabled() {   // CRS_ENC_BIGLN=0: tiled GEMM + separate LayerNorm on the index-build side (A/B runs)
  static int v = -1;
  if (v < 0) { const char* e = getenv("CRS_ENC_BIGLN"); v = (e && e[0] == '0') ? 0 : 1; }
  return v == 1;
}

bool qa_enabled() {   // CRS_ENC_QKVATTN=0: separate QKV GEMM and attention launches (A/B runs, tests)
  static int v = -1;
  if (v < 0) { const char* e = getenv("CRS_ENC_QKVATTN"); v = (e && e[0] == '0') ? 0 : 1; }
  return v == 1;
}

struct Layout {
  size_t x32, y32, x16, ctx, qkv, ffn, total;
  int max_split;
};

// Small token counts are latency-bound: use the one-shot panel GEMM (+ split-K partials reduced in
// the LayerNorm); large ones (index build) use the pipelined 128 x 128 kernel.
constexpr int kPanelMaxTokens = 4096;
// split-K panels (fp32 partials summed by the LayerNorm): round 1 measured them losing at 4096 tokens of bge (C3 step 2.79 ms
// against 2.05 through the tiled kernel's fused epilogue) -- with one-shot 384-column staging and 2 / 8 slabs.  With the
// workgroups walking their K range in 128-column pieces and the slab count capped by the row count (gemm_panel_splits) a
// SINGLE forward at 4096 tokens is faster through split-K panels (bge-base 256 x 16 tokens: 1996 -> 1855 us, one stream), but
// with eight batches in flight the extra slab traffic costs more than the latency it saves (C3: 2.21 against 2.11 ms per
// batch, tools/ab_c3.sh) -- so the limit stays at 2048 tokens (bge-base 128 x 16: 1500 -> 1140 us); CRS_SPLITK_MAX_TOKENS
// moves it
constexpr int kSplitKMaxTokens = 2048;
bool use_panel(int tokens, int k) {
  const int kc = crs::gemm_panel_chunk(k);
  if (tokens > kPanelMaxTokens || kc == 0) return false;
  static int splitk_max_tokens = -1;   // CRS_SPLITK_MAX_TOKENS: A/B runs
  if (splitk_max_tokens < 0) { const char* e = getenv("CRS_SPLITK_MAX_TOKENS"); splitk_max_tokens = e ? atoi(e) : kSplitKMaxTokens; }
  if (k / kc > 1 && tokens > splitk_max_tokens) return false;
  const int ns = crs::gemm_panel_splits(k, tokens);   // slab counts the LayerNorm kernel is instantiated for
  return ns == 1 || ns == 2 || ns == 3 || ns == 4 || ns == 6 || ns == 8;
}

Layout make_layout(const crs_encoder_desc* d, int batch, int seq) {
  const size_t t = (size_t)batch * seq, h = d->hidden, f = d->ffn;
  Layout l;
  size_t off = 0;
  int split = 1;
  if (use_panel((int)t, (int)f)) split = crs::gemm_panel_splits((int)f, (int)t);
  if (use_panel((int)t, (int)h) && crs::gemm_panel_splits((int)h, (int)t) > split) split = crs::gemm_panel_splits((int)h, (int)t);
  if (crs::gemm8_splitk((int)t, (int)h, (int)f) > split) split = crs::gemm8_splitk((int)t, (int)h, (int)f);
  if (crs::gemm8_splitk((int)t, (int)h, (int)h) > split) split = crs::gemm8_splitk((int)t, (int)h, (int)h);
  l.max_split = split;
  l.x32 = off; off += up256(t * h * 4);
  l.y32 = off; off += up256(t * h * 4 * split);
  l.x16 = off; off += up256(t * h * 2);
  l.ctx = off; off += up256(t * h * 2);
  l.qkv = off; off += up256(t * 3 * h * 2);
  l.ffn = off; off += up256(t * f * 2);
  l.total = off;
  return l;
}

int check_desc(const crs_encoder_desc* d) {
  if (!d) return crs::set_error(CRS_EINVAL, "null descriptor");
  if (d->hidden <= 0 || d->hidden > 1024 || d->hidden % 64) return crs::set_error(CRS_EINVAL, "hidden must be a multiple of 64, <= 1024");
  if (d->heads <= 0 || d->hidden % d->heads) return crs::set_error(CRS_EINVAL, "hidden must divide by heads");
  const int hd = d->hidden / d->heads;
  if (hd != 16 && hd != 32 && hd != 64) return crs::set_error(CRS_EINVAL, "head_dim must be 16, 32 or 64");
  if (d->ffn <= 0 || d->ffn % 64) return crs::set_error(CRS_EINVAL, "ffn must be a multiple of 64");
  if (d->layers <= 0 || d->vocab_size <= 0 || d->max_pos <= 0) return crs::set_error(CRS_EINVAL, "bad layers/vocab/max_pos");
  if (d->pooling != CRS_POOL_MEAN && d->pooling != CRS_POOL_CLS) return crs::set_error(CRS_EINVAL, "bad pooling mode");
  return CRS_OK;
}

#define CRS_TRY(expr, what)                                                             \
  do {                                                                                  \
    const int e_ = (expr);                                                              \
    if (e_ == -1) return crs::set_error(CRS_EINVAL, what ": unsupported shape");        \
    if (e_) { char m_[160]; snprintf(m_, sizeof m_, what ": %s", hipGetErrorString((hipError_t)e_)); return crs::set_error(CRS_EHIP, m_); } \
  } while (0)

}  // namespace

extern "C" {

int crs_encoder_workspace_bytes(const crs_encoder_desc* d, int batch, int seq, size_t* bytes) {
  const int rc = check_desc(d);
  if (rc) return rc;
  if (!bytes || batch <= 0 || seq <= 0 || seq > d->max_pos) return crs::set_error(CRS_EINVAL, "bad batch/seq (seq <= max_pos)");
  *bytes = make_layout(d, batch, seq).total;
  return CRS_OK;
}

int crs_gemm_f16(const void* a_dev, const void* w_dev, const float* bias_dev, const float* residual_dev,
                 void* out_dev, int m, int n, int k, int mode, void* stream) {
  if (!a_dev || !w_dev || !out_dev || m <= 0 || n <= 0 || k <= 0) return crs::set_error(CRS_EINVAL, "bad gemm arguments");
  if (k % 64) return crs::set_error(CRS_EINVAL, "gemm K must be a multiple of 64");
  if (mode < 0 || mode > 2 || (mode == 2 && !residual_dev)) return crs::set_error(CRS_EINVAL, "bad gemm mode / missing residual");
  CRS_TRY(crs::gemm_f16_launch((const _Float16*)a_dev, (const _Float16*)w_dev, bias_dev, residual_dev, out_dev, m, n,
                               k, mode, (hipStream_t)stream), "gemm");
  return CRS_OK;
}

static int encoder_forward(const crs_encoder_desc* d, const crs_encoder_weights* w, const int32_t* ids_dev,
                           const int32_t* lens_dev, int batch, int seq, void* workspace_dev,
                           size_t workspace_bytes, float* out_dev, int normalize, float* hidden_out_dev,
                           _Float16* q16_out_dev, int q16_row_elems, void* stream) {
  const int rc = check_desc(d);
  if (rc) return rc;
  if (!w || !w->layers || !ids_dev || !lens_dev || !workspace_dev || !out_dev) return crs::set_error(CRS_EINVAL, "null pointer");
  if (batch <= 0 || seq <= 0 || seq > d->max_pos) return crs::set_error(CRS_EINVAL, "bad batch/seq (seq <= max_pos)");
  const Layout l = make_layout(d, batch, seq);
  if (workspace_bytes < l.total) return crs::set_error(CRS_ENOSPC, "encoder workspace too small");
  hipStream_t st = (hipStream_t)stream;
  char* ws = reinterpret_cast<char*>(workspace_dev);
  float* x32 = reinterpret_cast<float*>(ws + l.x32);
  float* y32 = reinterpret_cast<float*>(ws + l.y32);
  _Float16* x16 = reinterpret_cast<_Float16*>(ws + l.x16);
  _Float16* ctx = reinterpret_cast<_Float16*>(ws + l.ctx);
  _Float16* qkv = reinterpret_cast<_Float16*>(ws + l.qkv);
  _Float16* ffn = reinterpret_cast<_Float16*>(ws + l.ffn);
  const int T = batch * seq, H = d->hidden, F = d->ffn;

  CRS_TRY(crs::embed_ln_launch(ids_dev, w->word_emb, w->pos_emb, w->type_emb, w->emb_ln_g, w->emb_ln_b, d->ln_eps, T,
                               seq, H, d->vocab_size, x32, x16, st), "embed_ln");
  const bool panel_h = use_panel(T, H), panel_f = use_panel(T, F);
  // index-build side (large token counts), hidden = 384: projection + bias + residual + LayerNorm in one pipelined kernel
  const bool big_ln = T > kPanelMaxTokens && bigln_enabled();
  const bool big_ln_h = big_ln && crs::gemm_rowln2_supported(H, H), big_ln_f = big_ln && crs::gemm_rowln2_supported(H, F);
  // fp16-epilogue projections (QKV, FFN-up) on the panel kernel: K = H in one chunk, or (CRS_ENC_PANEL_MULTI != 0) walked in
  // chunks by the workgroup -- bge-base at query-batch sizes, where the row-streaming kernel pays a 196 KB weight prologue
  // per workgroup for a handful of tiles
  static int panel_multi = -1;
  if (panel_multi < 0) { const char* e = getenv("CRS_ENC_PANEL_MULTI"); panel_multi = (e && e[0] == '0') ? 0 : 1; }
  // (the phase-scheduled 256 x 256 kernel takes QKV / FFN-up as soon as it has a chip's worth of tiles: bge-base from 4096 tokens)
  const bool single_h = T <= kPanelMaxTokens && crs::gemm_panel_chunk(H) != 0 && (crs::gemm_panel_chunk(H) == H || panel_multi) &&
                        !crs::gemm8_applies(T, 3 * H, H, 0);
  const int s8_h = (!panel_h) ? crs::gemm8_splitk(T, H, H) : 0, s8_f = (!panel_f) ? crs::gemm8_splitk(T, H, F) : 0;
  // short sequences in the launch-bound regime: QKV projection + attention as one kernel (enc_qkvattn.hip)
  const int small = (d->flags & CRS_ENC_SMALL_LDS) ? 1 : 0;
  const bool fuse_qa = T <= kPanelMaxTokens && !small && qa_enabled() && crs::qkv_attn_supported(H, d->heads, seq);
  for (int li = 0; li < d->layers; ++li) {
    const crs_encoder_layer& L = w->layers[li];
    if (fuse_qa) {
      CRS_TRY(crs::qkv_attn_launch(x16, (const _Float16*)L.w_qkv, L.b_qkv, lens_dev, ctx, batch, seq, H, d->heads, st), "qkv + attention");
    } else {
    if (single_h) CRS_TRY(crs::gemm_panel_launch(x16, (const _Float16*)L.w_qkv, L.b_qkv, qkv, T, 3 * H, H, 0, small, st), "qkv gemm");
    else CRS_TRY(crs::gemm_f16_launch(x16, (const _Float16*)L.w_qkv, L.b_qkv, nullptr, qkv, T, 3 * H, H, 0, st), "qkv gemm");
    CRS_TRY(crs::attention_launch(qkv, lens_dev, ctx, batch, seq, H, d->heads, st), "attention");
    }
    if (big_ln_h) {
      CRS_TRY(crs::gemm_rowln2_launch(ctx, (const _Float16*)L.w_o, L.b_o, x32, L.ln1_g, L.ln1_b, d->ln_eps, T, H, H, x32, x16, st), "out projection + layernorm 1");
    } else if (panel_h) {
      CRS_TRY(crs::gemm_panel_launch(ctx, (const _Float16*)L.w_o, nullptr, y32, T, H, H, 3, small, st), "out gemm");
      CRS_TRY(crs::layernorm_launch(y32, crs::gemm_panel_splits(H, T), L.b_o, x32, L.ln1_g, L.ln1_b, d->ln_eps, T, H, x32, x16, st), "layernorm 1");
    } else if (s8_h) {
      CRS_TRY(crs::gemm8_splitk_launch(ctx, (const _Float16*)L.w_o, y32, T, H, H, s8_h, st), "out gemm (split-K)");
      CRS_TRY(crs::layernorm_launch(y32, s8_h, L.b_o, x32, L.ln1_g, L.ln1_b, d->ln_eps, T, H, x32, x16, st), "layernorm 1");
    } else {
      CRS_TRY(crs::gemm_f16_launch(ctx, (const _Float16*)L.w_o, L.b_o, x32, y32, T, H, H, 2, st), "out gemm");
      CRS_TRY(crs::layernorm_launch(y32, 1, nullptr, nullptr, L.ln1_g, L.ln1_b, d->ln_eps, T, H, x32, x16, st), "layernorm 1");
    }
    if (single_h) CRS_TRY(crs::gemm_panel_launch(x16, (const _Float16*)L.w_up, L.b_up, ffn, T, F, H, 1, small, st), "ffn up gemm");
    else CRS_TRY(crs::gemm_f16_launch(x16, (const _Float16*)L.w_up, L.b_up, nullptr, ffn, T, F, H, 1, st), "ffn up gemm");
    if (big_ln_f) {
      CRS_TRY(crs::gemm_rowln2_launch(ffn, (const _Float16*)L.w_down, L.b_down, x32, L.ln2_g, L.ln2_b, d->ln_eps, T, H, F, x32, x16, st), "ffn down projection + layernorm 2");
    } else if (panel_f) {
      CRS_TRY(crs::gemm_panel_launch(ffn, (const _Float16*)L.w_down, nullptr, y32, T, H, F, 3, small, st), "ffn down gemm");
      CRS_TRY(crs::layernorm_launch(y32, crs::gemm_panel_splits(F, T), L.b_down, x32, L.ln2_g, L.ln2_b, d->ln_eps, T, H, x32, x16, st), "layernorm 2");
    } else if (s8_f) {
      CRS_TRY(crs::gemm8_splitk_launch(ffn, (const _Float16*)L.w_down, y32, T, H, F, s8_f, st), "ffn down gemm (split-K)");
      CRS_TRY(crs::layernorm_launch(y32, s8_f, L.b_down, x32, L.ln2_g, L.ln2_b, d->ln_eps, T, H, x32, x16, st), "layernorm 2");
    } else {
      CRS_TRY(crs::gemm_f16_launch(ffn, (const _Float16*)L.w_down, L.b_down, x32, y32, T, H, F, 2, st), "ffn down gemm");
      CRS_TRY(crs::layernorm_launch(y32, 1, nullptr, nullptr, L.ln2_g, L.ln2_b, d->ln_eps, T, H, x32, x16, st), "layernorm 2");
    }
  }
  if (hidden_out_dev) {
    const hipError_t e = hipMemcpyAsync(hidden_out_dev, x32, (size_t)T * H * 4, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return crs::set_error(CRS_EHIP, hipGetErrorString(e));
  }
  CRS_TRY(crs::pool_launch(x32, lens_dev, batch, seq, H, d->pooling, normalize, out_dev, q16_out_dev, q16_row_elems, st), "pool");
  return CRS_OK;
}

int crs_encoder_forward(const crs_encoder_desc* d, const crs_encoder_weights* w, const int32_t* ids_dev,
                        const int32_t* lens_dev, int batch, int seq, void* workspace_dev,
                        size_t workspace_bytes, float* out_dev, int normalize, float* hidden_out_dev,
                        void* stream) {
  return encoder_forward(d, w, ids_dev, lens_dev, batch, seq, workspace_dev, workspace_bytes, out_dev, normalize,
                         hidden_out_dev, nullptr, 0, stream);
}

int crs_encoder_forward_queries(const crs_encoder_desc* d, const crs_encoder_weights* w, const int32_t* ids_dev,
                                const int32_t* lens_dev, int batch, int seq, void* workspace_dev,
                                size_t workspace_bytes, float* out_dev, void* q16_out_dev, int slab_type,
                                void* stream) {
  if (!q16_out_dev) return crs::set_error(CRS_EINVAL, "null pointer");
  if (slab_type != CRS_SLAB_F16 && slab_type != CRS_SLAB_I8) return crs::set_error(CRS_EINVAL, "bad slab_type");
  if (!d) return crs::set_error(CRS_EINVAL, "null descriptor");
  return encoder_forward(d, w, ids_dev, lens_dev, batch, seq, workspace_dev, workspace_bytes, out_dev, 1, nullptr,
                         reinterpret_cast<_Float16*>(q16_out_dev), crs_row_elems(d->hidden, slab_type), stream);
}

}  // extern "C"
