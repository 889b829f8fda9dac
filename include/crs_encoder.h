/* crs_encoder.h -- C ABI of the sentence-encoder forward in libcrs_hip.so (gfx950).
 *
 * Replaces what runs under `self.model.encode(texts, batch_size, normalize_embeddings=True,
 * convert_to_numpy=True)` at reference rag/embedding.py:65-71: the BertModel forward + Pooling +
 * Normalize of sentence-transformers (not vendored in the reference).  Tokenisation stays on the
 * host; this boundary starts at token ids.
 *
 * Numerics: GEMM operands fp16 (MFMA, fp32 accumulate); residual stream, LayerNorm (eps from the
 * descriptor, 1e-12 for BERT), softmax, GELU (erf form), pooling and the L2 normalisation in fp32.
 * Sequences are right-padded: `lens[b]` tokens of row b are real, the rest are padding.
 * All pointers are device pointers; `stream` is a hipStream_t as void*.  Returns 0 or a negative
 * CRS_E* code (crs_hip.h); message via crs_last_error().
 */
#ifndef CRS_ENCODER_H
#define CRS_ENCODER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRS_POOL_MEAN 0   /* all-MiniLM-L6-v2: mean over real tokens           */
#define CRS_POOL_CLS 1    /* bge-base-en-v1.5: hidden state of token 0 ([CLS]) */

typedef struct crs_encoder_desc {
  int32_t vocab_size;
  int32_t hidden;      /* H: 384 (MiniLM), 768 (bge-base); multiple of 64, <= 1024 */
  int32_t layers;
  int32_t heads;       /* head_dim = hidden / heads must be 16, 32 or 64          */
  int32_t ffn;         /* intermediate size, multiple of 64                       */
  int32_t max_pos;
  float ln_eps;
  int32_t pooling;     /* CRS_POOL_*                                              */
  int32_t flags;       /* CRS_ENC_* bits (ABI 3); 0 = default kernel selection     */
} crs_encoder_desc;

/* Kernel-form selection for query-batch forwards that run BESIDE a corpus scan (the role-lane layout of
 * rag/_engine.py): only kernel forms of <= 48 KB of LDS per workgroup, so that a forward's workgroups fit on CUs that
 * hold two scan workgroups (2 x 48 KB of 160 KB) instead of waiting for the scan to drain.  Same arithmetic, same
 * results; slower when the forward runs alone. */
#define CRS_ENC_SMALL_LDS 1

/* Per-layer device pointers.  Matrices are fp16 row-major [out_features, in_features] exactly as
 * torch.nn.Linear stores them (y = x W^T + b); vectors are fp32. */
typedef struct crs_encoder_layer {
  const void* w_qkv;   /* fp16 [3H, H]: query, key, value weights stacked */
  const float* b_qkv;  /* [3H] */
  const void* w_o;     /* fp16 [H, H] */
  const float* b_o;
  const float* ln1_g;  /* attention.output.LayerNorm */
  const float* ln1_b;
  const void* w_up;    /* fp16 [F, H]  intermediate.dense */
  const float* b_up;
  const void* w_down;  /* fp16 [H, F]  output.dense */
  const float* b_down;
  const float* ln2_g;  /* output.LayerNorm */
  const float* ln2_b;
} crs_encoder_layer;

typedef struct crs_encoder_weights {
  const float* word_emb;  /* fp32 [vocab, H] */
  const float* pos_emb;   /* fp32 [max_pos, H] */
  const float* type_emb;  /* fp32 [>=1, H]; row 0 is added to every token */
  const float* emb_ln_g;
  const float* emb_ln_b;
  const crs_encoder_layer* layers;  /* HOST array of `layers` entries holding device pointers */
} crs_encoder_weights;

int crs_encoder_workspace_bytes(const crs_encoder_desc* d, int batch, int seq, size_t* bytes);

/* ids_dev int32 [batch, seq]; lens_dev int32 [batch] (1 <= len <= seq <= max_pos);
 * out_dev fp32 [batch, H]: pooled (+ L2-normalised when normalize != 0) sentence embeddings.
 * hidden_out_dev (may be NULL): fp32 [batch, seq, H] final hidden states, for parity tests. */
int crs_encoder_forward(const crs_encoder_desc* d, const crs_encoder_weights* w, const int32_t* ids_dev,
                        const int32_t* lens_dev, int batch, int seq, void* workspace_dev,
                        size_t workspace_bytes, float* out_dev, int normalize, float* hidden_out_dev,
                        void* stream);

/* Query-side variant of crs_encoder_forward (always L2-normalised): besides the fp32 embeddings it writes
 * them as the fp16 query block crs_cosine_topk takes -- [batch, crs_row_elems(H, slab_type)], zero padded --
 * from the pooling kernel itself, which saves the separate crs_queries_to_f16 launch on the retrieve path
 * (rag/retrieval.py:113-121: embed the query, then search). */
int crs_encoder_forward_queries(const crs_encoder_desc* d, const crs_encoder_weights* w, const int32_t* ids_dev,
                                const int32_t* lens_dev, int batch, int seq, void* workspace_dev,
                                size_t workspace_bytes, float* out_dev, void* q16_out_dev, int slab_type,
                                void* stream);

/* Building block exported for parity tests and for users with their own layer stack:
 *   C[M, N] = epilogue(A[M, K] (fp16) x W[N, K]^T (fp16) + bias[N])
 *   mode 0: fp16 out;  mode 1: erf-GELU, fp16 out;  mode 2: + residual fp32 [M, N], fp32 out. */
int crs_gemm_f16(const void* a_dev, const void* w_dev, const float* bias_dev, const float* residual_dev,
                 void* out_dev, int m, int n, int k, int mode, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CRS_ENCODER_H */
