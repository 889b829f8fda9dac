// Internal declarations of the encoder translation units (enc_*.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace crs {

// Workgroup ids go round-robin over the 8 XCDs (id % 8), each with its own L2.  xcd_chunked_id turns the hardware id into
// a work index such that every XCD walks ONE contiguous range of indices (in id order): tiles that share an operand
// panel are neighbours in index space, so they meet in the same L2 instead of being fetched by eight of them.
__device__ __forceinline__ int xcd_chunked_id(int id, int total) {
  const int x = id & 7, q = total >> 3, r = total & 7;
  return x * q + (x < r ? x : r) + (id >> 3);
}

// enc_gemm.hip: C = epilogue(A[M,K] W[N,K]^T + bias); mode 0 fp16, 1 GELU fp16, 2 +residual fp32
int gemm_f16_launch(const _Float16* a, const _Float16* w, const float* bias, const float* residual,
                    void* out, int m, int n, int k, int mode, hipStream_t stream);

// enc_gemm_stream.hip: row-streaming variant for short contractions (K in {128, 256, 384, 512}) and many rows:
// W fragments resident in VGPRs, A streamed once per 128-column block
bool gemm_stream_supported(int k);
// enc_gemm8.hip: 256 x 256 x 64 tiles, eight-barrier phase schedule (whole tiles only, K % 128 == 0)
bool gemm8_applies(int m, int n, int k, int mode);
int gemm8_launch(const _Float16* a, const _Float16* w, const float* bias, const float* residual, void* out, int m, int n, int k,
                 int mode, hipStream_t stream);
int gemm8_splitk(int m, int n, int k);   // K slabs of the split-K form for this shape (0: not applicable)
int gemm8_splitk_launch(const _Float16* a, const _Float16* w, float* partials, int m, int n, int k, int splits, hipStream_t stream);
// enc_gemm_big.hip: 256-row tiles for large M (index build)
int gemm_big_block_n(int m, int n, int k, int mode);   // 0: not applicable
int gemm_big_launch(const _Float16* a, const _Float16* w, const float* bias, const float* residual, void* out, int m, int n, int k,
                    int mode, hipStream_t stream);
int gemm_stream_launch(const _Float16* a, const _Float16* w, const float* bias, const float* residual, void* out,
                       int m, int n, int k, int mode, hipStream_t stream);

// Panel variant for small M (one K chunk per workgroup, all loads issued at once; split-K over
// blockIdx.z with fp32 partials [k/kc][M][N] in mode 3, summed by layernorm_launch).
int gemm_panel_chunk(int k);
int gemm_panel_splits(int k, int m);   // fp32 partial slabs a mode-3 panel launch of m rows leaves for contraction length k
int gemm_panel_launch(const _Float16* a, const _Float16* w, const float* bias, void* out, int m, int n, int k,
                      int mode, int small_lds, hipStream_t stream);

// enc_qkvattn.hip: QKV projection + attention of short sequences (16 / 32 / 64 tokens) in one kernel
bool qkv_attn_supported(int hidden, int heads, int seq);
int qkv_attn_launch(const _Float16* x16, const _Float16* w_qkv, const float* b_qkv, const int* lens, _Float16* ctx, int batch,
                    int seq, int hidden, int heads, hipStream_t stream);

// enc_rowln.hip, pipelined variant for large token counts (index build), hidden = 384
bool gemm_rowln2_supported(int hidden, int k);
int gemm_rowln2_launch(const _Float16* a, const _Float16* w, const float* bias, const float* residual, const float* g,
                       const float* b, float eps, int m, int hidden, int k, float* x32, _Float16* x16, hipStream_t stream);

// enc_attn.hip: ctx[T, H] = softmax(QK^T / sqrt(hd) + padding mask) V per (batch, head);
// qkv is [T, 3H] fp16 (Q | K | V column blocks), lens[b] real tokens per row (right padding).
int attention_launch(const _Float16* qkv, const int* lens, _Float16* ctx, int batch, int seq, int hidden,
                     int heads, hipStream_t stream);

// enc_misc.hip
int embed_ln_launch(const int* ids, const float* word, const float* pos, const float* type0, const float* g,
                    const float* b, float eps, int tokens, int seq, int hidden, int vocab, float* x32,
                    _Float16* x16, hipStream_t stream);
// LayerNorm( sum_{s<nsplit} y[s] + bias + residual ): bias / residual may be null (already folded in)
int layernorm_launch(const float* y, int nsplit, const float* bias, const float* residual, const float* g,
                     const float* b, float eps, int tokens, int hidden, float* x32, _Float16* x16,
                     hipStream_t stream);
int pool_launch(const float* x32, const int* lens, int batch, int seq, int hidden, int pooling, int normalize,
                float* out, _Float16* out16, int pdim16, hipStream_t stream);

}  // namespace crs
