// torch_ops.cpp -- PyTorch-ROCm custom ops over the C ABI of libcrs_hip.so (host-only C++, no device code).
//
// north_star / SURVEY section 8(b): "Python host code invokes hand-written HIP kernels via PyTorch-ROCm custom
// ops".  TORCH_LIBRARY(crs, ...) registers, for the HIP ("CUDA" dispatch key on torch-ROCm) backend:
//   crs::encoder_forward   replaces SentenceTransformer.encode's forward      (reference rag/embedding.py:65-71)
//   crs::slab_append       replaces collection.add(embeddings=...)            (reference rag/indexing.py:114-119)
//   crs::queries_to_f16    query side of the same conversion                  (reference rag/indexing.py:156-168)
//   crs::cosine_topk       replaces collection.query(query_embeddings, n)     (reference rag/indexing.py:171-176)
//   crs::refine_f32        over-fetch re-rank against the fp32 shadow         
//   crs::refine_f32_cert / crs::escalate_exact   the same with a per-query exactness proof, and the in-stream
//                          escalation of unproven queries (identical ids to an fp32 store: rag/indexing.py:171-176)
//   crs::merge_topk / crs::merge_topk_wire   cross-shard merge                (new vs the reference)
// Tensors are torch-owned; every op launches on the CURRENT HIP stream of the tensors' device, so the ops
// compose with torch streams and hipGraph capture.  Errors of the C ABI surface as RuntimeError (TORCH_CHECK)
// carrying crs_last_error(); the Python wrappers (rag/_native.py) translate where the reference's types differ.
// The C ABI (include/crs_hip.h, include/crs_encoder.h) stays the drop-in boundary for non-torch hosts
// (INTEGRATION.md section 2); this file only adapts it.
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>   // torch-ROCm tensors carry DeviceType "cuda": the masquerading guard/stream
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/library.h>

#include <tuple>
#include <vector>

#include "../../include/crs_encoder.h"
#include "../../include/crs_hip.h"

namespace {

using at::Tensor;

void* cur_stream(const Tensor& t) { return (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream(); }

void ok(int rc, const char* what) { TORCH_CHECK(rc == 0, what, ": libcrs_hip error ", rc, ": ", crs_last_error()); }

void want(const Tensor& t, at::ScalarType ty, const char* name) {
  TORCH_CHECK(t.is_cuda(), name, " must be a device (HIP) tensor");
  TORCH_CHECK(t.scalar_type() == ty, name, " has dtype ", t.scalar_type(), ", expected ", ty);
  TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");
}
const void* opt_ptr(const c10::optional<Tensor>& t) { return (t.has_value() && t->defined()) ? t->data_ptr() : nullptr; }
bool has(const c10::optional<Tensor>& t) { return t.has_value() && t->defined(); }

// every tensor of a call must live on the device whose stream and guard the op uses: with a one-process multi-device
// store a mismatched tensor would otherwise be a GPU memory fault (or a silent wrong-device read), not an exception
void same_device(const Tensor& ref, std::initializer_list<const Tensor*> ts, const char* op) {
  for (const Tensor* t : ts)
    if (t && t->defined()) TORCH_CHECK(t->device() == ref.device(), op, ": all tensors must be on ", ref.device(), ", got one on ", t->device());
}
const Tensor* opt_t(const c10::optional<Tensor>& t) { return has(t) ? &*t : nullptr; }

int slab_type_of(const Tensor& slab) {
  TORCH_CHECK(slab.scalar_type() == at::kHalf || slab.scalar_type() == at::kChar, "slab must be fp16 or int8");
  return slab.scalar_type() == at::kChar ? CRS_SLAB_I8 : CRS_SLAB_F16;
}

// ---- index build ---------------------------------------------------------------------------------------------
void slab_append(const Tensor& emb, Tensor slab, c10::optional<Tensor> scales, c10::optional<Tensor> shadow, int64_t row0,
                 c10::optional<Tensor> row_err) {
  want(emb, at::kFloat, "emb");
  same_device(emb, {&slab, opt_t(scales), opt_t(shadow), opt_t(row_err)}, "crs::slab_append");
  if (has(row_err)) { want(*row_err, at::kFloat, "row_err"); TORCH_CHECK(row_err->numel() >= 1, "row_err must hold one fp32"); }
  TORCH_CHECK(emb.dim() == 2 && slab.dim() == 2 && slab.is_cuda() && slab.is_contiguous(), "emb [n, dim], slab [cap, pdim]");
  const int st = slab_type_of(slab);
  const int64_t n = emb.size(0);
  const int dim = (int)emb.size(1);
  TORCH_CHECK(slab.size(1) == crs_row_elems(dim, st), "slab row length must be crs_row_elems(dim, slab_type)");
  TORCH_CHECK(row0 >= 0 && row0 + n <= slab.size(0), "rows [row0, row0 + n) exceed the slab");
  if (st == CRS_SLAB_I8) {
    TORCH_CHECK(scales.has_value(), "int8 slab needs scales");
    want(*scales, at::kFloat, "scales");
    TORCH_CHECK(scales->numel() >= row0 + n, "scales too short");
  }
  if (shadow.has_value() && shadow->defined()) {
    want(*shadow, at::kFloat, "shadow");
    TORCH_CHECK(shadow->dim() == 2 && shadow->size(1) == dim && shadow->size(0) >= row0 + n, "shadow must be fp32 [>= row0 + n, dim]");
  }
  c10::hip::HIPGuardMasqueradingAsCUDA g(emb.device());
  ok(crs_slab_append_f32(emb.data_ptr<float>(), n, dim, st, slab.data_ptr(), (float*)opt_ptr(scales), (float*)opt_ptr(shadow), row0,
                         (float*)opt_ptr(row_err), cur_stream(emb)), "crs::slab_append");
}

void queries_to_f16(const Tensor& q32, Tensor out16, int64_t slab_type) {
  want(q32, at::kFloat, "q32");
  want(out16, at::kHalf, "out16");
  same_device(q32, {&out16}, "crs::queries_to_f16");
  TORCH_CHECK(q32.dim() == 2 && out16.dim() == 2 && out16.size(0) == q32.size(0) &&
                  out16.size(1) == crs_row_elems((int)q32.size(1), (int)slab_type), "out16 must be [nq, crs_row_elems(dim, slab_type)]");
  c10::hip::HIPGuardMasqueradingAsCUDA g(q32.device());
  ok(crs_queries_to_f16(q32.data_ptr<float>(), (int)q32.size(0), (int)q32.size(1), (int)slab_type, out16.data_ptr(), cur_stream(q32)),
     "crs::queries_to_f16");
}

// ---- search --------------------------------------------------------------------------------------------------
void cosine_topk_out(const Tensor& q16, const Tensor& slab, c10::optional<Tensor> scales, int64_t n_rows, int64_t dim, int64_t k,
                     int64_t id_base, Tensor workspace, Tensor out_scores, Tensor out_ids) {
  want(q16, at::kHalf, "q16");
  TORCH_CHECK(slab.is_cuda() && slab.is_contiguous() && slab.dim() == 2 && q16.dim() == 2, "q16 [nq, pdim], slab [rows, pdim]");
  const int st = slab_type_of(slab);
  const int pdim = crs_row_elems((int)dim, st);
  TORCH_CHECK(q16.size(1) == pdim && slab.size(1) == pdim, "q16 / slab row length must be crs_row_elems(dim, slab_type) = ", pdim);
  TORCH_CHECK(n_rows >= 1 && n_rows <= slab.size(0), "n_rows out of range");
  want(out_scores, at::kFloat, "out_scores");
  want(out_ids, at::kLong, "out_ids");
  const int64_t nq = q16.size(0);
  TORCH_CHECK(out_scores.numel() == nq * k && out_ids.numel() == nq * k, "outputs must hold [nq, k]");
  TORCH_CHECK(workspace.is_cuda() && workspace.is_contiguous(), "workspace must be a contiguous device tensor");
  if (st == CRS_SLAB_I8) {
    TORCH_CHECK(has(scales), "int8 slab needs scales");
    want(*scales, at::kFloat, "scales");
    TORCH_CHECK(scales->numel() >= n_rows, "scales shorter than n_rows");
  }
  same_device(q16, {&slab, opt_t(scales), &workspace, &out_scores, &out_ids}, "crs::cosine_topk");
  c10::hip::HIPGuardMasqueradingAsCUDA g(q16.device());
  ok(crs_cosine_topk(q16.data_ptr(), (int)nq, (int)dim, st, slab.data_ptr(), (const float*)opt_ptr(scales), n_rows, (int)k, id_base,
                     workspace.data_ptr(), (size_t)workspace.nbytes(), out_scores.data_ptr<float>(), out_ids.data_ptr<int64_t>(),
                     cur_stream(q16)), "crs::cosine_topk");
}

std::tuple<Tensor, Tensor> cosine_topk(const Tensor& q16, const Tensor& slab, c10::optional<Tensor> scales, int64_t n_rows, int64_t dim,
                                       int64_t k, int64_t id_base) {
  TORCH_CHECK(q16.dim() == 2, "q16 must be [nq, pdim]");
  size_t need = 0;
  ok(crs_scan_workspace_bytes((int)q16.size(0), (int)dim, (int)k, n_rows, &need), "crs::cosine_topk (workspace)");
  c10::hip::HIPGuardMasqueradingAsCUDA g(q16.device());
  Tensor ws = at::empty({(int64_t)need}, q16.options().dtype(at::kByte));
  Tensor s = at::empty({q16.size(0), k}, q16.options().dtype(at::kFloat));
  Tensor i = at::empty({q16.size(0), k}, q16.options().dtype(at::kLong));
  cosine_topk_out(q16, slab, scales, n_rows, dim, k, id_base, ws, s, i);
  return {s, i};
}

void refine_f32_out(const Tensor& q32, const Tensor& shadow, int64_t n_rows, int64_t id_base, const Tensor& cand_ids, int64_t k_out,
                    Tensor out_scores, Tensor out_ids) {
  want(q32, at::kFloat, "q32");
  want(shadow, at::kFloat, "shadow");
  want(cand_ids, at::kLong, "cand_ids");
  want(out_scores, at::kFloat, "out_scores");
  want(out_ids, at::kLong, "out_ids");
  TORCH_CHECK(q32.dim() == 2 && shadow.dim() == 2 && cand_ids.dim() == 2 && shadow.size(1) == q32.size(1) &&
                  cand_ids.size(0) == q32.size(0) && n_rows <= shadow.size(0), "q32 [nq, dim], shadow [>= n_rows, dim], cand_ids [nq, k_in]");
  TORCH_CHECK(out_scores.numel() == q32.size(0) * k_out && out_ids.numel() == q32.size(0) * k_out, "outputs must hold [nq, k_out]");
  TORCH_CHECK(cand_ids.size(1) >= k_out, "cand_ids holds fewer than k_out candidates per query");
  same_device(q32, {&shadow, &cand_ids, &out_scores, &out_ids}, "crs::refine_f32");
  c10::hip::HIPGuardMasqueradingAsCUDA g(q32.device());
  ok(crs_refine_f32(q32.data_ptr<float>(), (int)q32.size(0), (int)q32.size(1), shadow.data_ptr<float>(), n_rows, id_base,
                    cand_ids.data_ptr<int64_t>(), (int)cand_ids.size(1), (int)k_out, out_scores.data_ptr<float>(),
                    out_ids.data_ptr<int64_t>(), cur_stream(q32)), "crs::refine_f32");
}

void score_rows_f32_out(const Tensor& q32, const Tensor& shadow, int64_t n_rows, int64_t id_base, const Tensor& ids, Tensor scores) {
  want(q32, at::kFloat, "q32");
  want(shadow, at::kFloat, "shadow");
  want(ids, at::kLong, "ids");
  want(scores, at::kFloat, "scores");
  TORCH_CHECK(q32.dim() == 2 && shadow.dim() == 2 && ids.dim() == 2 && shadow.size(1) == q32.size(1) && ids.size(0) == q32.size(0) &&
                  scores.sizes() == ids.sizes() && n_rows <= shadow.size(0), "q32 [nq, dim], shadow [>= n_rows, dim], ids / scores [nq, k]");
  same_device(q32, {&shadow, &ids, &scores}, "crs::score_rows_f32");
  c10::hip::HIPGuardMasqueradingAsCUDA g(q32.device());
  ok(crs_score_rows_f32(q32.data_ptr<float>(), (int)q32.size(0), (int)q32.size(1), shadow.data_ptr<float>(), n_rows, id_base, (int)ids.size(1),
                        ids.data_ptr<int64_t>(), scores.data_ptr<float>(), cur_stream(q32)), "crs::score_rows_f32");
}

// ---- exactness certificate + escalation (include/crs_hip.h, csrc/exact.hip) --------------------------------------------------
void refine_f32_cert_out(const Tensor& q32, const Tensor& q16, const Tensor& shadow, int64_t n_rows, int64_t id_base,
                         const Tensor& cand_ids, const Tensor& cand_scores, int64_t k_out, double row_err_max, int64_t slab_type,
                         Tensor out_scores, Tensor out_ids, Tensor status, Tensor exact_ws, int64_t cap) {
  want(q32, at::kFloat, "q32");
  want(q16, at::kHalf, "q16");
  want(shadow, at::kFloat, "shadow");
  want(cand_ids, at::kLong, "cand_ids");
  want(cand_scores, at::kFloat, "cand_scores");
  want(out_scores, at::kFloat, "out_scores");
  want(out_ids, at::kLong, "out_ids");
  want(status, at::kInt, "status");
  TORCH_CHECK(exact_ws.is_cuda() && exact_ws.is_contiguous(), "exact_ws must be a contiguous device tensor");
  const int64_t nq = q32.size(0);
  TORCH_CHECK(q32.dim() == 2 && q16.dim() == 2 && shadow.dim() == 2 && cand_ids.dim() == 2 && shadow.size(1) == q32.size(1) &&
                  cand_ids.size(0) == nq && cand_scores.sizes() == cand_ids.sizes() && n_rows <= shadow.size(0) && q16.size(0) == nq &&
                  q16.size(1) == crs_row_elems((int)q32.size(1), (int)slab_type),
              "q32 [nq, dim], q16 [nq, crs_row_elems], shadow [>= n_rows, dim], cand_ids / cand_scores [nq, k_in]");
  TORCH_CHECK(out_scores.numel() == nq * k_out && out_ids.numel() == nq * k_out && status.numel() == nq, "outputs must hold [nq, k_out], status [nq]");
  same_device(q32, {&q16, &shadow, &cand_ids, &cand_scores, &out_scores, &out_ids, &status, &exact_ws}, "crs::refine_f32_cert");
  c10::hip::HIPGuardMasqueradingAsCUDA g(q32.device());
  ok(crs_refine_f32_cert(q32.data_ptr<float>(), q16.data_ptr(), (int)nq, (int)q32.size(1), (int)slab_type, shadow.data_ptr<float>(), n_rows,
                         id_base, cand_ids.data_ptr<int64_t>(), cand_scores.data_ptr<float>(), (int)cand_ids.size(1), (int)k_out,
                         (float)row_err_max, out_scores.data_ptr<float>(), out_ids.data_ptr<int64_t>(), status.data_ptr<int32_t>(),
                         exact_ws.data_ptr(), (size_t)exact_ws.nbytes(), (int)cap, cur_stream(q32)), "crs::refine_f32_cert");
}

void escalate_exact(const Tensor& q32, const Tensor& q16, const Tensor& slab, c10::optional<Tensor> scales, const Tensor& shadow,
                    int64_t n_rows, int64_t id_base, int64_t k_out, Tensor out_scores, Tensor out_ids, Tensor status, Tensor exact_ws,
                    int64_t cap) {
  want(q32, at::kFloat, "q32");
  want(q16, at::kHalf, "q16");
  want(shadow, at::kFloat, "shadow");
  want(out_scores, at::kFloat, "out_scores");
  want(out_ids, at::kLong, "out_ids");
  want(status, at::kInt, "status");
  TORCH_CHECK(slab.is_cuda() && slab.is_contiguous() && slab.dim() == 2, "slab [rows, pdim]");
  TORCH_CHECK(exact_ws.is_cuda() && exact_ws.is_contiguous(), "exact_ws must be a contiguous device tensor");
  const int st = slab_type_of(slab);
  const int64_t nq = q32.size(0);
  const int pdim = crs_row_elems((int)q32.size(1), st);
  TORCH_CHECK(q32.dim() == 2 && q16.dim() == 2 && q16.size(0) == nq && q16.size(1) == pdim && slab.size(1) == pdim && shadow.dim() == 2 &&
                  shadow.size(1) == q32.size(1) && n_rows >= 1 && n_rows <= slab.size(0) && n_rows <= shadow.size(0),
              "q32 [nq, dim], q16 [nq, pdim], slab [>= n_rows, pdim], shadow [>= n_rows, dim]");
  TORCH_CHECK(out_scores.numel() == nq * k_out && out_ids.numel() == nq * k_out && status.numel() == nq, "outputs must hold [nq, k_out], status [nq]");
  if (st == CRS_SLAB_I8) {
    TORCH_CHECK(has(scales), "int8 slab needs scales");
    want(*scales, at::kFloat, "scales");
    TORCH_CHECK(scales->numel() >= n_rows, "scales shorter than n_rows");
  }
  same_device(q32, {&q16, &slab, opt_t(scales), &shadow, &out_scores, &out_ids, &status, &exact_ws}, "crs::escalate_exact");
  c10::hip::HIPGuardMasqueradingAsCUDA g(q32.device());
  ok(crs_escalate_exact(q32.data_ptr<float>(), q16.data_ptr(), (int)nq, (int)q32.size(1), st, slab.data_ptr(), (const float*)opt_ptr(scales),
                        shadow.data_ptr<float>(), n_rows, id_base, (int)k_out, out_scores.data_ptr<float>(), out_ids.data_ptr<int64_t>(),
                        status.data_ptr<int32_t>(), exact_ws.data_ptr(), (size_t)exact_ws.nbytes(), (int)cap, cur_stream(q32)),
     "crs::escalate_exact");
}

std::tuple<Tensor, Tensor> refine_f32(const Tensor& q32, const Tensor& shadow, int64_t n_rows, int64_t id_base, const Tensor& cand_ids,
                                      int64_t k_out) {
  c10::hip::HIPGuardMasqueradingAsCUDA g(q32.device());
  Tensor s = at::empty({q32.size(0), k_out}, q32.options().dtype(at::kFloat));
  Tensor i = at::empty({q32.size(0), k_out}, q32.options().dtype(at::kLong));
  refine_f32_out(q32, shadow, n_rows, id_base, cand_ids, k_out, s, i);
  return {s, i};
}

void merge_topk_out(const Tensor& scores, const Tensor& ids, int64_t k_out, Tensor out_scores, Tensor out_ids) {
  want(scores, at::kFloat, "scores");
  want(ids, at::kLong, "ids");
  want(out_scores, at::kFloat, "out_scores");
  want(out_ids, at::kLong, "out_ids");
  TORCH_CHECK(scores.dim() == 3 && ids.sizes() == scores.sizes(), "scores / ids must be [nlists, nq, k_in]");
  TORCH_CHECK(out_scores.numel() == scores.size(1) * k_out && out_ids.numel() == scores.size(1) * k_out, "outputs must hold [nq, k_out]");
  same_device(scores, {&ids, &out_scores, &out_ids}, "crs::merge_topk");
  c10::hip::HIPGuardMasqueradingAsCUDA g(scores.device());
  ok(crs_merge_topk(scores.data_ptr<float>(), ids.data_ptr<int64_t>(), (int)scores.size(0), (int)scores.size(1), (int)scores.size(2),
                    (int)k_out, out_scores.data_ptr<float>(), out_ids.data_ptr<int64_t>(), cur_stream(scores)), "crs::merge_topk");
}

std::tuple<Tensor, Tensor> merge_topk(const Tensor& scores, const Tensor& ids, int64_t k_out) {
  TORCH_CHECK(scores.dim() == 3, "scores must be [nlists, nq, k_in]");
  c10::hip::HIPGuardMasqueradingAsCUDA g(scores.device());
  Tensor s = at::empty({scores.size(1), k_out}, scores.options().dtype(at::kFloat));
  Tensor i = at::empty({scores.size(1), k_out}, scores.options().dtype(at::kLong));
  merge_topk_out(scores, ids, k_out, s, i);
  return {s, i};
}

void merge_topk_wire_out(const Tensor& wire, int64_t nlists, int64_t nq, int64_t k_in, int64_t k_out, Tensor out_scores, Tensor out_ids) {
  TORCH_CHECK(wire.is_cuda() && wire.is_contiguous() && wire.scalar_type() == at::kByte, "wire must be a contiguous uint8 device tensor");
  TORCH_CHECK((size_t)wire.numel() >= (size_t)nlists * crs_wire_bytes((int)nq, (int)k_in), "wire buffer shorter than nlists * crs_wire_bytes(nq, k_in)");
  want(out_scores, at::kFloat, "out_scores");
  want(out_ids, at::kLong, "out_ids");
  TORCH_CHECK(out_scores.numel() == nq * k_out && out_ids.numel() == nq * k_out, "outputs must hold [nq, k_out]");
  same_device(wire, {&out_scores, &out_ids}, "crs::merge_topk_wire");
  c10::hip::HIPGuardMasqueradingAsCUDA g(wire.device());
  ok(crs_merge_topk_wire(wire.data_ptr(), (int)nlists, (int)nq, (int)k_in, (int)k_out, out_scores.data_ptr<float>(),
                         out_ids.data_ptr<int64_t>(), cur_stream(wire)), "crs::merge_topk_wire");
}

// ---- encoder -------------------------------------------------------------------------------------------------
// desc = [vocab_size, hidden, layers, heads, ffn, max_pos, pooling, flags (CRS_ENC_*, optional)]; weights = [word_emb, pos_emb, type_emb, emb_ln_g, emb_ln_b]
// followed by 12 tensors per layer in crs_encoder_layer order (w_qkv b_qkv w_o b_o ln1_g ln1_b w_up b_up w_down b_down ln2_g ln2_b).
void encoder_forward(const Tensor& ids, const Tensor& lens, at::TensorList weights, at::IntArrayRef desc, double ln_eps, Tensor workspace,
                     Tensor out, c10::optional<Tensor> q16_out, int64_t slab_type, bool normalize, c10::optional<Tensor> hidden_out) {
  want(ids, at::kInt, "ids");
  want(lens, at::kInt, "lens");
  want(out, at::kFloat, "out");
  TORCH_CHECK(desc.size() == 7 || desc.size() == 8, "desc = [vocab_size, hidden, layers, heads, ffn, max_pos, pooling(, flags)]");
  crs_encoder_desc d{(int32_t)desc[0], (int32_t)desc[1], (int32_t)desc[2], (int32_t)desc[3], (int32_t)desc[4], (int32_t)desc[5],
                     (float)ln_eps, (int32_t)desc[6], desc.size() == 8 ? (int32_t)desc[7] : 0};
  TORCH_CHECK((int64_t)weights.size() == 5 + 12 * (int64_t)d.layers, "weights must hold 5 + 12 * layers tensors");
  TORCH_CHECK(ids.dim() == 2 && lens.numel() == ids.size(0) && out.numel() == ids.size(0) * d.hidden, "ids [B, S], lens [B], out [B, H]");
  for (const Tensor& t : weights) TORCH_CHECK(t.is_cuda() && t.is_contiguous() && t.device() == ids.device(), "weights must be contiguous tensors on the device of ids");
  same_device(ids, {&lens, &workspace, &out, opt_t(q16_out), opt_t(hidden_out)}, "crs::encoder_forward");
  std::vector<crs_encoder_layer> layers((size_t)d.layers);
  for (int l = 0; l < d.layers; ++l) {
    const Tensor* w = &weights[5 + 12 * l];
    layers[l] = crs_encoder_layer{w[0].data_ptr(), (const float*)w[1].data_ptr(), w[2].data_ptr(), (const float*)w[3].data_ptr(),
                                  (const float*)w[4].data_ptr(), (const float*)w[5].data_ptr(), w[6].data_ptr(), (const float*)w[7].data_ptr(),
                                  w[8].data_ptr(), (const float*)w[9].data_ptr(), (const float*)w[10].data_ptr(), (const float*)w[11].data_ptr()};
  }
  crs_encoder_weights cw{(const float*)weights[0].data_ptr(), (const float*)weights[1].data_ptr(), (const float*)weights[2].data_ptr(),
                         (const float*)weights[3].data_ptr(), (const float*)weights[4].data_ptr(), layers.data()};
  const int b = (int)ids.size(0), s = (int)ids.size(1);
  c10::hip::HIPGuardMasqueradingAsCUDA g(ids.device());
  if (q16_out.has_value() && q16_out->defined()) {
    want(*q16_out, at::kHalf, "q16_out");
    TORCH_CHECK(normalize && !(hidden_out.has_value() && hidden_out->defined()), "q16_out needs normalize=True and no hidden_out");
    TORCH_CHECK(q16_out->numel() == (int64_t)b * crs_row_elems(d.hidden, (int)slab_type), "q16_out must be [B, crs_row_elems(H, slab_type)]");
    ok(crs_encoder_forward_queries(&d, &cw, ids.data_ptr<int32_t>(), lens.data_ptr<int32_t>(), b, s, workspace.data_ptr(),
                                   (size_t)workspace.nbytes(), out.data_ptr<float>(), q16_out->data_ptr(), (int)slab_type, cur_stream(ids)),
       "crs::encoder_forward");
    return;
  }
  float* hid = nullptr;
  if (hidden_out.has_value() && hidden_out->defined()) {
    want(*hidden_out, at::kFloat, "hidden_out");
    TORCH_CHECK(hidden_out->numel() == (int64_t)b * s * d.hidden, "hidden_out must be [B, S, H]");
    hid = hidden_out->data_ptr<float>();
  }
  ok(crs_encoder_forward(&d, &cw, ids.data_ptr<int32_t>(), lens.data_ptr<int32_t>(), b, s, workspace.data_ptr(), (size_t)workspace.nbytes(),
                         out.data_ptr<float>(), normalize ? 1 : 0, hid, cur_stream(ids)), "crs::encoder_forward");
}

}  // namespace

TORCH_LIBRARY(crs, m) {
  m.def("slab_append(Tensor emb, Tensor(a!) slab, Tensor(b!)? scales, Tensor(c!)? shadow, int row0, Tensor(d!)? row_err=None) -> ()");
  m.def("queries_to_f16(Tensor q32, Tensor(a!) out16, int slab_type) -> ()");
  m.def("cosine_topk(Tensor q16, Tensor slab, Tensor? scales, int n_rows, int dim, int k, int id_base) -> (Tensor, Tensor)");
  m.def("cosine_topk_out(Tensor q16, Tensor slab, Tensor? scales, int n_rows, int dim, int k, int id_base, Tensor(a!) workspace, "
        "Tensor(b!) out_scores, Tensor(c!) out_ids) -> ()");
  m.def("refine_f32(Tensor q32, Tensor shadow, int n_rows, int id_base, Tensor cand_ids, int k_out) -> (Tensor, Tensor)");
  m.def("refine_f32_out(Tensor q32, Tensor shadow, int n_rows, int id_base, Tensor cand_ids, int k_out, Tensor(a!) out_scores, "
        "Tensor(b!) out_ids) -> ()");
  m.def("score_rows_f32_out(Tensor q32, Tensor shadow, int n_rows, int id_base, Tensor ids, Tensor(a!) scores) -> ()");
  m.def("refine_f32_cert_out(Tensor q32, Tensor q16, Tensor shadow, int n_rows, int id_base, Tensor cand_ids, Tensor cand_scores, int k_out, "
        "float row_err_max, int slab_type, Tensor(a!) out_scores, Tensor(b!) out_ids, Tensor(c!) status, Tensor(d!) exact_ws, int cap) -> ()");
  m.def("escalate_exact(Tensor q32, Tensor q16, Tensor slab, Tensor? scales, Tensor shadow, int n_rows, int id_base, int k_out, "
        "Tensor(a!) out_scores, Tensor(b!) out_ids, Tensor(c!) status, Tensor(d!) exact_ws, int cap) -> ()");
  m.def("merge_topk(Tensor scores, Tensor ids, int k_out) -> (Tensor, Tensor)");
  m.def("merge_topk_out(Tensor scores, Tensor ids, int k_out, Tensor(a!) out_scores, Tensor(b!) out_ids) -> ()");
  m.def("merge_topk_wire_out(Tensor wire, int nlists, int nq, int k_in, int k_out, Tensor(a!) out_scores, Tensor(b!) out_ids) -> ()");
  m.def("encoder_forward(Tensor ids, Tensor lens, Tensor[] weights, int[] desc, float ln_eps, Tensor(a!) workspace, Tensor(b!) out, "
        "Tensor(c!)? q16_out, int slab_type, bool normalize, Tensor(d!)? hidden_out) -> ()");
}

TORCH_LIBRARY_IMPL(crs, CUDA, m) {   // the HIP backend of torch-ROCm dispatches under the "CUDA" key
  m.impl("slab_append", &slab_append);
  m.impl("queries_to_f16", &queries_to_f16);
  m.impl("cosine_topk", &cosine_topk);
  m.impl("cosine_topk_out", &cosine_topk_out);
  m.impl("refine_f32", &refine_f32);
  m.impl("refine_f32_out", &refine_f32_out);
  m.impl("score_rows_f32_out", &score_rows_f32_out);
  m.impl("refine_f32_cert_out", &refine_f32_cert_out);
  m.impl("escalate_exact", &escalate_exact);
  m.impl("merge_topk", &merge_topk);
  m.impl("merge_topk_out", &merge_topk_out);
  m.impl("merge_topk_wire_out", &merge_topk_wire_out);
  m.impl("encoder_forward", &encoder_forward);
}
