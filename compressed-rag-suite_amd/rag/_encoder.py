"""ctypes binding of the encoder half of libcrs_hip.so (include/crs_encoder.h) + weight upload.

``HipEncoder`` owns the device copies of a BERT-family checkpoint (HuggingFace state-dict names,
fp32 numpy in; matrices are cast to fp16 and Q/K/V stacked on upload) and runs the forward
through ``crs_encoder_forward``.  Token ids in, pooled sentence embeddings out; no CPU fallback.
"""
from __future__ import annotations

import ctypes
from ctypes import POINTER, Structure, byref, c_float, c_int, c_int32, c_size_t, c_void_p
from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np

from rag import _native as nat

POOL_MEAN, POOL_CLS = 0, 1
ENC_SMALL_LDS = 1      # CRS_ENC_SMALL_LDS: kernel forms of <= 48 KB of LDS (forwards that run beside a scan)


class EncoderDesc(Structure):
    _fields_ = [("vocab_size", c_int32), ("hidden", c_int32), ("layers", c_int32), ("heads", c_int32),
                ("ffn", c_int32), ("max_pos", c_int32), ("ln_eps", c_float), ("pooling", c_int32), ("flags", c_int32)]


class EncoderLayer(Structure):
    _fields_ = [(n, c_void_p) for n in ("w_qkv", "b_qkv", "w_o", "b_o", "ln1_g", "ln1_b", "w_up", "b_up",
                                        "w_down", "b_down", "ln2_g", "ln2_b")]


class EncoderWeights(Structure):
    _fields_ = [("word_emb", c_void_p), ("pos_emb", c_void_p), ("type_emb", c_void_p), ("emb_ln_g", c_void_p),
                ("emb_ln_b", c_void_p), ("layers", POINTER(EncoderLayer))]


nat.register_signatures({
    "crs_encoder_workspace_bytes": (c_int, [POINTER(EncoderDesc), c_int, c_int, POINTER(c_size_t)]),
    "crs_encoder_forward": (c_int, [POINTER(EncoderDesc), POINTER(EncoderWeights), c_void_p, c_void_p, c_int,
                                    c_int, c_void_p, c_size_t, c_void_p, c_int, c_void_p, c_void_p]),
    "crs_encoder_forward_queries": (c_int, [POINTER(EncoderDesc), POINTER(EncoderWeights), c_void_p, c_void_p, c_int,
                                            c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_int, c_void_p]),
    "crs_gemm_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                             c_void_p]),
})


@dataclass(frozen=True)
class ModelShape:
    vocab_size: int
    hidden: int
    layers: int
    heads: int
    ffn: int
    max_pos: int
    ln_eps: float = 1e-12
    pooling: str = "mean"
    max_seq: int = 256


def gemm_f16(a, w, bias=None, residual=None, mode: int = 0):
    """a: cuda fp16 [M, K]; w: cuda fp16 [N, K]; returns fp16 [M, N] (modes 0/1) or fp32 (mode 2)."""
    import torch
    m, k = a.shape
    n = w.shape[0]
    out = torch.empty((m, n), dtype=torch.float32 if mode == 2 else torch.float16, device=a.device)
    nat.check(nat.load().crs_gemm_f16(nat._ptr(a), nat._ptr(w), nat._ptr(bias), nat._ptr(residual), nat._ptr(out),
                                      m, n, k, mode, nat._stream_ptr()))
    return out


class HipEncoder:
    def __init__(self, shape: ModelShape, weights: Dict[str, np.ndarray], device=None):
        import torch
        nat.require_gpu()
        nat.load()
        self.shape = shape
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self._keep = []  # device tensors the C structs point into

        def dev32(name):
            t = torch.from_numpy(np.ascontiguousarray(weights[name], dtype=np.float32)).to(self.device)
            self._keep.append(t)
            return t

        def dev16(*names):
            arrs = [np.ascontiguousarray(weights[n], dtype=np.float32) for n in names]
            t = torch.from_numpy(np.concatenate(arrs, axis=0)).to(self.device).to(torch.float16).contiguous()
            self._keep.append(t)
            return t

        def cat32(*names):
            t = torch.from_numpy(np.concatenate([np.asarray(weights[n], dtype=np.float32) for n in names])).to(self.device)
            self._keep.append(t)
            return t

        self.desc = EncoderDesc(shape.vocab_size, shape.hidden, shape.layers, shape.heads, shape.ffn, shape.max_pos,
                                shape.ln_eps, POOL_CLS if shape.pooling == "cls" else POOL_MEAN, 0)
        self._layers = (EncoderLayer * shape.layers)()
        for i in range(shape.layers):
            p = f"encoder.layer.{i}."
            vals = [
                dev16(p + "attention.self.query.weight", p + "attention.self.key.weight", p + "attention.self.value.weight"),
                cat32(p + "attention.self.query.bias", p + "attention.self.key.bias", p + "attention.self.value.bias"),
                dev16(p + "attention.output.dense.weight"), dev32(p + "attention.output.dense.bias"),
                dev32(p + "attention.output.LayerNorm.weight"), dev32(p + "attention.output.LayerNorm.bias"),
                dev16(p + "intermediate.dense.weight"), dev32(p + "intermediate.dense.bias"),
                dev16(p + "output.dense.weight"), dev32(p + "output.dense.bias"),
                dev32(p + "output.LayerNorm.weight"), dev32(p + "output.LayerNorm.bias"),
            ]
            for (fname, _), t in zip(EncoderLayer._fields_, vals):
                setattr(self._layers[i], fname, t.data_ptr())
        self.weights = EncoderWeights(
            dev32("embeddings.word_embeddings.weight").data_ptr(),
            dev32("embeddings.position_embeddings.weight").data_ptr(),
            dev32("embeddings.token_type_embeddings.weight").data_ptr(),
            dev32("embeddings.LayerNorm.weight").data_ptr(),
            dev32("embeddings.LayerNorm.bias").data_ptr(),
            ctypes.cast(self._layers, POINTER(EncoderLayer)))
        self._ws = None
        # the same tensors in the order torch.ops.crs.encoder_forward takes them (csrc/torch_ops.cpp)
        self._wlist = self._keep[-5:] + self._keep[:-5]     # embeddings first, then 12 tensors per layer
        assert len(self._wlist) == 5 + 12 * shape.layers

    @property
    def _desc_list(self):
        """crs_encoder_desc as the int list torch.ops.crs.encoder_forward takes (read from `desc`, so tests that
        switch `desc.pooling` are honoured)."""
        d = self.desc
        return [d.vocab_size, d.hidden, d.layers, d.heads, d.ffn, d.max_pos, d.pooling, d.flags]

    def workspace_bytes(self, batch: int, seq: int) -> int:
        out = c_size_t(0)
        nat.check(nat.load().crs_encoder_workspace_bytes(byref(self.desc), batch, seq, byref(out)))
        return int(out.value)

    def forward(self, ids, lens, normalize: bool = True, return_hidden: bool = False, out=None, workspace=None,
                q16_out=None, slab_type: int = nat.SLAB_F16, small_lds: bool = False):
        """ids: int32 [B, S] (numpy or cuda tensor, right padded), lens: int32 [B].
        Returns cuda fp32 [B, H] (and [B, S, H] hidden states when return_hidden).  With `q16_out`
        (cuda fp16 [B, padded_dim]) the pooled, normalised embeddings are also written there in the scan's
        query layout (crs_encoder_forward_queries)."""
        import torch
        if not isinstance(ids, torch.Tensor):
            ids = torch.from_numpy(np.ascontiguousarray(ids, dtype=np.int32))
        if not isinstance(lens, torch.Tensor):
            lens = torch.from_numpy(np.ascontiguousarray(lens, dtype=np.int32))
        ids = ids.to(device=self.device, dtype=torch.int32).contiguous()
        lens = lens.to(device=self.device, dtype=torch.int32).contiguous()
        b, s = ids.shape
        need = self.workspace_bytes(b, s)
        ws = workspace
        if ws is None:   # private scratch; pass `workspace` to run several forwards concurrently on different streams
            if self._ws is None or self._ws.numel() < need:
                self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            ws = self._ws
        elif ws.numel() < need:
            raise ValueError(f"encoder workspace too small: {ws.numel()} < {need}")
        if out is None:
            out = torch.empty((b, self.shape.hidden), dtype=torch.float32, device=self.device)
        if q16_out is not None:
            if return_hidden or not normalize:
                raise ValueError("q16_out needs normalize=True and return_hidden=False")
            if tuple(q16_out.shape) != (b, nat.padded_dim(self.shape.hidden, slab_type)) or q16_out.dtype != torch.float16:
                raise ValueError("q16_out must be fp16 [batch, padded_dim]")
        hidden = torch.empty((b, s, self.shape.hidden), dtype=torch.float32, device=self.device) if return_hidden else None
        desc = self._desc_list
        if small_lds:          # per call, not process-wide: the role-lane engine asks for it, everyone else gets the default forms
            desc = desc[:7] + [desc[7] | ENC_SMALL_LDS]
        with nat._translate():
            nat.ops().encoder_forward(ids, lens, self._wlist, desc, float(self.desc.ln_eps), ws, out, q16_out,
                                      int(slab_type), bool(normalize), hidden)
        return (out, hidden) if return_hidden else out
