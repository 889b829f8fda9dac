"""Text embedding generation on the MI355X.

Drop-in for the reference's ``EmbeddingModel`` (/root/reference/rag/embedding.py:14-91): same
config keys (``model_name``, ``device``, ``batch_size``, ``normalize``), same methods and return
types (``embed`` -> float32 numpy [n, d]; a single string gives [1, d]).  Underneath, what
sentence-transformers did (tokenise -> BertModel -> pooling -> normalise) is: host WordPiece ->
``crs_encoder_forward`` (hand-written HIP: MFMA GEMMs, fused attention, fp32 LayerNorm / pooling).

Model resolution, offline by construction (the reference fetches by hub name, which cannot work
here): ``model_path`` (additive key) or ``model_name`` may be a LOCAL sentence-transformers
directory (config.json, model.safetensors, vocab.txt, optional sentence_bert_config.json and
1_Pooling/config.json); else ``$CRS_MODEL_DIR/<basename of model_name>`` is tried; else, when
``model_name`` is ``synthetic:<minilm|bge|tiny>`` or ``CRS_ALLOW_SYNTHETIC_WEIGHTS=1`` is set,
seeded random weights of the named architecture are used with a hash tokeniser (plumbing and
benchmarks only -- embeddings are then meaningless as language).  Anything else raises.
"""
from __future__ import annotations

import json
import logging
import os
from typing import Dict, List, Optional, Union

import numpy as np

from rag.chunking import Chunk
from rag import _native as nat
from rag.tokenizer import HashTokenizer, WordPieceTokenizer, make_wordpiece_tokenizer, pad_batch

logger = logging.getLogger(__name__)

# architectures this build knows by name (public model cards; SURVEY.md section 2.3 shapes)
_KNOWN = {
    "all-minilm-l6-v2": dict(vocab_size=30522, hidden=384, layers=6, heads=12, ffn=1536, max_pos=512,
                             pooling="mean", max_seq=256),
    "bge-base-en-v1.5": dict(vocab_size=30522, hidden=768, layers=12, heads=12, ffn=3072, max_pos=512,
                             pooling="cls", max_seq=512),
    "tiny": dict(vocab_size=1000, hidden=64, layers=2, heads=4, ffn=256, max_pos=64, pooling="mean", max_seq=64),
}
_ALIASES = {"minilm": "all-minilm-l6-v2", "bge": "bge-base-en-v1.5", "bge-base": "bge-base-en-v1.5"}


def synthetic_weights(shape, seed: int = 0, scale: float = 0.05) -> Dict[str, np.ndarray]:
    """Seeded random checkpoint with HuggingFace BertModel tensor names (one PCG64 stream per tensor)."""
    h, f = shape.hidden, shape.ffn
    names = [("embeddings.word_embeddings.weight", (shape.vocab_size, h)),
             ("embeddings.position_embeddings.weight", (shape.max_pos, h)),
             ("embeddings.token_type_embeddings.weight", (2, h)),
             ("embeddings.LayerNorm.weight", (h,)), ("embeddings.LayerNorm.bias", (h,))]
    for i in range(shape.layers):
        p = f"encoder.layer.{i}."
        names += [(p + "attention.self.query.weight", (h, h)), (p + "attention.self.query.bias", (h,)),
                  (p + "attention.self.key.weight", (h, h)), (p + "attention.self.key.bias", (h,)),
                  (p + "attention.self.value.weight", (h, h)), (p + "attention.self.value.bias", (h,)),
                  (p + "attention.output.dense.weight", (h, h)), (p + "attention.output.dense.bias", (h,)),
                  (p + "attention.output.LayerNorm.weight", (h,)), (p + "attention.output.LayerNorm.bias", (h,)),
                  (p + "intermediate.dense.weight", (f, h)), (p + "intermediate.dense.bias", (f,)),
                  (p + "output.dense.weight", (h, f)), (p + "output.dense.bias", (h,)),
                  (p + "output.LayerNorm.weight", (h,)), (p + "output.LayerNorm.bias", (h,))]
    out = {}
    for idx, (name, shp) in enumerate(names):
        rng = np.random.Generator(np.random.PCG64([seed, idx]))
        if name.endswith("LayerNorm.weight"):
            a = 1.0 + 0.05 * rng.standard_normal(shp, dtype=np.float32)
        elif name.endswith(".bias"):
            a = 0.02 * rng.standard_normal(shp, dtype=np.float32)
        else:
            a = scale * rng.standard_normal(shp, dtype=np.float32)
        out[name] = a.astype(np.float32)
    return out


def _load_local_dir(path: str):
    """(ModelShape, weights dict, tokenizer, pre_lower, normalize_module) from a sentence-transformers directory.

    What ``SentenceTransformer(path)`` (reference rag/embedding.py:33) assembles from the same files:
    modules.json lists the pipeline (Transformer at "", Pooling at "1_Pooling", optional Normalize); the
    Transformer module reads config.json + model.safetensors + the tokenizer files and takes
    ``max_seq_length`` / ``do_lower_case`` from sentence_bert_config.json -- the latter is only an extra
    ``str.lower()`` pass BEFORE the tokenizer, whose own casing rule comes from its own files."""
    from safetensors.numpy import load_file
    from rag._encoder import ModelShape
    from rag.tokenizer import tokenizer_from_model_dir
    tr_dir, pool_dir, has_normalize = path, os.path.join(path, "1_Pooling"), False
    mj = os.path.join(path, "modules.json")
    if os.path.exists(mj):
        with open(mj) as fh:
            for mod in json.load(fh):
                kind = str(mod.get("type", "")).rsplit(".", 1)[-1]
                sub = os.path.join(path, mod.get("path", "") or "")
                if kind == "Transformer":
                    tr_dir = sub
                elif kind == "Pooling":
                    pool_dir = sub
                elif kind == "Normalize":
                    has_normalize = True
                elif kind:
                    raise NotImplementedError(f"sentence-transformers module '{mod.get('type')}' is not supported by this encoder")
    with open(os.path.join(tr_dir, "config.json")) as fh:
        cfg = json.load(fh)
    max_seq, pooling, pre_lower = cfg.get("max_position_embeddings", 512), "mean", False
    sb = os.path.join(tr_dir, "sentence_bert_config.json")
    if os.path.exists(sb):
        with open(sb) as fh:
            sbc = json.load(fh)
        max_seq = sbc.get("max_seq_length", max_seq) or max_seq
        pre_lower = bool(sbc.get("do_lower_case", False))
    pc = os.path.join(pool_dir, "config.json")
    if os.path.exists(pc):
        with open(pc) as fh:
            pcfg = json.load(fh)
        modes = [m for m in ("cls_token", "mean_tokens", "max_tokens", "mean_sqrt_len_tokens", "weightedmean_tokens", "lasttoken")
                 if pcfg.get("pooling_mode_" + m)]
        if modes == ["cls_token"]:
            pooling = "cls"
        elif modes not in ([], ["mean_tokens"]):
            raise NotImplementedError(f"pooling mode(s) {modes} are not supported (mean and CLS are)")
    shape = ModelShape(cfg["vocab_size"], cfg["hidden_size"], cfg["num_hidden_layers"], cfg["num_attention_heads"],
                       cfg["intermediate_size"], cfg["max_position_embeddings"], cfg.get("layer_norm_eps", 1e-12),
                       pooling, min(max_seq, cfg["max_position_embeddings"]))
    raw = load_file(os.path.join(tr_dir, "model.safetensors"))
    weights = {(k[5:] if k.startswith("bert.") else k): np.asarray(v, dtype=np.float32) for k, v in raw.items()}
    return shape, weights, tokenizer_from_model_dir(tr_dir), pre_lower, has_normalize


class EmbeddingModel:
    """Sentence encoder wrapper (BERT-family checkpoints) running on the GPU."""

    def __init__(self, config: dict):
        self.model_name = config.get('model_name', 'sentence-transformers/all-MiniLM-L6-v2')
        self.batch_size = config.get('batch_size', 32)
        self.normalize = config.get('normalize', True)
        self.device = self._get_device(config.get('device', 'cuda'))
        logger.info(f"Loading embedding model: {self.model_name}")
        self._pre_lower = False      # sentence_bert_config.json do_lower_case: an extra str.lower() before the tokenizer
        shape, weights, self.tokenizer = self._resolve(config)
        if config.get('max_seq_length'):
            from dataclasses import replace
            shape = replace(shape, max_seq=min(int(config['max_seq_length']), shape.max_pos))
        if config.get('pooling'):
            from dataclasses import replace
            shape = replace(shape, pooling=config['pooling'])
        from rag._encoder import HipEncoder
        self.shape = shape
        self.model = HipEncoder(shape, weights, device=self.device)
        self.dimension = shape.hidden
        logger.info(f"Model loaded on {self.device}")
        logger.info(f"Embedding dimension: {self.dimension}")

    def _get_device(self, device_preference: str) -> str:
        """Always the ROCm device (torch-ROCm reports it as 'cuda'); there is no CPU path."""
        nat.require_gpu()
        if device_preference not in ("cuda", None) and not str(device_preference).startswith("cuda"):
            logger.warning(f"device '{device_preference}' requested; this build only runs on the GPU ('cuda')")
            return "cuda"
        return device_preference or "cuda"

    def _resolve(self, config: dict):
        from rag._encoder import ModelShape
        name = self.model_name
        for cand in (config.get('model_path'), name,
                     os.path.join(os.environ.get("CRS_MODEL_DIR", ""), os.path.basename(name)) if os.environ.get("CRS_MODEL_DIR") else None):
            if cand and os.path.isdir(cand) and os.path.exists(os.path.join(cand, "model.safetensors")):
                logger.info(f"Loading local checkpoint {cand}")
                shape, weights, tok, self._pre_lower, _has_norm = _load_local_dir(cand)
                return shape, weights, tok
        key = name.split(":", 1)[1] if name.startswith("synthetic:") else os.path.basename(name)
        key = _ALIASES.get(key.lower(), key.lower())
        if key in _KNOWN and (name.startswith("synthetic:") or os.environ.get("CRS_ALLOW_SYNTHETIC_WEIGHTS") == "1"):
            logger.warning(f"Using SYNTHETIC weights for architecture '{key}' (no checkpoint available offline)")
            shape = ModelShape(ln_eps=1e-12, **_KNOWN[key])
            seed = int(config.get('synthetic_seed', 0))
            return shape, synthetic_weights(shape, seed), HashTokenizer(shape.vocab_size)
        raise FileNotFoundError(
            f"No local checkpoint for '{name}': pass a sentence-transformers directory as model_name/model_path, "
            f"set CRS_MODEL_DIR, or use 'synthetic:minilm' (models cannot be downloaded here)")

    # ---- encoding ------------------------------------------------------------------------------
    def tokenize(self, texts: List[str]):
        """-> list of id lists ([CLS] ... [SEP], truncated to max_seq).  Text is stripped (and lower-cased when the
        model's sentence_bert_config.json says so) first, as sentence-transformers' Transformer.tokenize does."""
        texts = [str(t).strip() for t in texts]
        if self._pre_lower:
            texts = [t.lower() for t in texts]
        if hasattr(self.tokenizer, "encode_batch"):
            return self.tokenizer.encode_batch(texts, self.shape.max_seq)
        return [self.tokenizer.encode(t, self.shape.max_seq) for t in texts]

    def embed_device(self, texts: Union[str, List[str]]):
        """Embeddings as a cuda fp32 tensor [n, d] in input order (additive, zero-copy index build)."""
        import torch
        if isinstance(texts, str):
            texts = [texts]
        n = len(texts)
        out = torch.empty((n, self.dimension), dtype=torch.float32, device=self.model.device)
        if n == 0:
            return out
        token_ids = self.tokenize(texts)
        if n <= self.batch_size:   # one batch: no length sort, no scatter (its index tensor is a blocking H2D copy per call)
            ids, lens = pad_batch(token_ids, getattr(self.tokenizer, "pad_id", 0),
                                  short_steps=tuple(st for st in (16, 32, 64) if st <= self.shape.max_seq))
            return self.model.forward(ids, lens, normalize=bool(self.normalize), out=out)
        # longest first, like SentenceTransformer.encode: least padding per batch
        order = sorted(range(n), key=lambda i: -len(texts[i]))
        # Two batches in flight on two side streams: the forward's HBM-bound phases (LayerNorm epilogues: a quarter of a
        # MiniLM layer at index-build size) of one batch run under the matrix phases of the other (+ 3-5 % tokens/s,
        # bench.py --workload enc-* --enc-inflight 2); host-side padding of the next batch overlaps too.
        cur = torch.cuda.current_stream(out.device)
        lanes = [torch.cuda.Stream(device=out.device) for _ in range(2)]
        lane_ws = [None, None]
        ready = torch.cuda.Event()
        ready.record(cur)
        for b, lo in enumerate(range(0, n, self.batch_size)):
            sel = order[lo: lo + self.batch_size]
            ids, lens = pad_batch([token_ids[i] for i in sel], getattr(self.tokenizer, "pad_id", 0),
                                  short_steps=tuple(st for st in (16, 32, 64) if st <= self.shape.max_seq))
            st = lanes[b & 1]
            with torch.cuda.stream(st):
                if b < 2:
                    st.wait_event(ready)        # `out` (and anything the caller queued) before the first write
                need = self.model.workspace_bytes(int(ids.shape[0]), int(ids.shape[1]))
                if lane_ws[b & 1] is None or lane_ws[b & 1].numel() < need:     # the encoder's own scratch is shared:
                    lane_ws[b & 1] = torch.empty(need, dtype=torch.uint8, device=out.device)   # concurrent forwards need one each
                emb = self.model.forward(ids, lens, normalize=bool(self.normalize), workspace=lane_ws[b & 1])
                out[torch.as_tensor(sel, device=out.device)] = emb
        for st in lanes:
            out.record_stream(st)
            cur.wait_stream(st)
        return out

    def embed(self, texts: Union[str, List[str]], show_progress: bool = False) -> np.ndarray:
        """Embeddings for text(s): numpy float32 [n, d] (n = 1 for a single string)."""
        return self.embed_device(texts).cpu().numpy()

    def embed_chunks(self, chunks: List[Chunk], show_progress: bool = True) -> np.ndarray:
        return self.embed([chunk.text for chunk in chunks], show_progress=show_progress)

    def embed_chunks_device(self, chunks: List[Chunk], show_progress: bool = True):
        """embed_chunks without leaving the device (additive): cuda fp32 [n, d] for VectorStore.create_index."""
        return self.embed_device([chunk.text for chunk in chunks])

    def get_dimension(self) -> int:
        return self.dimension
