"""Vector store: a flat, HBM-resident fp16 / int8 slab searched exactly on the MI355X.

Drop-in for the reference's ChromaDB-backed ``VectorStore`` (/root/reference/rag/indexing.py:14-211):
same constructor config keys, same method names, argument meaning, return shapes and exceptions.
What changed underneath: ``collection.add`` -> one ``crs_slab_append_f32`` launch per batch (rows are
L2-normalised, cast to fp16 or int8+scale and appended to a device slab); ``collection.query`` ->
``crs_cosine_topk`` (exact brute-force scan + top-k, include/crs_hip.h).  Documents, ids and
metadata stay on the host, indexed by row.

New, additive surface: ``search_batch`` (many queries per launch), ``add_embeddings_device``
(zero-copy append of encoder output), optional ``index_dtype`` / ``refine_fp32`` config keys that
default so an unmodified reference config.json works.  With ``torch.distributed`` initialised and
``sharded=True`` each rank keeps a contiguous row shard and searches are merged with one RCCL
all-gather.

There is no CPU fallback: without a GPU or without libcrs_hip.so every search raises.
"""
from __future__ import annotations

import json
import logging
import os
from typing import Any, Dict, List, Optional, Sequence

import numpy as np

from rag.chunking import Chunk
from rag import _native as nat
from rag import _shard

logger = logging.getLogger(__name__)

_EMPTY = {'ids': [[]], 'documents': [[]], 'metadatas': [[]], 'distances': [[]]}


class SlabCollection:
    """What ``VectorStore.collection`` exposes (the retriever reads ``.metadata`` and the harness
    ``.count()``, reference rag/retrieval.py:48-50).  Owns the device slab and the host sidecars."""

    def __init__(self, name: str, index_dtype: str, refine_fp32: bool, device):
        self.name = name
        self.metadata = {"hnsw:space": "cosine"}
        self.index_dtype = index_dtype
        self.refine_fp32 = refine_fp32
        self.device = device
        self.dim: Optional[int] = None
        self.pdim: Optional[int] = None
        self.n = 0                     # rows in this rank's shard
        self.capacity = 0
        self.slab = None               # torch [capacity, pdim] fp16 | int8
        self.scales = None             # torch [capacity] fp32 (int8 only)
        self.shadow = None             # torch [capacity, dim] fp32 (refine_fp32 only)
        self.ids: List[str] = []
        self.documents: List[str] = []
        self.metadatas: List[dict] = []
        self._workspace = None

    @property
    def slab_type(self) -> int:
        return nat.SLAB_I8 if self.index_dtype == "int8" else nat.SLAB_F16

    def count(self) -> int:
        return len(self.ids)

    # -- storage -------------------------------------------------------------------------------
    def _reserve(self, rows: int, dim: int):
        import torch
        if self.dim is None:
            self.dim, self.pdim = dim, nat.padded_dim(dim, self.slab_type)
        elif dim != self.dim:
            raise ValueError(f"Embedding dimension {dim} doesn't match the index dimension {self.dim}")
        if rows <= self.capacity:
            return
        cap = max(rows, int(self.capacity * 1.5) + 1024)
        dt = torch.int8 if self.slab_type == nat.SLAB_I8 else torch.float16
        new = torch.zeros((cap, self.pdim), dtype=dt, device=self.device)
        if self.slab is not None and self.n:
            new[: self.n].copy_(self.slab[: self.n])
        self.slab = new
        if self.slab_type == nat.SLAB_I8:
            ns = torch.zeros(cap, dtype=torch.float32, device=self.device)
            if self.scales is not None and self.n:
                ns[: self.n].copy_(self.scales[: self.n])
            self.scales = ns
        if self.refine_fp32:
            nsh = torch.zeros((cap, self.dim), dtype=torch.float32, device=self.device)
            if self.shadow is not None and self.n:
                nsh[: self.n].copy_(self.shadow[: self.n])
            self.shadow = nsh
        self.capacity = cap

    def append_device(self, emb):
        """emb: cuda fp32 [m, dim] (contiguous).  Appends m rows to the slab."""
        m, dim = emb.shape
        self._reserve(self.n + m, dim)
        nat.slab_append_f32(emb, self.slab, self.n, self.slab_type, scales=self.scales, shadow=self.shadow)
        self.n += m

    def workspace(self, nq: int, k: int):
        import torch
        need = nat.scan_workspace_bytes(nq, self.dim, k, max(self.n, 1))
        if self._workspace is None or self._workspace.numel() < need:
            self._workspace = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._workspace


class VectorStore:
    """Vector store on an HBM slab.  Handles storage, indexing and exact similarity search."""

    def __init__(self, config: dict):
        self.collection_name = config.get('collection_name', 'rag_documents')
        self.persist_directory = config.get('persist_directory', None)
        # additive knobs (absent from the reference config.json -> defaults)
        self.index_dtype = config.get('index_dtype', 'fp16')
        if self.index_dtype not in ('fp16', 'int8'):
            raise ValueError(f"index_dtype must be 'fp16' or 'int8', got {self.index_dtype!r}")
        self.refine_fp32 = bool(config.get('refine_fp32', False))
        self.refine_factor = int(config.get('refine_factor', 4))
        self.sharded = bool(config.get('sharded', False))
        self._device = config.get('device', None)
        self.client = self  # the reference keeps a chromadb client here; nothing else reads it
        self.collection: Optional[SlabCollection] = None
        self._shard_map: Optional[_shard.ShardMap] = None
        self._initialize_collection()

    # -- helpers -------------------------------------------------------------------------------
    def _torch_device(self):
        import torch
        nat.require_gpu()
        if self._device is not None and str(self._device).startswith("cuda"):
            return torch.device(self._device)
        return torch.device("cuda", torch.cuda.current_device())

    def _dist(self):
        if not self.sharded:
            return None
        import torch.distributed as dist
        return dist if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 else None

    def _persist_paths(self):
        base = os.path.join(self.persist_directory, self.collection_name)
        return base + ".slab.npz", base + ".docs.json"

    def _initialize_collection(self):
        """Re-open a persisted collection if there is one (reference: get_collection, :46-55)."""
        if not self.persist_directory:
            logger.info("Using in-memory (HBM) storage")
            return
        slab_path, docs_path = self._persist_paths()
        if not (os.path.exists(slab_path) and os.path.exists(docs_path)):
            logger.info(f"Collection '{self.collection_name}' will be created on first add")
            return
        try:
            import torch
            z = np.load(slab_path, allow_pickle=False)
            with open(docs_path) as fh:
                side = json.load(fh)
            col = SlabCollection(self.collection_name, str(z["index_dtype"]), self.refine_fp32, self._torch_device())
            n, dim = int(z["n"]), int(z["dim"])
            col._reserve(n, dim)
            col.slab[:n].copy_(torch.from_numpy(z["slab"]))
            if col.slab_type == nat.SLAB_I8:
                col.scales[:n].copy_(torch.from_numpy(z["scales"]))
            if self.refine_fp32 and "shadow" in z.files:
                col.shadow[:n].copy_(torch.from_numpy(z["shadow"]))
            elif self.refine_fp32:
                col.refine_fp32 = False
                logger.warning("persisted collection has no fp32 shadow; refine_fp32 disabled")
            col.n = n
            col.ids, col.documents, col.metadatas = side["ids"], side["documents"], side["metadatas"]
            self.index_dtype = col.index_dtype
            self.collection = col
            logger.info(f"Loaded existing collection: {self.collection_name} ({col.count()} rows)")
        except nat.NativeError:
            raise
        except Exception as e:  # unreadable files behave like "no collection yet"
            logger.warning(f"Could not load persisted collection: {e}")

    def persist(self):
        """Write the slab + sidecars under persist_directory (the PersistentClient analogue)."""
        if not self.persist_directory or self.collection is None:
            return
        os.makedirs(self.persist_directory, exist_ok=True)
        col = self.collection
        slab_path, docs_path = self._persist_paths()
        arrays = {"slab": col.slab[: col.n].cpu().numpy(), "n": np.int64(col.n), "dim": np.int64(col.dim),
                  "index_dtype": np.str_(col.index_dtype)}
        if col.slab_type == nat.SLAB_I8:
            arrays["scales"] = col.scales[: col.n].cpu().numpy()
        if col.refine_fp32:
            arrays["shadow"] = col.shadow[: col.n].cpu().numpy()
        np.savez(slab_path, **arrays)
        with open(docs_path, "w") as fh:
            json.dump({"ids": col.ids, "documents": col.documents, "metadatas": col.metadatas}, fh)

    @staticmethod
    def _chunk_metadata(chunk, fields: Sequence[str]) -> dict:
        meta = {}
        for field in fields:
            value = getattr(chunk, field, None)
            if value is None:
                continue
            meta[field] = value if isinstance(value, (str, int, float)) else str(value)
        return meta

    # -- index build ---------------------------------------------------------------------------
    def create_index(self, chunks: List[Chunk], embeddings, metadata_fields: Optional[List[str]] = None):
        """Append chunks + their embeddings (numpy fp32 [n, d], or a cuda fp32 tensor) to the index."""
        if len(chunks) == 0:
            logger.warning("No chunks provided for indexing")
            return
        if len(chunks) != len(embeddings):
            raise ValueError(f"Chunk count ({len(chunks)}) doesn't match embedding count ({len(embeddings)})")
        fields = ['page_number', 'section', 'tokens'] if metadata_fields is None else metadata_fields
        try:
            import torch
            if self.collection is None:
                self.collection = SlabCollection(self.collection_name, self.index_dtype, self.refine_fp32,
                                                 self._torch_device())
                logger.info(f"Created new collection: {self.collection_name}")
            col = self.collection
            if isinstance(embeddings, torch.Tensor):
                emb = embeddings.to(device=col.device, dtype=torch.float32).contiguous()
            else:
                emb = torch.from_numpy(np.ascontiguousarray(embeddings, dtype=np.float32)).to(col.device)
            if emb.ndim != 2:
                raise ValueError(f"embeddings must be 2-D, got shape {tuple(emb.shape)}")
            logger.info(f"Adding {len(chunks)} chunks to index...")
            dist = self._dist()
            if dist is not None:
                # contiguous row shards: rank r keeps rows [lo, hi) of this batch on its GPU
                if self._shard_map is None:
                    self._shard_map = _shard.ShardMap(dist.get_world_size())
                lo, hi = _shard.shard_slice(len(chunks), dist.get_world_size(), dist.get_rank())
                self._shard_map.add_batch(len(col.ids), len(chunks))
                if hi > lo:
                    col.append_device(emb[lo:hi].contiguous())
            else:
                col.append_device(emb)
            col.ids.extend(chunk.chunk_id for chunk in chunks)
            col.documents.extend(chunk.text for chunk in chunks)
            col.metadatas.extend(self._chunk_metadata(chunk, fields) for chunk in chunks)
            if self.persist_directory and dist is None:
                self.persist()
            logger.info(f"Index created successfully! Total documents: {col.count()}")
        except (ValueError, nat.NativeError):
            raise
        except Exception as e:
            logger.error(f"Failed to add documents to collection: {e}")
            raise

    # -- search --------------------------------------------------------------------------------
    def _topk_device(self, q32, top_k: int, allowed_rows=None):
        """q32: cuda fp32 [nq, dim] -> (scores [nq,k] fp32, rows [nq,k] int64 host-sidecar rows), cuda."""
        import torch
        col = self.collection
        dist = self._dist()
        nq = q32.shape[0]
        q16 = nat.queries_to_f16(q32, col.slab_type)
        slab, scales, shadow, n = col.slab, col.scales, col.shadow, col.n
        row_map = None
        if allowed_rows is not None:   # metadata filter: scan a gathered sub-slab, map rows back
            row_map = torch.as_tensor(allowed_rows, dtype=torch.int64, device=col.device)
            slab = col.slab[row_map].contiguous()
            scales = col.scales[row_map].contiguous() if scales is not None else None
            shadow = col.shadow[row_map].contiguous() if shadow is not None else None
            n = int(row_map.numel())
        k_scan = top_k
        refine = col.refine_fp32 and shadow is not None
        if refine:
            k_scan = min(nat.MAX_K, max(top_k, top_k * self.refine_factor))
        if n > 0:
            k_loc = min(k_scan, nat.MAX_K)
            s, i = nat.cosine_topk(q16, slab, n, col.dim, k_loc, slab_type=col.slab_type, scales=scales,
                                   workspace=col.workspace(nq, k_loc))
            if refine:
                qn = torch.nn.functional.normalize(q32, p=2, dim=1, eps=1e-12).contiguous()
                nat.rescore_f32(qn, shadow, n, 0, s, i)
            s, i = s[:, :top_k].contiguous(), i[:, :top_k].contiguous()
        else:
            s = torch.full((nq, top_k), float("-inf"), dtype=torch.float32, device=col.device)
            i = torch.full((nq, top_k), -1, dtype=torch.int64, device=col.device)
        if row_map is not None:
            i = torch.where(i >= 0, row_map[i.clamp(min=0)], i)
        if dist is not None:
            # one RCCL all-gather of the per-shard lists, then the k-way merge kernel on every rank;
            # wire ids carry the rank so the merged order is (score desc, rank asc, local row asc)
            s, i = _shard.allgather_merge(dist, s, _shard.tag(i, dist.get_rank()), top_k, nat.merge_topk)
            ih = i.cpu().numpy()
            rows = np.full(ih.shape, -1, dtype=np.int64)
            for a in range(ih.shape[0]):
                for b in range(ih.shape[1]):
                    if ih[a, b] >= 0:
                        rows[a, b] = self._shard_map.global_row(*_shard.untag(int(ih[a, b])))
            i = torch.from_numpy(rows).to(col.device)
        return s, i

    def _filter_rows(self, where: Optional[dict], where_document: Optional[dict]):
        if not where and not where_document:
            return None
        col = self.collection
        if self._dist() is not None:
            raise NotImplementedError("metadata filters are not supported on a sharded store")
        keep = []
        for row, (meta, doc) in enumerate(zip(col.metadatas, col.documents)):
            ok = True
            for key, want in (where or {}).items():
                if isinstance(want, dict):  # {"$eq": v} / {"$ne": v} / {"$in": [...]}
                    (op, val), = want.items()
                    have = meta.get(key)
                    ok &= {"$eq": have == val, "$ne": have != val,
                           "$in": have in val if isinstance(val, (list, tuple)) else False}.get(op, False)
                else:
                    ok &= meta.get(key) == want
            for op, val in (where_document or {}).items():
                if op == "$contains":
                    ok &= val in doc
                elif op == "$not_contains":
                    ok &= val not in doc
            if ok:
                keep.append(row)
        return keep

    def search(self, query_embedding, top_k: int = 5, where: Optional[dict] = None,
               where_document: Optional[dict] = None) -> Dict[str, Any]:
        """Nearest chunks for ONE query.  Returns {'ids','documents','metadatas','distances'} as
        lists of one list each; distances are cosine distances (1 - cos), ascending."""
        if self.collection is None:
            raise ValueError("No collection available. Create index first.")
        if self.collection.count() == 0:
            logger.warning("Collection is empty. No results to return.")
            return {k: [[]] for k in _EMPTY}
        top_k = min(top_k, self.collection.count())
        # the reference flattens whatever it is given into one vector (indexing.py:156-168)
        if isinstance(query_embedding, np.ndarray):
            flat = query_embedding.reshape(-1)
        else:
            flat = np.asarray(list(query_embedding), dtype=np.float32).reshape(-1)
        try:
            res = self.search_batch(flat.reshape(1, -1), top_k, where=where, where_document=where_document)
            return {key: [res[key][0]] for key in ('ids', 'documents', 'metadatas', 'distances')}
        except Exception as e:
            logger.error(f"Search failed: {e}")
            raise

    def search_batch(self, query_embeddings, top_k: int = 5, where: Optional[dict] = None,
                     where_document: Optional[dict] = None) -> Dict[str, Any]:
        """Many queries per launch: query_embeddings fp32 [nq, d] (numpy or cuda tensor).
        Same dict as ``search`` with one inner list per query."""
        import torch
        if self.collection is None:
            raise ValueError("No collection available. Create index first.")
        col = self.collection
        nq = len(query_embeddings)
        if col.count() == 0 or nq == 0:
            return {k: [[] for _ in range(max(nq, 1))] for k in _EMPTY}
        top_k = min(top_k, col.count())
        if top_k > nat.MAX_K:
            raise ValueError(f"top_k {top_k} exceeds the scan kernel's limit of {nat.MAX_K}")
        if isinstance(query_embeddings, torch.Tensor):
            q32 = query_embeddings.to(device=col.device, dtype=torch.float32).contiguous()
        else:
            q32 = torch.from_numpy(np.ascontiguousarray(query_embeddings, dtype=np.float32)).to(col.device)
        if q32.shape[1] != col.dim:
            raise ValueError(f"Query dimension {q32.shape[1]} doesn't match the index dimension {col.dim}")
        allowed = self._filter_rows(where, where_document)
        if allowed is not None and len(allowed) == 0:
            return {k: [[] for _ in range(nq)] for k in _EMPTY}
        scores, rows = self._topk_device(q32, top_k, allowed)
        sh, rh = scores.cpu().numpy(), rows.cpu().numpy()
        out = {'ids': [], 'documents': [], 'metadatas': [], 'distances': []}
        for a in range(nq):
            valid = [b for b in range(top_k) if rh[a, b] >= 0]
            out['ids'].append([col.ids[rh[a, b]] for b in valid])
            out['documents'].append([col.documents[rh[a, b]] for b in valid])
            out['metadatas'].append([col.metadatas[rh[a, b]] for b in valid])
            out['distances'].append([float(np.float32(1.0) - sh[a, b]) for b in valid])
        return out

    # -- management ----------------------------------------------------------------------------
    def delete_collection(self):
        """Delete the collection (frees the slab; removes persisted files)."""
        if self.collection:
            self.collection = None
            self._shard_map = None
            if self.persist_directory:
                for path in self._persist_paths():
                    if os.path.exists(path):
                        os.remove(path)
            logger.info(f"Deleted collection: {self.collection_name}")

    def reset_collection(self):
        """Reset the collection (delete and re-initialise)."""
        self.delete_collection()
        self._initialize_collection()

    def get_stats(self) -> Dict[str, Any]:
        """Collection statistics."""
        if self.collection is None:
            return {"status": "empty", "count": 0}
        try:
            col = self.collection
            return {"name": self.collection_name, "count": col.count(), "metadata": col.metadata,
                    "index_dtype": col.index_dtype, "dimension": col.dim, "rows_on_this_gpu": col.n,
                    "slab_bytes": int(col.n * (col.pdim or 0) * (1 if col.slab_type == nat.SLAB_I8 else 2))}
        except Exception as e:
            logger.error(f"Failed to get stats: {e}")
            return {"status": "error", "error": str(e)}
