"""Vector store: a flat, HBM-resident fp16 / int8 slab searched exactly on the MI355X.

Drop-in for the reference's ChromaDB-backed ``VectorStore`` (/root/reference/rag/indexing.py:14-211):
same constructor config keys, same method names, argument meaning, return shapes and exceptions.
What changed underneath: ``collection.add`` -> one ``crs::slab_append`` launch per batch (rows are
L2-normalised, cast to fp16 or int8+scale and appended to a device slab); ``collection.query`` ->
``crs::cosine_topk`` (exact brute-force scan + top-k; include/crs_hip.h, csrc/torch_ops.cpp).  Documents,
ids and metadata stay on the host, indexed by row.

New, additive surface (all keys absent from the reference config.json default so that it works unmodified):
  ``search_batch``            many queries per launch
  ``index_dtype``             'fp16' (default) | 'int8' (per-row scale)
  ``refine_fp32``             (default ON) keep an fp32 shadow of the rows (4 x dim bytes per row beside the fp16 / int8
                              slab), over-fetch ``refine_overfetch`` (32) candidates and re-rank them in fp32: the ranking an
                              fp32 store such as the reference's returns (rag/indexing.py:114-119,171-176).  False = the
                              plain fp16 / int8 ranking, no shadow
  ``refine_exact``            'auto' (default) | True | False: every re-ranked list carries a per-query PROOF that it is the
                              fp32 top-k of all rows (crs::refine_f32_cert: k-th fp32 score > k'-th slab score + a measured
                              error bound); unproven queries (near-ties deeper than the over-fetch, e.g. near-duplicate
                              chunks) are escalated on the device (crs::escalate_exact: one more sweep lists every row that
                              can still rank, fp32 re-rank of the list).  'auto' escalates on fp16 slabs; on int8 slabs the
                              bound (~1e-2 for 768-d rows) is wider than typical score gaps, so the certificate rarely
                              holds at k' = 32 and escalating would cost a second sweep for most batches: int8 stays
                              EMPIRICAL (re-rank only) unless refine_exact=True.  ``last_exactness`` reports the counts
  ``num_gpus`` / ``devices``  ONE process driving N devices: contiguous row shards, per-device scans, partial lists
                              copied to the first device and merged there -- RAGPipeline stays one object
  ``sharded``                 SPMD (one process per GPU, torch.distributed): each rank keeps a row shard; ONE RCCL
                              all-gather of the per-shard wire blocks + merge on every rank
``where`` / ``where_document`` filters work on every layout (the sidecars are replicated; each shard scans the
allowed rows it owns).  ``top_k`` is unlimited as in the reference: above the scan kernels' 64 the shard is
scored by the library's GEMM kernel and selected with a device sort.

There is no CPU fallback: without a GPU or without the native libraries every search raises.
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


class _Shard:
    """The rows one device holds: slab (+ scales, + fp32 shadow) and, per local row, the global sidecar row."""

    def __init__(self, index_dtype: str, refine_fp32: bool, device):
        import torch
        self.index_dtype, self.refine_fp32, self.device = index_dtype, refine_fp32, device
        # largest |stored row - fp32 row|_2 of this shard, raised by every crs::slab_append: the row term of the
        # exactness certificate (csrc/exact.hip).  Lives on the device; read back lazily (row_err_max()).
        self.row_err = torch.zeros(1, dtype=torch.float32, device=device)
        self._row_err_host = None
        self._exact_ws = {}            # (nq, cap) -> uint8 workspace of the certificate / escalation
        self.dim: Optional[int] = None
        self.pdim: Optional[int] = None
        self.n = 0
        self.capacity = 0
        self.slab = None               # torch [capacity, pdim] fp16 | int8
        self.scales = None             # torch [capacity] fp32 (int8 only)
        self.shadow = None             # torch [capacity, dim] fp32 (refine_fp32 only)
        self.rows_global = None        # torch [capacity] int64: sidecar row of each local row
        self.identity = True           # rows_global[i] == i for every row (single shard, unsharded): no mapping needed
        self._workspace = None

    @property
    def slab_type(self) -> int:
        return nat.SLAB_I8 if self.index_dtype == "int8" else nat.SLAB_F16

    def _grow(self, old, shape, dtype):
        import torch
        new = torch.zeros(shape, dtype=dtype, device=self.device)
        if old is not None and self.n:
            new[: self.n].copy_(old[: self.n])
        return new

    def reserve(self, rows: int, dim: int):
        import torch
        if self.dim is None:
            self.dim, self.pdim = dim, nat.padded_dim(dim, self.slab_type)
        elif dim != self.dim:
            raise ValueError(f"Embedding dimension {dim} doesn't match the index dimension {self.dim}")
        if rows <= self.capacity:
            return
        cap = max(rows, int(self.capacity * 1.5) + 1024)
        self.slab = self._grow(self.slab, (cap, self.pdim), torch.int8 if self.slab_type == nat.SLAB_I8 else torch.float16)
        if self.slab_type == nat.SLAB_I8:
            self.scales = self._grow(self.scales, (cap,), torch.float32)
        if self.refine_fp32:
            self.shadow = self._grow(self.shadow, (cap, self.dim), torch.float32)
        self.rows_global = self._grow(self.rows_global, (cap,), torch.int64)
        self.capacity = cap

    def append(self, emb, first_global_row: int):
        """emb: fp32 [m, dim] on this device (contiguous) -> m more rows; they are sidecar rows first_global_row.."""
        import torch
        m, dim = emb.shape
        self.reserve(self.n + m, dim)
        if m == 0:
            return
        with torch.cuda.device(self.device):
            nat.slab_append_f32(emb, self.slab, self.n, self.slab_type, scales=self.scales, shadow=self.shadow,
                                row_err=self.row_err)
            self._row_err_host = None
            self.rows_global[self.n: self.n + m] = torch.arange(first_global_row, first_global_row + m, device=self.device)
        if first_global_row != self.n:
            self.identity = False
        self.n += m

    def row_err_max(self) -> float:
        """The tracked row error as a host float (one 4-byte D2H after an append, cached until the next one)."""
        if self._row_err_host is None:
            self._row_err_host = float(self.row_err.item())
        return self._row_err_host

    def exact_workspace(self, nq: int, cap: int):
        import torch
        key = (nq, cap)
        ws = self._exact_ws.get(key)
        if ws is None:
            if len(self._exact_ws) > 8:
                self._exact_ws.clear()
            ws = self._exact_ws[key] = torch.empty(nat.exact_workspace_bytes(nq, cap), dtype=torch.uint8, device=self.device)
        return ws

    def workspace(self, nq: int, k: int, n_rows: int):
        import torch
        need = nat.scan_workspace_bytes(nq, self.dim, k, max(n_rows, 1))
        if self._workspace is None or self._workspace.numel() < need:
            self._workspace = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._workspace


class SlabCollection:
    """What ``VectorStore.collection`` exposes (the retriever reads ``.metadata`` and the harness ``.count()``,
    reference rag/retrieval.py:48-50).  Owns the per-device shards and the host sidecars."""

    def __init__(self, name: str, index_dtype: str, refine_fp32: bool, devices: Sequence):
        self.name = name
        self.metadata = {"hnsw:space": "cosine"}
        self.index_dtype = index_dtype
        self.refine_fp32 = refine_fp32
        self.shards: List[_Shard] = [_Shard(index_dtype, refine_fp32, d) for d in devices]
        self.ids: List[str] = []
        self.documents: List[str] = []
        self.metadatas: List[dict] = []

    @property
    def slab_type(self) -> int:
        return nat.SLAB_I8 if self.index_dtype == "int8" else nat.SLAB_F16

    # first-shard views (single-device stores: the whole index)
    @property
    def device(self):
        return self.shards[0].device

    @property
    def dim(self):
        return self.shards[0].dim

    @property
    def pdim(self):
        return self.shards[0].pdim

    @property
    def n(self) -> int:
        return sum(s.n for s in self.shards)

    @property
    def slab(self):
        return self.shards[0].slab

    @property
    def scales(self):
        return self.shards[0].scales

    @property
    def shadow(self):
        return self.shards[0].shadow

    def count(self) -> int:
        return len(self.ids)


class VectorStore:
    """Vector store on an HBM slab.  Handles storage, indexing and exact similarity search."""

    def __init__(self, config: dict):
        self.collection_name = config.get('collection_name', 'rag_documents')
        self.persist_directory = config.get('persist_directory', None)
        # additive knobs (absent from the reference config.json -> defaults)
        self.index_dtype = config.get('index_dtype', 'fp16')
        if self.index_dtype not in ('fp16', 'int8'):
            raise ValueError(f"index_dtype must be 'fp16' or 'int8', got {self.index_dtype!r}")
        refine = config.get('refine_fp32', 'auto')
        self.refine_fp32 = True if refine == 'auto' else bool(refine)
        self.refine_overfetch = int(config.get('refine_overfetch', 32))
        exact = config.get('refine_exact', 'auto')
        if exact not in ('auto', True, False):
            raise ValueError(f"refine_exact must be 'auto', True or False, got {exact!r}")
        self.refine_exact = exact
        self.exact_cap = int(config.get('exact_cap', nat.EXACT_CAP))
        # certificate outcome of the most recent search: queries proven exact by the over-fetch alone / escalated to
        # exactness on the device / left unproven (escalation off, or a band of more than EXACT_MAX_CAP near-identical rows)
        self.last_exactness = {"queries": 0, "certified": 0, "escalated": 0, "unproven": 0}
        self.sharded = bool(config.get('sharded', False))
        self._device = config.get('device', None)
        self._devices_cfg = config.get('devices', None)
        self.num_gpus = int(config.get('num_gpus', 0) or 0)
        if self.sharded and (self.num_gpus > 1 or self._devices_cfg):
            raise ValueError("'sharded' (one process per GPU) and 'num_gpus'/'devices' (one process, N devices) are exclusive")
        if self.sharded and self.persist_directory:
            raise NotImplementedError("persist_directory is not supported with sharded=True (each rank holds only its rows); "
                                      "use num_gpus for a persisted multi-GPU store")
        self.client = self  # the reference keeps a chromadb client here; nothing else reads it
        self.collection: Optional[SlabCollection] = None
        self._wire = {}     # (nq, k) -> WireBlock (SPMD exchange buffers)
        self._initialize_collection()

    # -- helpers -------------------------------------------------------------------------------
    def _torch_devices(self):
        import torch
        nat.require_gpu()
        if self._devices_cfg:
            return [torch.device(d) for d in self._devices_cfg]
        if self.num_gpus > 1:
            have = torch.cuda.device_count()
            if self.num_gpus > have:
                raise nat.NativeError(f"num_gpus={self.num_gpus} but only {have} device(s) are visible")
            return [torch.device("cuda", g) for g in range(self.num_gpus)]
        if self._device is not None and str(self._device).startswith("cuda"):
            d = torch.device(self._device)
            return [d if d.index is not None else torch.device("cuda", torch.cuda.current_device())]
        return [torch.device("cuda", torch.cuda.current_device())]

    def _dist(self):
        if not self.sharded:
            return None
        import torch.distributed as dist
        return dist if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 else None

    def _new_collection(self) -> SlabCollection:
        return SlabCollection(self.collection_name, self.index_dtype, self.refine_fp32, self._torch_devices())

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
        import torch
        try:
            z = np.load(slab_path, allow_pickle=False)
            with open(docs_path) as fh:
                side = json.load(fh)
            n, dim, dtype = int(z["n"]), int(z["dim"]), str(z["index_dtype"])
            rows = {"slab": z["slab"], "scales": z["scales"] if dtype == "int8" else None,
                    "shadow": z["shadow"] if "shadow" in z.files else None}
            if len(side["ids"]) != n or rows["slab"].shape[0] != n:
                raise ValueError(f"slab holds {rows['slab'].shape[0]} rows, sidecar {len(side['ids'])}, header {n}")
        except Exception as e:
            # a truncated / foreign file must not silently become an empty store that the next add overwrites
            raise RuntimeError(f"persisted collection '{self.collection_name}' under {self.persist_directory} is unreadable: {e}") from e
        self.index_dtype = dtype
        refine = self.refine_fp32 and rows["shadow"] is not None
        if self.refine_fp32 and not refine:
            logger.warning("persisted collection has no fp32 shadow; refine_fp32 disabled")
        col = SlabCollection(self.collection_name, dtype, refine, self._torch_devices())
        for g, (lo, hi) in enumerate(_shard.batch_slices(n, len(col.shards))):
            sh = col.shards[g]
            sh.reserve(hi - lo, dim)
            if hi > lo:
                sh.slab[: hi - lo].copy_(torch.from_numpy(rows["slab"][lo:hi]))
                if rows["scales"] is not None:
                    sh.scales[: hi - lo].copy_(torch.from_numpy(rows["scales"][lo:hi]))
                if refine:
                    sh.shadow[: hi - lo].copy_(torch.from_numpy(rows["shadow"][lo:hi]))
                sh.rows_global[: hi - lo] = torch.arange(lo, hi, device=sh.device)
            sh.n = hi - lo
            sh.identity = (lo == 0)
            # the certificate's row term: the persisted maximum, or the analytic worst case for files written before it existed
            sh.row_err.fill_(float(z["row_err_max"]) if "row_err_max" in z.files
                             else nat.exact_row_error_bound(dim, nat.SLAB_I8 if dtype == "int8" else nat.SLAB_F16))
        col.ids, col.documents, col.metadatas = side["ids"], side["documents"], side["metadatas"]
        self.collection = col
        logger.info(f"Loaded existing collection: {self.collection_name} ({col.count()} rows)")

    def persist(self):
        """Write the slab + sidecars under persist_directory (the PersistentClient analogue): rows in sidecar order
        whatever the device layout, each file written to a temporary name and moved into place."""
        if not self.persist_directory or self.collection is None:
            return
        import torch
        os.makedirs(self.persist_directory, exist_ok=True)
        col = self.collection
        slab_path, docs_path = self._persist_paths()
        order = torch.cat([s.rows_global[: s.n].cpu() for s in col.shards]).argsort().numpy()

        def gather(name):
            return torch.cat([getattr(s, name)[: s.n].cpu() for s in col.shards]).numpy()[order]

        arrays = {"slab": gather("slab"), "n": np.int64(col.n), "dim": np.int64(col.dim), "index_dtype": np.str_(col.index_dtype),
                  "row_err_max": np.float32(max(sh.row_err_max() for sh in col.shards))}
        if col.slab_type == nat.SLAB_I8:
            arrays["scales"] = gather("scales")
        if col.refine_fp32:
            arrays["shadow"] = gather("shadow")
        tmp = slab_path + ".tmp.npz"
        np.savez(tmp, **arrays)
        os.replace(tmp, slab_path)
        tmp = docs_path + ".tmp"
        with open(tmp, "w") as fh:
            json.dump({"ids": col.ids, "documents": col.documents, "metadatas": col.metadatas}, fh)
        os.replace(tmp, docs_path)

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
                self.collection = self._new_collection()
                logger.info(f"Created new collection: {self.collection_name}")
            col = self.collection
            if isinstance(embeddings, torch.Tensor):
                emb = embeddings.to(dtype=torch.float32)
            else:
                emb = torch.from_numpy(np.ascontiguousarray(embeddings, dtype=np.float32))
            if emb.ndim != 2:
                raise ValueError(f"embeddings must be 2-D, got shape {tuple(emb.shape)}")
            logger.info(f"Adding {len(chunks)} chunks to index...")
            start = len(col.ids)
            dist = self._dist()
            if dist is not None:      # SPMD: this rank keeps rows [lo, hi) of the batch on its GPU
                lo, hi = _shard.shard_slice(len(chunks), dist.get_world_size(), dist.get_rank())
                col.shards[0].append(emb[lo:hi].to(col.shards[0].device).contiguous(), start + lo)
            else:                     # one process: shard g of the batch goes to device g (one shard: everything)
                for sh, (lo, hi) in zip(col.shards, _shard.batch_slices(len(chunks), len(col.shards))):
                    sh.append(emb[lo:hi].to(sh.device).contiguous(), start + lo)
            col.ids.extend(chunk.chunk_id for chunk in chunks)
            col.documents.extend(chunk.text for chunk in chunks)
            col.metadatas.extend(self._chunk_metadata(chunk, fields) for chunk in chunks)
            if self.persist_directory:
                self.persist()
            logger.info(f"Index created successfully! Total documents: {col.count()}")
        except (ValueError, nat.NativeError):
            raise
        except Exception as e:
            logger.error(f"Failed to add documents to collection: {e}")
            raise

    # -- search --------------------------------------------------------------------------------
    def _escalates(self, sh: _Shard) -> bool:
        """refine_exact 'auto': fp16 slabs escalate unproven queries, int8 slabs stay empirical (module docstring)."""
        if self.refine_exact == 'auto':
            return sh.slab_type == nat.SLAB_F16
        return bool(self.refine_exact)

    def _search_shard(self, sh: _Shard, q32, top_k: int, allowed_t, cap: Optional[int] = None):
        """q32: fp32 [nq, dim] on the shard's device -> (scores [nq, top_k], GLOBAL sidecar rows [nq, top_k], certificate
        status int32 [nq] or None) there.  Nothing here waits for the device."""
        import torch
        nq = q32.shape[0]
        slab, scales, shadow, n = sh.slab, sh.scales, sh.shadow, sh.n
        row_map = None if sh.identity else (sh.rows_global[:n] if n else None)
        if allowed_t is not None and n:   # metadata filter: scan a gathered sub-slab of the allowed rows this shard owns
            local = torch.isin(sh.rows_global[:n], allowed_t.to(sh.device)).nonzero().flatten()
            slab = sh.slab[local].contiguous()
            scales = sh.scales[local].contiguous() if scales is not None else None
            shadow = sh.shadow[local].contiguous() if shadow is not None else None
            row_map, n = sh.rows_global[local], int(local.numel())
        if n == 0:
            return (torch.full((nq, top_k), float("-inf"), dtype=torch.float32, device=sh.device),
                    torch.full((nq, top_k), -1, dtype=torch.int64, device=sh.device), None)
        refine = sh.refine_fp32 and shadow is not None
        status = None
        if top_k > nat.MAX_K:
            s, i = self._topk_large(sh, q32, slab, scales, shadow if refine else None, n, top_k)
        else:
            q16 = nat.queries_to_f16(q32, sh.slab_type)
            k_scan = nat.overfetch(nq, top_k, self.refine_overfetch) if refine else top_k
            s, i = nat.cosine_topk(q16, slab, n, sh.dim, k_scan, slab_type=sh.slab_type, scales=scales,
                                   workspace=sh.workspace(nq, k_scan, n))
            if refine:
                # fp32 re-rank of the k_scan candidates + the proof that no un-fetched row can reach the list; queries
                # without proof are made exact on the device (one more sweep for them; a no-op launch otherwise)
                qn = torch.nn.functional.normalize(q32, p=2, dim=1, eps=1e-12).contiguous()
                cap = cap or self.exact_cap
                ws = sh.exact_workspace(nq, cap)
                s, i, status = nat.refine_f32_cert(qn, q16, shadow, n, 0, i, s, top_k, sh.row_err_max(), sh.slab_type, ws, cap)
                if self._escalates(sh):
                    nat.escalate_exact(qn, q16, slab, shadow, n, 0, top_k, s, i, status, ws, cap, scales=scales)
        if row_map is not None:
            i = torch.where(i >= 0, row_map[i.clamp(min=0)], i)
        return s, i, status

    def _topk_large(self, sh: _Shard, q32, slab, scales, shadow, n: int, top_k: int):
        """top_k above the scan kernels' limit (the reference accepts any n_results, rag/indexing.py:152-153):
        all scores of a row block through the library's GEMM kernel (crs_gemm_f16, fp32 out), device top-k per
        block, final order by two stable sorts (score desc, row asc).  int8 rows are widened per block."""
        import torch
        from rag._encoder import gemm_f16
        q16 = nat.queries_to_f16(q32, nat.SLAB_F16)
        if q16.shape[1] != slab.shape[1]:            # int8 slabs pad rows to 256 elements
            q16 = torch.nn.functional.pad(q16, (0, slab.shape[1] - q16.shape[1]))
        nq = q32.shape[0]
        best_s = torch.empty((nq, 0), dtype=torch.float32, device=sh.device)
        best_i = torch.empty((nq, 0), dtype=torch.int64, device=sh.device)
        block = 1 << 16
        zero = torch.zeros((nq, min(block, n)), dtype=torch.float32, device=sh.device)
        for lo in range(0, n, block):
            hi = min(n, lo + block)
            if shadow is not None:                   # exact fp32 scores when the store keeps the fp32 rows
                sc = torch.nn.functional.normalize(q32, p=2, dim=1, eps=1e-12) @ shadow[lo:hi].T
            else:
                w = slab[lo:hi] if scales is None else (slab[lo:hi].float() * scales[lo:hi, None]).half()
                sc = gemm_f16(q16, w.contiguous(), residual=zero[:, : hi - lo].contiguous(), mode=2)
            ts, ti = sc.topk(min(top_k, hi - lo), dim=1)
            best_s, best_i = torch.cat([best_s, ts], 1), torch.cat([best_i, ti + lo], 1)
            if best_s.shape[1] > 4 * top_k:
                best_s, best_i = self._order(best_s, best_i, top_k)
        s, i = self._order(best_s, best_i, top_k)
        if s.shape[1] < top_k:
            pad = top_k - s.shape[1]
            s = torch.nn.functional.pad(s, (0, pad), value=float("-inf"))
            i = torch.nn.functional.pad(i, (0, pad), value=-1)
        return s, i

    @staticmethod
    def _order(s, i, k: int):
        """(score desc, row asc) via two stable sorts; empty slots (row < 0) last; keep k."""
        import torch
        big = torch.iinfo(torch.int64).max
        o = torch.argsort(torch.where(i >= 0, i, big), dim=1, stable=True)
        s, i = torch.gather(s, 1, o), torch.gather(i, 1, o)
        o = torch.argsort(torch.where(i >= 0, s, float("-inf")), dim=1, descending=True, stable=True)[:, :k]
        return torch.gather(s, 1, o), torch.gather(i, 1, o)

    def _topk_device(self, q32, top_k: int, allowed_rows=None):
        """q32: fp32 [nq, dim] on the first device -> (scores [nq, k] fp32, sidecar rows [nq, k] int64) there."""
        import torch
        col = self.collection
        nq = q32.shape[0]
        allowed_t = None
        if allowed_rows is not None:
            allowed_t = torch.as_tensor(allowed_rows, dtype=torch.int64, device=col.device)
        parts = []
        for sh in col.shards:          # launches are asynchronous: the devices scan their shards concurrently
            with torch.cuda.device(sh.device):
                q = q32 if q32.device == sh.device else q32.to(sh.device, non_blocking=True)
                parts.append(self._search_shard(sh, q, top_k, allowed_t))
        # certificate bookkeeping (the one host wait of a refined search; search_batch reads the results right after anyway):
        # status 2 = an escalated query's band held more rows than the list -- repeat that shard with a longer list
        tally = {"queries": nq, "certified": 0, "escalated": 0, "unproven": 0}
        worst = None
        for g, sh in enumerate(col.shards):
            if parts[g][2] is None:
                continue
            st = parts[g][2].cpu().numpy()
            cap = self.exact_cap
            while (st == 2).any() and cap < nat.EXACT_MAX_CAP:
                cap = min(nat.EXACT_MAX_CAP, cap * 4)
                with torch.cuda.device(sh.device):
                    q = q32 if q32.device == sh.device else q32.to(sh.device)
                    parts[g] = self._search_shard(sh, q, top_k, allowed_t, cap=cap)
                st = parts[g][2].cpu().numpy()
            if (st == 2).any():
                logger.warning(f"{int((st == 2).sum())} queries have more than {nat.EXACT_MAX_CAP} rows within the error band of "
                               f"their top-{top_k} (near-identical chunks): their lists are the fp32 re-rank of the over-fetch, unproven")
            if not self._escalates(sh):
                st = np.where(st == 1, 2, st)          # not escalated: unproven
            worst = st if worst is None else np.maximum(worst, st)
        if worst is not None:          # a query counts once: by its worst shard
            tally.update(certified=int((worst == 0).sum()), escalated=int((worst == 1).sum()), unproven=int((worst == 2).sum()))
            self.last_exactness = tally
        parts = [(s_, i_) for s_, i_, _ in parts]
        if len(parts) > 1:             # one process, N devices: partial lists to the first device, merge there
            dev0 = col.device
            with torch.cuda.device(dev0):
                gs = torch.stack([s.to(dev0) for s, _ in parts]).contiguous()
                gi = torch.stack([i.to(dev0) for _, i in parts]).contiguous()
                s, i = nat.merge_topk(gs, gi, top_k) if top_k <= nat.MAX_K else self._order(
                    gs.permute(1, 0, 2).reshape(nq, -1), gi.permute(1, 0, 2).reshape(nq, -1), top_k)
        else:
            s, i = parts[0]
        dist = self._dist()
        if dist is not None:           # SPMD: ONE all-gather of the wire blocks, k-way merge on every rank
            if top_k > nat.MAX_K:
                raise ValueError(f"top_k {top_k} > {nat.MAX_K} is not supported on an SPMD-sharded store")
            key = (nq, top_k, dist.get_world_size())
            wb = self._wire.get(key)
            if wb is None:
                wb = self._wire[key] = nat.WireBlock(nq, top_k, col.device, dist.get_world_size())
            wb.scores.copy_(s)
            wb.ids.copy_(i)
            s, i = _shard.allgather_merge(dist, wb.buf, wb.gathered, nq, top_k, top_k, nat.merge_topk_wire)
        return s, i

    def _filter_rows(self, where: Optional[dict], where_document: Optional[dict]):
        if not where and not where_document:
            return None
        col = self.collection
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

    def engine_view(self):
        """This store as ONE device shard for rag._engine.RetrievalEngine, or None when its layout needs the general path
        (several devices, SPMD sharding, a non-identity row map)."""
        col = self.collection
        if col is None or len(col.shards) != 1 or self.sharded or col.n == 0:
            return None
        sh = col.shards[0]
        if not sh.identity:
            return None
        from rag._engine import ShardView
        return ShardView(sh.slab, sh.scales, sh.shadow if sh.refine_fp32 else None, sh.n, sh.dim, sh.slab_type, 0, sh.row_err_max())

    def rows_f32(self, rows):
        """The fp32 rows the store kept for these sidecar rows (numpy [len(rows), dim]; the normalised encoder output of the
        chunks as indexed), or None when it keeps none / the layout is not a single identity shard."""
        import torch
        col = self.collection
        if col is None or len(col.shards) != 1 or not col.shards[0].identity or col.shards[0].shadow is None:
            return None
        sh = col.shards[0]
        idx = torch.as_tensor(np.asarray(rows, dtype=np.int64), device=sh.device)
        return sh.shadow[idx].cpu().numpy()

    def search_rows(self, query_embeddings, top_k: int):
        """search_batch without the sidecar lookup: (scores fp32 [nq, k], sidecar rows int64 [nq, k]) as numpy, best first,
        -1 rows = fewer than k hits.  (retrieve_batch builds its dicts straight from these.)"""
        import torch
        if self.collection is None:
            raise ValueError("No collection available. Create index first.")
        col = self.collection
        nq = len(query_embeddings)
        if col.count() == 0 or nq == 0:
            return np.zeros((nq, 0), dtype=np.float32), np.zeros((nq, 0), dtype=np.int64)
        top_k = min(top_k, col.count())
        if isinstance(query_embeddings, torch.Tensor):
            q32 = query_embeddings.to(device=col.device, dtype=torch.float32).contiguous()
        else:
            q32 = torch.from_numpy(np.ascontiguousarray(query_embeddings, dtype=np.float32)).to(col.device)
        if q32.shape[1] != col.dim:
            raise ValueError(f"Query dimension {q32.shape[1]} doesn't match the index dimension {col.dim}")
        scores, rows = self._topk_device(q32, top_k, None)
        return scores.cpu().numpy(), rows.cpu().numpy()

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
        dist = (np.float32(1.0) - sh).astype(np.float64)          # one vectorised pass; float(np.float32) per hit was the cost
        ids_l, docs_l, metas_l = col.ids, col.documents, col.metadatas
        out = {'ids': [], 'documents': [], 'metadatas': [], 'distances': []}
        for a in range(nq):
            valid = [r for r in rh[a].tolist() if r >= 0]
            out['ids'].append([ids_l[r] for r in valid])
            out['documents'].append([docs_l[r] for r in valid])
            out['metadatas'].append([metas_l[r] for r in valid])
            out['distances'].append(dist[a, : len(valid)].tolist())
        return out

    # -- management ----------------------------------------------------------------------------
    def delete_collection(self):
        """Delete the collection (frees the slab; removes persisted files)."""
        if self.collection:
            self.collection = None
            self._wire = {}
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
            elem = 1 if col.slab_type == nat.SLAB_I8 else 2
            return {"name": self.collection_name, "count": col.count(), "metadata": col.metadata,
                    "index_dtype": col.index_dtype, "dimension": col.dim, "rows_on_this_gpu": col.shards[0].n,
                    "rows_per_device": [s.n for s in col.shards],
                    "slab_bytes": int(sum(s.n for s in col.shards) * (col.pdim or 0) * elem)}
        except Exception as e:
            logger.error(f"Failed to get stats: {e}")
            return {"status": "error", "error": str(e)}
