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
                              slab), over-fetch ``refine_overfetch`` (24; 16 on shards below 4 M rows) candidates and re-rank them in fp32: the ranking an
                              fp32 store such as the reference's returns (rag/indexing.py:114-119,171-176).  False = the
                              plain fp16 / int8 ranking, no shadow
  ``refine_exact``            'auto' (default) | True | False: every re-ranked list carries a per-query PROOF that it is the
                              fp32 top-k of all rows (crs::refine_f32_cert: k-th fp32 score > k'-th slab score + a measured
                              error bound); unproven queries (near-ties deeper than the over-fetch, e.g. near-duplicate
                              chunks) are escalated on the device (crs::escalate_exact: one more sweep lists every row that
                              can still rank, fp32 re-rank of the list).  'auto' escalates on fp16 slabs; on int8 slabs the
                              bound (~1e-2 for 768-d rows) is wider than typical score gaps, so the certificate rarely
                              holds and escalating would cost a second sweep for most batches: int8 stays
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

    # -- metadata filters: value -> rows, built once and extended as rows arrive (the per-query loop over all metadatas is gone)
    def _inverted(self):
        inv = self.__dict__.setdefault("_inv", {})
        done = self.__dict__.get("_inv_rows", 0)
        for row in range(done, len(self.metadatas)):
            for key, val in self.metadatas[row].items():
                try:
                    inv.setdefault(key, {}).setdefault(val, []).append(row)
                except TypeError:           # unhashable value: never equal to a scalar filter value
                    pass
        self.__dict__["_inv_rows"] = len(self.metadatas)
        return inv

    def _where_rows(self, where: dict, n: int):
        """Rows passing a ChromaDB `where` document: {key: value | {op: value}} (AND over keys), {"$and": [...]},
        {"$or": [...]}; ops $eq $ne $in $nin $gt $gte $lt $lte (chromadb 1.3.0's documented operators; the reference only
        forwards the dict).  Values are looked up in the inverted index: the cost follows the DISTINCT values of a key."""
        every = lambda: np.arange(n, dtype=np.int64)                                          # noqa: E731
        none = lambda: np.zeros(0, dtype=np.int64)                                            # noqa: E731
        rows_of = lambda col, v: np.asarray(col.get(v, []), dtype=np.int64)                   # noqa: E731
        out = None
        for key, want in where.items():
            if key in ("$and", "$or"):
                parts = [self._where_rows(w, n) for w in want]
                if key == "$and":
                    hit = every()
                    for p_ in parts:
                        hit = np.intersect1d(hit, p_, assume_unique=True)
                else:
                    hit = np.unique(np.concatenate(parts)) if parts else none()
            else:
                col = self._inverted().get(key, {})
                op, val = ("$eq", want)
                if isinstance(want, dict):
                    (op, val), = want.items()
                try:
                    if op in ("$eq", "$ne"):
                        if val is None:       # meta.get(key) == None: the rows WITHOUT the key
                            have = [np.asarray(v, dtype=np.int64) for v in col.values()]
                            eq = np.setdiff1d(every(), np.concatenate(have) if have else none())
                        else:
                            eq = rows_of(col, val)
                        hit = eq if op == "$eq" else np.setdiff1d(every(), eq, assume_unique=True)
                    elif op in ("$in", "$nin") and isinstance(val, (list, tuple)):
                        parts = [rows_of(col, v) for v in val]
                        inn = np.unique(np.concatenate(parts)) if parts else none()
                        hit = inn if op == "$in" else np.setdiff1d(every(), inn, assume_unique=True)
                    elif op in ("$gt", "$gte", "$lt", "$lte") and isinstance(val, (int, float)) and not isinstance(val, bool):
                        cmp = {"$gt": lambda x: x > val, "$gte": lambda x: x >= val, "$lt": lambda x: x < val, "$lte": lambda x: x <= val}[op]
                        parts = [np.asarray(r, dtype=np.int64) for v, r in col.items()
                                 if isinstance(v, (int, float)) and not isinstance(v, bool) and cmp(v)]
                        hit = np.sort(np.concatenate(parts)) if parts else none()
                    else:
                        hit = none()
                except TypeError:             # unhashable filter value: nothing can equal it
                    hit = every() if op in ("$ne", "$nin") else none()
            out = hit if out is None else np.intersect1d(out, hit, assume_unique=True)
        return every() if out is None else out

    def _doc_rows(self, cond: dict, base):
        rows = base
        for op, val in cond.items():
            if op == "$contains":
                rows = [r for r in rows if val in self.documents[r]]
            elif op == "$not_contains":
                rows = [r for r in rows if val not in self.documents[r]]
            elif op == "$and":
                for c in val:
                    rows = self._doc_rows(c, rows)
            elif op == "$or":
                keep = set()
                for c in val:
                    keep.update(self._doc_rows(c, rows))
                rows = [r for r in rows if r in keep]
        return rows

    def rows_matching(self, where: Optional[dict], where_document: Optional[dict]):
        """Sidecar rows (ascending numpy int64) that pass the filters the reference forwards to ChromaDB
        (rag/indexing.py:129-130,174).  None = no filter."""
        if not where and not where_document:
            return None
        n = len(self.ids)
        rows = self._where_rows(where, n) if where else np.arange(n, dtype=np.int64)
        if where_document:                    # substring tests have no index: one pass over the surviving documents
            rows = np.asarray(self._doc_rows(where_document, rows.tolist()), dtype=np.int64)
        return rows


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
        self.refine_overfetch = int(config.get('refine_overfetch', 24))
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
        self._filters = {}      # filter key -> {"n": rows when built, "rows": allowed sidecar rows, "shards": {g: compacted sub-slab}}
        self._persisted_rows = None   # rows the append-only files hold (None: nothing / legacy format on disk)
        self._docs_bytes = 0          # length of the sidecar file the header vouches for
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

    # -- persistence (the PersistentClient analogue, reference rag/indexing.py:32-34): APPEND-ONLY files, so that an add
    # costs O(batch), not O(index):
    #     <name>.meta.json    header {format, n, dim, pdim, index_dtype, shadow, row_err_max}; rewritten (tmp + rename) LAST
    #     <name>.slab.bin     rows [n, pdim] fp16 | int8, raw little-endian, in sidecar order
    #     <name>.scales.bin   fp32 [n]            (int8)
    #     <name>.shadow.bin   fp32 [n, dim]       (refine_fp32)
    #     <name>.docs.jsonl   one {"id", "document", "metadata"} line per row
    # The header's n is the truth: bytes or lines past it (a crash between the appends and the header) are ignored and
    # overwritten by the next add; files SHORTER than the header says mean corruption and raise.  The round-2 format
    # (<name>.slab.npz + <name>.docs.json, rewritten whole on every add) is still read.
    def _persist_paths(self):
        base = os.path.join(self.persist_directory, self.collection_name)
        return {k: base + ext for k, ext in (("meta", ".meta.json"), ("slab", ".slab.bin"), ("scales", ".scales.bin"),
                                             ("shadow", ".shadow.bin"), ("docs", ".docs.jsonl"),
                                             ("legacy_slab", ".slab.npz"), ("legacy_docs", ".docs.json"))}

    def _load_rows_into(self, col, n, dim, rows, refine, row_err):
        import torch
        for g, (lo, hi) in enumerate(_shard.batch_slices(n, len(col.shards))):
            sh = col.shards[g]
            sh.reserve(hi - lo, dim)
            if hi > lo:
                sh.slab[: hi - lo].copy_(torch.from_numpy(np.ascontiguousarray(rows["slab"][lo:hi])))
                if rows["scales"] is not None:
                    sh.scales[: hi - lo].copy_(torch.from_numpy(np.ascontiguousarray(rows["scales"][lo:hi])))
                if refine:
                    sh.shadow[: hi - lo].copy_(torch.from_numpy(np.ascontiguousarray(rows["shadow"][lo:hi])))
                sh.rows_global[: hi - lo] = torch.arange(lo, hi, device=sh.device)
            sh.n = hi - lo
            sh.identity = (lo == 0)
            sh.row_err.fill_(row_err)

    def _initialize_collection(self):
        """Re-open a persisted collection if there is one (reference: get_collection, :46-55)."""
        if not self.persist_directory:
            logger.info("Using in-memory (HBM) storage")
            return
        paths = self._persist_paths()
        legacy = not os.path.exists(paths["meta"])
        if legacy and not (os.path.exists(paths["legacy_slab"]) and os.path.exists(paths["legacy_docs"])):
            logger.info(f"Collection '{self.collection_name}' will be created on first add")
            return
        try:
            if legacy:
                z = np.load(paths["legacy_slab"], allow_pickle=False)
                with open(paths["legacy_docs"]) as fh:
                    side = json.load(fh)
                n, dim, dtype = int(z["n"]), int(z["dim"]), str(z["index_dtype"])
                rows = {"slab": z["slab"], "scales": z["scales"] if dtype == "int8" else None,
                        "shadow": z["shadow"] if "shadow" in z.files else None}
                row_err = float(z["row_err_max"]) if "row_err_max" in z.files else None
                ids, docs, metas = side["ids"], side["documents"], side["metadatas"]
                if len(ids) != n or rows["slab"].shape[0] != n:
                    raise ValueError(f"slab holds {rows['slab'].shape[0]} rows, sidecar {len(ids)}, header {n}")
            else:
                with open(paths["meta"]) as fh:
                    meta = json.load(fh)
                n, dim, pdim, dtype = int(meta["n"]), int(meta["dim"]), int(meta["pdim"]), str(meta["index_dtype"])
                elem = np.int8 if dtype == "int8" else np.float16

                def raw(key, np_dtype, width):
                    need = n * width * np.dtype(np_dtype).itemsize
                    if os.path.getsize(paths[key]) < need:
                        raise ValueError(f"{os.path.basename(paths[key])} holds fewer than the header's {n} rows")
                    return np.fromfile(paths[key], dtype=np_dtype, count=n * width).reshape(n, width) if n else np.zeros((0, width), np_dtype)

                rows = {"slab": raw("slab", elem, pdim),
                        "scales": raw("scales", np.float32, 1).reshape(-1) if dtype == "int8" else None,
                        "shadow": raw("shadow", np.float32, dim) if meta.get("shadow") else None}
                row_err = meta.get("row_err_max")
                ids, docs, metas = [], [], []
                with open(paths["docs"], "rb") as fh:
                    blob = fh.read(int(meta["docs_bytes"]))
                if len(blob) < int(meta["docs_bytes"]):
                    raise ValueError("sidecar shorter than the header says")
                for line in blob.decode("utf-8").splitlines():
                    rec = json.loads(line)
                    ids.append(rec["id"]); docs.append(rec["document"]); metas.append(rec["metadata"])
                if len(ids) != n:
                    raise ValueError(f"sidecar holds {len(ids)} rows, header {n}")
                self._docs_bytes = int(meta["docs_bytes"])
        except Exception as e:
            # a truncated / foreign file must not silently become an empty store that the next add overwrites
            raise RuntimeError(f"persisted collection '{self.collection_name}' under {self.persist_directory} is unreadable: {e}") from e
        self.index_dtype = dtype
        refine = self.refine_fp32 and rows["shadow"] is not None
        if self.refine_fp32 and not refine:
            logger.warning("persisted collection has no fp32 shadow; refine_fp32 disabled")
        col = SlabCollection(self.collection_name, dtype, refine, self._torch_devices())
        # the certificate's row term: the persisted maximum, or the analytic worst case for files written before it existed
        if row_err is None:
            row_err = nat.exact_row_error_bound(dim, nat.SLAB_I8 if dtype == "int8" else nat.SLAB_F16)
        self._load_rows_into(col, n, dim, rows, refine, float(row_err))
        col.ids, col.documents, col.metadatas = ids, docs, metas
        self.collection = col
        # appends continue the files only when they hold exactly what this store keeps (same shadow choice); else the next
        # add rewrites them once in the current format
        self._persisted_rows = n if (not legacy and bool(meta.get("shadow")) == bool(refine)) else None
        logger.info(f"Loaded existing collection: {self.collection_name} ({col.count()} rows)")

    def persist(self, first_new_row: Optional[int] = None):
        """Bring the files under persist_directory up to date.  With first_new_row = the files' row count, only rows
        [first_new_row, n) are APPENDED (create_index does this: O(batch)); otherwise everything is rewritten."""
        if not self.persist_directory or self.collection is None:
            return
        import torch
        os.makedirs(self.persist_directory, exist_ok=True)
        col = self.collection
        paths = self._persist_paths()
        n = col.count()
        append = first_new_row is not None and self._persisted_rows == first_new_row and first_new_row <= n
        lo = first_new_row if append else 0
        pdim, dim = col.pdim, col.dim
        elem = 1 if col.slab_type == nat.SLAB_I8 else 2

        def new_rows(name):
            """rows [lo, n) of a per-shard array, in sidecar order"""
            parts, order = [], []
            for sh in col.shards:
                rg = sh.rows_global[: sh.n]
                sel = (rg >= lo).nonzero().flatten() if lo else None
                arr = getattr(sh, name)[: sh.n]
                parts.append((arr if sel is None else arr[sel]).cpu())
                order.append((rg if sel is None else rg[sel]).cpu())
            o = torch.cat(order).argsort()
            return torch.cat(parts)[o].contiguous().numpy()

        def write(key, arr, row_bytes):
            mode = "r+b" if (append and os.path.exists(paths[key])) else "wb"
            with open(paths[key], mode) as fh:
                if mode == "r+b":
                    fh.seek(lo * row_bytes)
                    fh.truncate()                 # drop anything a crashed add left past the header's row count
                fh.write(arr.tobytes())
                fh.flush()
                os.fsync(fh.fileno())

        write("slab", new_rows("slab"), pdim * elem)
        if col.slab_type == nat.SLAB_I8:
            write("scales", new_rows("scales"), 4)
        if col.refine_fp32:
            write("shadow", new_rows("shadow"), dim * 4)
        elif not append and os.path.exists(paths["shadow"]):
            os.remove(paths["shadow"])
        # sidecar lines: keep the bytes the header vouches for (the first `lo` rows), replace the rest
        keep = self._docs_bytes if (append and os.path.exists(paths["docs"])) else 0
        with open(paths["docs"], "r+b" if keep else "wb") as fh:
            fh.seek(keep)
            fh.truncate()
            for r in range(lo if keep else 0, n):
                fh.write((json.dumps({"id": col.ids[r], "document": col.documents[r], "metadata": col.metadatas[r]}) + "\n").encode("utf-8"))
            fh.flush()
            os.fsync(fh.fileno())
            self._docs_bytes = fh.tell()
        meta = {"format": 2, "n": n, "dim": dim, "pdim": pdim, "index_dtype": col.index_dtype, "shadow": bool(col.refine_fp32),
                "row_err_max": max(sh.row_err_max() for sh in col.shards), "docs_bytes": self._docs_bytes}
        tmp = paths["meta"] + ".tmp"
        with open(tmp, "w") as fh:
            json.dump(meta, fh)
            fh.flush()
            os.fsync(fh.fileno())
        os.replace(tmp, paths["meta"])
        for key in ("legacy_slab", "legacy_docs"):       # superseded
            if os.path.exists(paths[key]):
                os.remove(paths[key])
        self._persisted_rows = n

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
                self.persist(first_new_row=start)    # appends rows [start, n) when the files already hold [0, start)
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

    def _filtered_view(self, g: int, sh: _Shard, filt: dict):
        """The allowed rows shard g owns, compacted ONCE per distinct filter into a sub-slab (+ scales, shadow, row map) and
        kept with the filter's cache entry: a filtered query then costs one scan of exactly the allowed rows -- no per-query
        gather, no Python pass over the metadata (the reference hands `where` to ChromaDB, rag/indexing.py:129-130,174)."""
        import torch
        ent = filt["shards"].get(g)
        if ent is None:
            allowed_t = torch.as_tensor(filt["rows"], dtype=torch.int64, device=sh.device)
            n = sh.n
            if sh.identity:
                local = allowed_t[allowed_t < n]
            else:
                local = torch.isin(sh.rows_global[:n], allowed_t).nonzero().flatten()
            ent = {"n": int(local.numel())}
            if ent["n"]:
                ent["slab"] = sh.slab[local].contiguous()
                ent["scales"] = sh.scales[local].contiguous() if sh.scales is not None else None
                ent["shadow"] = sh.shadow[local].contiguous() if sh.shadow is not None else None
                ent["row_map"] = sh.rows_global[local].contiguous()
            filt["shards"][g] = ent
        return ent

    def _search_shard(self, sh: _Shard, q32, top_k: int, filt, cap: Optional[int] = None, g: int = 0):
        """q32: fp32 [nq, dim] on the shard's device -> (scores [nq, top_k], GLOBAL sidecar rows [nq, top_k], certificate
        status int32 [nq] or None) there.  Nothing here waits for the device."""
        import torch
        nq = q32.shape[0]
        slab, scales, shadow, n = sh.slab, sh.scales, sh.shadow, sh.n
        row_map = None if sh.identity else (sh.rows_global[:n] if n else None)
        if filt is not None and n:        # metadata filter: scan the cached, compacted sub-slab of the allowed rows this shard owns
            ent = self._filtered_view(g, sh, filt)
            n = ent["n"]
            if n:
                slab, scales, shadow, row_map = ent["slab"], ent["scales"], ent["shadow"], ent["row_map"]
        if n == 0:
            return (torch.full((nq, top_k), float("-inf"), dtype=torch.float32, device=sh.device),
                    torch.full((nq, top_k), -1, dtype=torch.int64, device=sh.device), None)
        refine = sh.refine_fp32 and shadow is not None
        status = None
        if top_k > nat.MAX_K:
            s, i = self._topk_large(sh, q32, slab, scales, shadow if refine else None, n, top_k)
        else:
            q16 = nat.queries_to_f16(q32, sh.slab_type)
            k_scan = nat.overfetch(nq, top_k, self.refine_overfetch, n, sh.slab_type) if refine else top_k
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
        """top_k above the scan kernels' limit (the reference accepts any n_results, rag/indexing.py:152-153): all slab scores
        of a row block through the library's GEMM kernel (crs_gemm_f16, fp32 out), device top-k per block, order by two
        stable sorts (score desc, row asc).  int8 rows are widened per block.  With the fp32 shadow the slab pass over-fetches
        by half and the candidates are re-scored in fp32 by the library (crs::score_rows_f32) before the final order -- the
        over-fetch re-rank without a certificate (that exists for top_k <= 64)."""
        import torch
        from rag._encoder import gemm_f16
        q16 = nat.queries_to_f16(q32, nat.SLAB_F16)
        if q16.shape[1] != slab.shape[1]:            # int8 slabs pad rows to 256 elements
            q16 = torch.nn.functional.pad(q16, (0, slab.shape[1] - q16.shape[1]))
        nq = q32.shape[0]
        keep = min(n, top_k + max(64, top_k // 2)) if shadow is not None else top_k
        best_s = torch.empty((nq, 0), dtype=torch.float32, device=sh.device)
        best_i = torch.empty((nq, 0), dtype=torch.int64, device=sh.device)
        block = 1 << 16
        zero = torch.zeros((nq, min(block, n)), dtype=torch.float32, device=sh.device)
        for lo in range(0, n, block):
            hi = min(n, lo + block)
            w = slab[lo:hi] if scales is None else (slab[lo:hi].float() * scales[lo:hi, None]).half()
            sc = gemm_f16(q16, w.contiguous(), residual=zero[:, : hi - lo].contiguous(), mode=2)
            ts, ti = sc.topk(min(keep, hi - lo), dim=1)
            best_s, best_i = torch.cat([best_s, ts], 1), torch.cat([best_i, ti + lo], 1)
            if best_s.shape[1] > 4 * keep:
                best_s, best_i = self._order(best_s, best_i, keep)
        s, i = self._order(best_s, best_i, keep)
        if shadow is not None:
            qn = torch.nn.functional.normalize(q32, p=2, dim=1, eps=1e-12).contiguous()
            s, i = self._order(nat.score_rows_f32(qn, shadow, n, 0, i), i, top_k)
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

    def _topk_device(self, q32, top_k: int, filt=None):
        """q32: fp32 [nq, dim] on the first device -> (scores [nq, k] fp32, sidecar rows [nq, k] int64) there.
        filt: a filter cache entry (_filter_entry) or None."""
        import torch
        col = self.collection
        nq = q32.shape[0]
        parts = []
        for g, sh in enumerate(col.shards):   # launches are asynchronous: the devices scan their shards concurrently
            with torch.cuda.device(sh.device):
                q = q32 if q32.device == sh.device else q32.to(sh.device, non_blocking=True)
                parts.append(self._search_shard(sh, q, top_k, filt, g=g))
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
                    parts[g] = self._search_shard(sh, q, top_k, filt, cap=cap, g=g)
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

    def _filter_entry(self, where: Optional[dict], where_document: Optional[dict]):
        """Cache entry of a distinct (where, where_document) pair: the allowed sidecar rows (SlabCollection.rows_matching:
        an inverted metadata index, no per-query pass over the rows) and, filled by the first search, each shard's compacted
        sub-slab.  None = no filter.  Entries die when rows are added; the eight most recent filters are kept."""
        if not where and not where_document:
            return None
        col = self.collection
        try:
            key = json.dumps([where, where_document], sort_keys=True, default=repr)
        except TypeError:
            key = repr((where, where_document))
        ent = self._filters.get(key)
        if ent is None or ent["n"] != col.count():
            ent = {"n": col.count(), "rows": col.rows_matching(where, where_document), "shards": {}}
            self._filters.pop(key, None)
            while len(self._filters) >= 8:
                self._filters.pop(next(iter(self._filters)))
            self._filters[key] = ent
        return ent

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
        filt = self._filter_entry(where, where_document)
        if filt is not None and len(filt["rows"]) == 0:
            return {k: [[] for _ in range(nq)] for k in _EMPTY}
        scores, rows = self._topk_device(q32, top_k, filt)
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
            self._filters = {}
            self._persisted_rows = None
            if self.persist_directory:
                for path in self._persist_paths().values():
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
