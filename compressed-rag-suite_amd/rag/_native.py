"""Bindings of the native library.

Compute calls go through the PyTorch-ROCm custom ops ``torch.ops.crs.*`` that csrc/torch_ops.cpp registers with
TORCH_LIBRARY (libcrs_torch.so; tensors in, current HIP stream) over the C ABI of libcrs_hip.so
(include/crs_hip.h, include/crs_encoder.h).  The C ABI itself is bound with ctypes for the host-side
queries (row padding, workspace sizes, plan description, the timing hook) and stays the drop-in boundary for
non-torch hosts.  There is no CPU fallback anywhere in this module: if a shared library is
missing, or there is no GPU, the calls raise.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_size_t, c_void_p, POINTER, byref

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CRS_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "csrc", "libcrs_hip.so")  # override: A/B builds
TORCH_LIB_PATH = os.path.join(os.path.dirname(LIB_PATH), "libcrs_torch.so")

SLAB_F16 = 0
SLAB_I8 = 1
MAX_K = 64
EXACT_MAX_CAP = 13312      # CRS_EXACT_MAX_CAP: longest per-query row list of crs_escalate_exact
EXACT_CAP = 1024           # default list length (12 KB of LDS per query in the re-rank)

# name -> (restype, argtypes); mirrors include/crs_hip.h one to one
_SIGNATURES = {
    "crs_last_error": (c_char_p, []),
    "crs_abi_version": (c_int, []),
    "crs_padded_dim": (c_int, [c_int]),
    "crs_row_elems": (c_int, [c_int, c_int]),
    "crs_slab_append_f32": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                    c_int64, c_void_p, c_void_p]),
    "crs_queries_to_f16": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "crs_scan_workspace_bytes": (c_int, [c_int, c_int, c_int, c_int64, POINTER(c_size_t)]),
    "crs_cosine_topk": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int64, c_int,
                                c_int64, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "crs_merge_topk": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                               c_void_p]),
    "crs_rescore_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int64, c_int64, c_int, c_void_p,
                                c_void_p, c_void_p]),
    "crs_score_rows_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "crs_refine_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int64, c_int64, c_void_p, c_int, c_int,
                               c_void_p, c_void_p, c_void_p]),
    "crs_exact_workspace_bytes": (c_size_t, [c_int, c_int]),
    "crs_exact_row_error_bound": (c_float, [c_int, c_int]),
    "crs_refine_f32_cert": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int64, c_int64, c_void_p, c_void_p,
                                    c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "crs_escalate_exact": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                   c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "crs_wire_bytes": (c_size_t, [c_int, c_int]),
    "crs_wire_scores_offset": (c_size_t, [c_int, c_int]),
    "crs_merge_topk_wire": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "crs_scan_plan_describe": (c_int, [c_int, c_int, c_int, c_int64, c_int, ctypes.c_char_p, c_size_t]),
    "crs_stream_create_cu_masked": (c_int, [c_int, c_int, POINTER(c_void_p)]),
    "crs_stream_destroy": (c_int, [c_void_p]),
    "crs_time_cosine_topk": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int64,
                                     c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_int,
                                     POINTER(c_float), POINTER(c_float)]),
}

_lib = None


class NativeError(RuntimeError):
    """A libcrs_hip.so call failed (the message comes from crs_last_error())."""


def register_signatures(table: dict) -> None:
    """Let sibling modules (the encoder binding) add their entry points before load()."""
    _SIGNATURES.update(table)
    if _lib is not None:
        for name, (res, args) in table.items():
            fn = getattr(_lib, name)
            fn.restype, fn.argtypes = res, args


def load() -> ctypes.CDLL:
    """dlopen the in-tree library; raises loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"or `make -C {os.path.dirname(LIB_PATH)}` -- this path has no CPU fallback")
        # torch first: it ships its own HIP runtime and both must resolve to ONE libamdhip64 in the
        # process (loading ours first leaves the kernels on a runtime that sees no device)
        import torch  # noqa: F401
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError = header/library mismatch
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


_ops = None


def ops():
    """``torch.ops.crs`` -- the PyTorch custom ops of libcrs_torch.so (loaded once; raises when it has not been built)."""
    global _ops
    if _ops is None:
        load()                                  # the C-ABI library first (libcrs_torch.so links against it by file name)
        if not os.path.exists(TORCH_LIB_PATH):
            raise NativeError(f"{TORCH_LIB_PATH} is missing: build it with `make -C {os.path.dirname(TORCH_LIB_PATH)}` "
                              f"-- this path has no CPU fallback")
        import torch
        torch.ops.load_library(TORCH_LIB_PATH)
        _ops = torch.ops.crs
    return _ops


class _translate:
    """torch custom ops raise RuntimeError (TORCH_CHECK); the callers of this module catch NativeError."""
    def __enter__(self):
        return self

    def __exit__(self, et, ev, tb):
        if et is not None and issubclass(et, RuntimeError) and not issubclass(et, NativeError):
            raise NativeError(str(ev).split("\n")[0]) from ev
        return False


def exported_symbols():
    return sorted(_SIGNATURES)


def check(rc: int) -> None:
    if rc != 0:
        msg = load().crs_last_error()
        raise NativeError(f"libcrs_hip error {rc}: {msg.decode() if msg else '?'}")


def padded_dim(dim: int, slab_type: int = SLAB_F16) -> int:
    """Padded row length of a slab (and of the queries searched against it)."""
    return int(load().crs_row_elems(int(dim), int(slab_type)))


def _stream_ptr():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise NativeError("no ROCm GPU visible: the MI355X vector store has no CPU fallback")


# ----------------------------------------------------------------------------- wrappers
def slab_append_f32(emb, slab, row0: int, slab_type: int, scales=None, shadow=None, row_err=None) -> None:
    """emb: cuda fp32 [n, dim]; slab: cuda fp16/int8 [cap, pdim]; writes rows row0..row0+n.
    row_err: cuda fp32 [1] (zeroed by the owner of the slab) raised to the largest |stored row - fp32 row|_2."""
    if emb.shape[0] == 0:
        return
    with _translate():
        ops().slab_append(emb, slab, scales, shadow, int(row0), row_err)


def queries_to_f16(q32, slab_type: int = SLAB_F16, out=None):
    import torch
    nq, dim = q32.shape
    if out is None:
        out = torch.empty((nq, padded_dim(dim, slab_type)), dtype=torch.float16, device=q32.device)
    if nq:
        with _translate():
            ops().queries_to_f16(q32, out, int(slab_type))
    return out


def scan_workspace_bytes(nq: int, dim: int, k: int, n_rows: int) -> int:
    out = c_size_t(0)
    check(load().crs_scan_workspace_bytes(nq, dim, k, n_rows, byref(out)))
    return int(out.value)


def cosine_topk(q16, slab, n_rows: int, dim: int, k: int, *, slab_type: int = SLAB_F16, scales=None,
                id_base: int = 0, workspace=None, out_scores=None, out_ids=None):
    """q16: cuda fp16 [nq, pdim]; slab: cuda [>= n_rows, pdim]. Returns (scores fp32, ids int64) [nq, k]."""
    import torch
    nq = q16.shape[0]
    need = scan_workspace_bytes(nq, dim, k, n_rows)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=q16.device)
    if out_scores is None:
        out_scores = torch.empty((nq, k), dtype=torch.float32, device=q16.device)
    if out_ids is None:
        out_ids = torch.empty((nq, k), dtype=torch.int64, device=q16.device)
    with _translate():
        ops().cosine_topk_out(q16, slab, scales, int(n_rows), int(dim), int(k), int(id_base), workspace, out_scores, out_ids)
    return out_scores, out_ids


def merge_topk(scores, ids, k_out: int, out_scores=None, out_ids=None):
    """scores/ids: cuda [G, nq, k_in] (fp32 / int64) -> global top-k_out per query."""
    import torch
    g, nq, k_in = scores.shape
    out_s = out_scores if out_scores is not None else torch.empty((nq, k_out), dtype=torch.float32, device=scores.device)
    out_i = out_ids if out_ids is not None else torch.empty((nq, k_out), dtype=torch.int64, device=scores.device)
    with _translate():
        ops().merge_topk_out(scores, ids, int(k_out), out_s, out_i)
    return out_s, out_i


def rescore_f32(q32, shadow, n_rows: int, id_base: int, scores, ids) -> None:
    """In-place variant kept from ABI v1 (C ABI only): re-score all k candidates and re-sort each row."""
    nq, dim = q32.shape
    k = scores.shape[1]
    check(load().crs_rescore_f32(_ptr(q32), nq, dim, _ptr(shadow), n_rows, id_base, k, _ptr(scores),
                                 _ptr(ids), _stream_ptr()))


def score_rows_f32(q32, shadow, n_rows: int, id_base: int, ids):
    """fp32 scores of candidate lists of any length: ids int64 [nq, k] -> fp32 [nq, k] (-inf where ids < 0)."""
    import torch
    scores = torch.full(ids.shape, float("-inf"), dtype=torch.float32, device=q32.device)
    with _translate():
        ops().score_rows_f32_out(q32, shadow, int(n_rows), int(id_base), ids.contiguous(), scores)
    return scores


def refine_f32(q32, shadow, n_rows: int, id_base: int, cand_ids, k_out: int, out_scores=None, out_ids=None):
    """Exact fp32 re-rank of over-fetched candidates: cand_ids int64 [nq, k_in] (what cosine_topk found in the
    fp16/int8 slab) -> the k_out best by <q32, shadow[id - id_base]> (score desc, id asc)."""
    import torch
    nq, dim = q32.shape
    if out_scores is None:
        out_scores = torch.empty((nq, k_out), dtype=torch.float32, device=q32.device)
    if out_ids is None:
        out_ids = torch.empty((nq, k_out), dtype=torch.int64, device=q32.device)
    with _translate():
        ops().refine_f32_out(q32, shadow, int(n_rows), int(id_base), cand_ids, int(k_out), out_scores, out_ids)
    return out_scores, out_ids


def overfetch(nq: int, top_k: int, want: int = 24, n_rows: int = 1 << 62, slab_type: int = SLAB_F16) -> int:
    """Candidates the scan fetches for the fp32 re-rank (never below top_k, never above MAX_K).
    `want` (24) on shards of >= 4 M rows, where the top scores crowd together (10 M x 384: about 1e-3 apart at rank 10; the gap
    from rank 10 to rank 24 is ~9e-3 against a certificate bound of ~6e-4): all 8192 benchmark queries certify with 24 as with 32
    candidates, while with 16 about 3e-4 of them would escalate -- a second sweep of the whole shard each.  Same box, C4 batch:
    1.408 / 1.432 / 1.451 ms at 16 / 24 / 32 (chain slots, tile refine and merge grow with the length).  16 on smaller shards: the
    same gaps are wider (fewer rows in the tail), an escalation sweep is short, and 32 candidates measured 16 % of a 1.25 M-row
    shard's batch (0.301 -> 0.259 ms: one rank of an 8-GPU step).  Launches of more than 64 queries keep to 16 when top_k
    allows: the large-batch kernels (scan_wide.hip) carry a 16-slot chain, and 32 candidates would send those launches to the
    64-query kernel once per query block.  int8 slabs: 16 -- their certificate is too wide to hold at either length
    (0.07 % / 34 % of C5's queries at 16 / 32), the empirical Recall@10 is 1.0 on 8192 queries at both, and the 32-slot chain makes the
    issue-bound int8 kernel 6 % slower (C5 31.3 -> 33.5 k q/s)."""
    want = int(want)
    if n_rows < 4_000_000 or (nq > 64 and top_k <= 16) or slab_type == SLAB_I8:
        want = min(want, 16)
    return min(MAX_K, max(int(top_k), want))


def exact_workspace_bytes(nq: int, cap: int = EXACT_CAP) -> int:
    return int(load().crs_exact_workspace_bytes(int(nq), int(cap)))


def exact_row_error_bound(dim: int, slab_type: int) -> float:
    """Analytic worst case of |stored row - fp32 row|_2 (used when a slab did not track it)."""
    return float(load().crs_exact_row_error_bound(int(dim), int(slab_type)))


def refine_f32_cert(q32, q16, shadow, n_rows: int, id_base: int, cand_ids, cand_scores, k_out: int, row_err_max: float,
                    slab_type: int, exact_ws, cap: int = EXACT_CAP, out_scores=None, out_ids=None, status=None):
    """crs_refine_f32 + the per-query exactness proof: returns (scores [nq, k_out], ids [nq, k_out], status int32 [nq])
    with status 0 = the list is provably the fp32 top-k of all n_rows rows, 1 = not proven (feed escalate_exact)."""
    import torch
    nq = q32.shape[0]
    if out_scores is None:
        out_scores = torch.empty((nq, k_out), dtype=torch.float32, device=q32.device)
    if out_ids is None:
        out_ids = torch.empty((nq, k_out), dtype=torch.int64, device=q32.device)
    if status is None:
        status = torch.empty(nq, dtype=torch.int32, device=q32.device)
    with _translate():
        ops().refine_f32_cert_out(q32, q16, shadow, int(n_rows), int(id_base), cand_ids, cand_scores, int(k_out),
                                  float(row_err_max), int(slab_type), out_scores, out_ids, status, exact_ws, int(cap))
    return out_scores, out_ids, status


def escalate_exact(q32, q16, slab, shadow, n_rows: int, id_base: int, k_out: int, out_scores, out_ids, status, exact_ws,
                   cap: int = EXACT_CAP, scales=None) -> None:
    """Make every status-1 query of a refine_f32_cert result exact, in place, on the current stream (no host sync;
    returns at once on the device when nothing is to do).  status 2 afterwards = list longer than cap."""
    with _translate():
        ops().escalate_exact(q32, q16, slab, scales, shadow, int(n_rows), int(id_base), int(k_out), out_scores, out_ids,
                             status, exact_ws, int(cap))


class WireBlock:
    """One rank's per-shard result in the one-collective wire layout of include/crs_hip.h:
    [ids int64 [nq, k] | scores fp32 [nq, k] | pad].  `.ids` / `.scores` are views into `.buf`, so the
    search kernels write the block in place and `buf` is what the all-gather sends."""

    def __init__(self, nq: int, k: int, device, world: int = 1, gather: bool = False):
        import torch
        lib = load()
        self.nq, self.k, self.world = nq, k, world
        self.nbytes = int(lib.crs_wire_bytes(nq, k))
        off = int(lib.crs_wire_scores_offset(nq, k))
        self.buf = torch.zeros(self.nbytes, dtype=torch.uint8, device=device)
        self.ids = self.buf[:off].view(torch.int64).view(nq, k)
        self.scores = self.buf[off:off + nq * k * 4].view(torch.float32).view(nq, k)
        self.gathered = torch.zeros(world * self.nbytes, dtype=torch.uint8, device=device) if (world > 1 or gather) else None


def merge_topk_wire(gathered, nlists: int, nq: int, k_in: int, k_out: int, out_scores=None, out_ids=None):
    """gathered: cuda uint8 [nlists * crs_wire_bytes(nq, k_in)] (the all-gathered WireBlocks) -> global top-k_out."""
    import torch
    out_s = out_scores if out_scores is not None else torch.empty((nq, k_out), dtype=torch.float32, device=gathered.device)
    out_i = out_ids if out_ids is not None else torch.empty((nq, k_out), dtype=torch.int64, device=gathered.device)
    with _translate():
        ops().merge_topk_wire_out(gathered, int(nlists), int(nq), int(k_in), int(k_out), out_s, out_i)
    return out_s, out_i


def cu_masked_stream(first_cu: int, n_cus: int, device=None):
    """A torch stream whose kernels run on CUs [first_cu, first_cu + n_cus) only (crs_stream_create_cu_masked).  The HIP stream
    lives as long as the process (a handful per engine)."""
    import torch
    out = c_void_p(0)
    check(load().crs_stream_create_cu_masked(int(first_cu), int(n_cus), byref(out)))
    return torch.cuda.ExternalStream(int(out.value), device=device)


def scan_plan_describe(nq: int, dim: int, k: int, n_rows: int, slab_type: int = SLAB_F16) -> str:
    """Kernel family + launch geometry crs_cosine_topk would use (for bench lines and profiles)."""
    buf = ctypes.create_string_buffer(256)
    check(load().crs_scan_plan_describe(nq, dim, k, n_rows, slab_type, buf, 256))
    return buf.value.decode()


def time_cosine_topk(q16, slab, n_rows: int, dim: int, k: int, iters: int, *, slab_type: int = SLAB_F16,
                     scales=None):
    """hipEvent-timed launches on the current stream: returns (ms per scan+merge, ms per scan kernel)."""
    import torch
    nq = q16.shape[0]
    need = scan_workspace_bytes(nq, dim, k, n_rows)
    ws = torch.empty(need, dtype=torch.uint8, device=q16.device)
    out_s = torch.empty((nq, k), dtype=torch.float32, device=q16.device)
    out_i = torch.empty((nq, k), dtype=torch.int64, device=q16.device)
    ms_total, ms_scan = c_float(0), c_float(0)
    check(load().crs_time_cosine_topk(_ptr(q16), nq, dim, slab_type, _ptr(slab), _ptr(scales), n_rows, k,
                                      _ptr(ws), ws.numel(), _ptr(out_s), _ptr(out_i), _stream_ptr(),
                                      iters, byref(ms_total), byref(ms_scan)))
    return float(ms_total.value), float(ms_scan.value)
