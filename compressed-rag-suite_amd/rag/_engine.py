"""Throughput engine of the retrieve path: S query batches in flight over ROLE lanes, each batch replayed from hipGraphs.

What ``ContextRetriever.retrieve_batch`` (large batches) and ``bench.py`` run -- one object, so that the rate bench.py reports
is the rate the plugin surface reaches.  The reference times one query at a time in a Python loop
(/root/reference/evaluation/retrieval/benchmark.py:241-247); it has no batched entry point, so this layer is new.

One BATCH of Qb queries = device segments with the (optional) collectives between them:
    E  encode   token ids [Qb, S] -> fp32 unit embeddings + the scan's fp16 query block        (crs::encoder_forward)
    S  search   exact scan of this rank's shard, over-fetching k' candidates                   (crs::cosine_topk)
                -> fp32 re-rank against the shadow + per-query exactness certificate            (crs::refine_f32_cert)
                -> escalation of unproven queries, a no-op launch when all are proven           (crs::escalate_exact)
                -> best k in this rank's wire block
    M  merge    (N > 1) after ONE all-gather of the wire blocks: k-way merge                    (crs::merge_topk_wire)
Every segment reads and writes fixed buffers of its buffer set (``_Ctx``), so it is captured once and replayed.

Lanes.  The GPU runs a process's streams on four hardware queues; a query-encoder forward is a chain of ~40 dependent launches
of a few microseconds each (latency, not work), a scan is one kernel that wants every byte of HBM bandwidth, and an encoder
kernel with > 48 KB of LDS cannot start on a CU that holds two scan workgroups.  ``lanes='split'`` gives the streams ROLES --
encoder forwards of upcoming batches on encoder lanes, searches on a search lane, tied by events per buffer set -- and asks the
encoder for its <= 48 KB kernel forms (``CRS_ENC_SMALL_LDS`` in the descriptor's flags: per call, not process-wide).
``'batch'`` keeps every batch wholly on its own stream.  ``'auto'`` splits over scans of >= 512 MB per batch, where it measured
faster (bge-class encoders: single rank only, together with encode groups), and keeps one lane per batch
otherwise.

Encode groups.  With role lanes a bge-class encoder forward serves G = 8 consecutive batches (``encode_group``): the per-batch
searches are unchanged, the forward's GEMMs see 8 x the tokens (C3: 134 -> 165 k q/s).  See ``__init__``.
"""
from __future__ import annotations

import os
import sys
from dataclasses import dataclass
from typing import List, Optional

from rag import _native as nat


@dataclass
class ShardView:
    """The rows of ONE device that an engine searches (tensors stay owned by the store / the caller)."""
    slab: object                 # cuda fp16 | int8 [>= n, pdim]
    scales: object               # cuda fp32 [>= n] (int8) or None
    shadow: object               # cuda fp32 [>= n, dim] or None (no fp32 re-rank then)
    n: int
    dim: int
    slab_type: int
    id_base: int = 0             # added to local rows: the shard's first global row
    row_err_max: float = -1.0    # tracked |stored row - fp32 row|_2 maximum (< 0: the analytic worst case)


class _Ctx:
    """Buffers of one in-flight query batch (a batch touches nothing outside its _Ctx + read-only state)."""


class _Group:
    """Buffers of one encoder forward: the token / embedding blocks of G consecutive batches (G = 1: of one batch)."""


class RetrievalEngine:
    @staticmethod
    def plan_layout(hidden: int, scan_bytes: int, *, encode: bool = True, multi: bool = False, lanes: str = "auto", encode_group="auto",
                    n_ctx: int = 0, enc_lanes: int = 0, search_lanes: int = 0, group_cap: int = 0, batch_tokens: int = 0) -> dict:
        """Lanes, encode groups and buffer sets for an encoder of width `hidden` beside scans of `scan_bytes` per batch (measured
        rules; every explicit argument wins over its rule):
          * role lanes ('split') for MiniLM-class encoders always, for bge-class ones over scans of >= 512 MB on one rank; else every
            batch on its own stream;
          * bge-class encoders (hidden > 384; single rank): 8 batches per encoder forward over 24 buffer sets (16 over 32 when a
            batch is < 4096 tokens), 2 encoder + 1 search lane;
          * MiniLM-class over SHORT scans (< 2 GB: one rank's share of a 4- or 8-GPU C4 step, C2's 100 k rows): the 38-launch forward
            is what the lanes wait for -- 32 batches per forward on ONE encoder lane, 64 buffer sets, and TWO search lanes so that a
            batch's tail of small kernels and the next batch's sweep overlap (8-GPU rank proxy: 0.255 -> 0.185 - 0.19 ms per batch;
            C2: 398 -> 833 - 846 k q/s -- there the chain WAS the batch);
          * MiniLM-class over long scans (C4 on one GPU): 16 batches per forward on one encoder lane, 32 buffer sets, ONE search lane
            -- 38 kernel boundaries per 16 batches instead of per batch: 47.3 -> 48.6 - 49.0 k q/s on one box (8 per forward over 16
            sets: 48.6 k; a second search lane: 49.8 k, not taken -- consecutive scans then overlap and a trace's per-kernel
            durations stop meaning "one scan").
        group_cap > 0 limits the group (a caller that knows its calls bring fewer batches than a group holds)."""
        big, short = hidden > 384, scan_bytes < (2 << 30)
        pipelined = lanes == "split" or (lanes == "auto" and encode and ((not big) or (scan_bytes >= (512 << 20) and not multi)))
        if encode_group == "auto":
            big_g = 16 if 0 < batch_tokens < 4096 else 8      # (C5's 1024-token batches: 38.9 -> 39.7 k q/s with 16; C3's 4096: 8 is best)
            encode_group = (big_g if big else 32 if short else 16) if (encode and pipelined) else 1
        encode_group = max(1, int(os.environ.get("CRS_ENCODE_GROUP", encode_group))) if encode else 1
        if group_cap > 0:
            encode_group = min(encode_group, group_cap)
        if n_ctx <= 0:
            n_ctx = 8 if encode_group == 1 else (3 * encode_group if (big and encode_group <= 8) else 2 * encode_group)
        while n_ctx % encode_group:
            encode_group -= 1
        if pipelined:
            minilm_groups = encode_group > 1 and not big
            n_enc = enc_lanes if enc_lanes > 0 else (1 if minilm_groups else 2)
            n_srch = search_lanes if search_lanes > 0 else (2 if (minilm_groups and short) else 1)
        else:
            n_enc = n_srch = n_ctx
        return {"pipelined": pipelined, "encode_group": encode_group, "n_ctx": n_ctx, "n_enc": n_enc, "n_srch": n_srch}

    def __init__(self, encoder, view: ShardView, queries_per_batch: int, seq: int, top_k: int, *, k_scan: int = 24, k_scan_exact: int = 0,
                 refine: bool = True, exact="auto", exact_cap: int = nat.EXACT_CAP, n_ctx: int = 0, lanes: str = "auto",
                 enc_lanes: int = 0, search_lanes: int = 0, graphs: bool = True, dist=None, world: int = 1, rank: int = 0,
                 queries_per_rank: bool = False, encode_shard: int = 1, proxy_encode_shard: int = 1, encode: bool = True,
                 enc_small_lds="auto", enc_cus: int = 0, encode_group="auto", group_cap: int = 0, search_fuse="auto"):
        """queries_per_batch: the GLOBAL batch every rank searches (strong scaling), or with ``queries_per_rank`` the queries
        THIS rank contributes (weak scaling: the scan then sees world x that many).  encode_shard = W > 1: each rank encodes
        Qb / W queries and the embeddings are all-gathered first (two collectives per batch instead of one).
        proxy_encode_shard (diagnostic, one rank): encode Qb / W queries and tile them in place of that all-gather.
        encode=False (diagnostic): the caller fills ``ctx.q_out`` / uses set_queries(); no encoder in the batch."""
        import torch
        nat.require_gpu()
        self.torch = torch
        self.enc, self.view = encoder, view
        self.dev = view.slab.device
        self.dist, self.world, self.rank = dist, world, rank
        self.multi = dist is not None
        self.k, self.seq = int(top_k), int(seq)
        self.refine = bool(refine) and view.shadow is not None
        self.exact = (view.slab_type == nat.SLAB_F16) if exact == "auto" else bool(exact)
        self.exact = self.exact and self.refine
        self.exact_cap = int(exact_cap)
        self.encode = bool(encode)
        qb = int(queries_per_batch)
        self.nq_all = qb * world if queries_per_rank else qb
        # candidates the scan fetches: nat.overfetch's rule from the wanted length, or (k_scan_exact > 0: A/B runs) exactly that many
        self.k_scan = (min(nat.MAX_K, max(self.k, int(k_scan_exact))) if k_scan_exact > 0 else
                       nat.overfetch(self.nq_all, self.k, int(k_scan), view.n, view.slab_type)) if self.refine else self.k
        shard_w = 1
        if self.multi and not queries_per_rank and encode_shard > 1 and qb % encode_shard == 0:
            shard_w = encode_shard
        if not self.multi and proxy_encode_shard > 1 and qb % proxy_encode_shard == 0:
            shard_w = proxy_encode_shard
        self.shard_w = shard_w
        self.q_loc = qb // shard_w                          # queries THIS rank encodes per batch
        self.enc_lo = (rank if self.multi else 0) * self.q_loc if shard_w > 1 else 0
        self.gather_q = (self.multi and (queries_per_rank or shard_w > 1)) or (not self.multi and shard_w > 1)
        self.pd = nat.padded_dim(view.dim, view.slab_type)
        scan_bytes = view.n * self.pd * (1 if view.slab_type == nat.SLAB_I8 else 2)
        hidden = encoder.shape.hidden if encoder is not None else view.dim
        plan = self.plan_layout(hidden, scan_bytes, encode=self.encode, multi=self.multi, lanes=lanes, encode_group=encode_group,
                                n_ctx=int(n_ctx), enc_lanes=enc_lanes, search_lanes=search_lanes, group_cap=int(group_cap),
                                batch_tokens=self.q_loc * self.seq)
        self.pipelined, self.enc_group, self.n_ctx = plan["pipelined"], plan["encode_group"], plan["n_ctx"]
        self.n_enc, self.n_srch = plan["n_enc"], plan["n_srch"]
        # <= 48 KB kernel forms of the encoder (they can start beside a scan's resident workgroups): with role lanes always;
        # 'auto' otherwise keeps the default forms
        self.small_lds = self.pipelined if enc_small_lds == "auto" else bool(enc_small_lds)
        self.use_graph = bool(graphs)
        if self.pipelined:
            self.enc_cus = int(os.environ.get("CRS_ENC_CUS", enc_cus))
            if self.enc_cus > 0:
                # A/B: encoder lanes confined to a few CUs each (lane i: CUs [i * enc_cus, (i + 1) * enc_cus)); measured no gain
                # (DESIGN.md section 4): what the encoder costs the sweep is its kernel boundaries, not the CUs it sits on
                self.enc_streams = [nat.cu_masked_stream(i * self.enc_cus, self.enc_cus, self.dev) for i in range(self.n_enc)]
            else:
                self.enc_streams = [torch.cuda.Stream(device=self.dev) for _ in range(self.n_enc)]
            # (search lanes, or the encoder lanes, as high-priority streams: measured null on C3 / C4 / C5 / the 8-GPU rank proxy)
            # (the chip partitioned by CU masks -- searches on 224 / 192 / 160 CUs, encoders on the rest -- loses on C5 and C3, where
            # the bge forwards need far more than the rest, and is neutral on C4: DESIGN.md section 4)
            self.srch_streams = [torch.cuda.Stream(device=self.dev) for _ in range(self.n_srch)]
        else:
            self.enc_streams = self.srch_streams = [torch.cuda.Stream(device=self.dev) for _ in range(self.n_ctx)]
        # ENCODE GROUPS: one encoder forward serves G consecutive batches (their token blocks are slices of one [G * q_loc, seq]
        # block; the searches stay per batch).  A bge-class forward over one 256-query batch (4096 tokens) leaves its GEMMs with
        # 48 - 144 tiles for 256 CUs: 1.70 ms per 256 queries alone, 1.41 at two batches per forward, 1.18 at four
        # (tools/bench_encoder.py); beside the scans, C3 134 -> 165 k q/s and C5 34.4 -> 39.3 k with G = 8 over 24 buffer sets on role
        # lanes (tools/r3_group.sh).  A MiniLM-class forward is a launch-latency chain whatever its size: grouping takes it off
        # the critical path of short scans (plan_layout), and is worth + 1 - 2 % on C4, - 10 % on C2.  The price is latency: a
        # batch's search waits for its group's forward.  N > 1: the group's embeddings travel in ONE all-gather.
        # FUSED SEARCH GRAPHS (single rank): the search segments of F consecutive members of a group replayed as ONE hipGraph --
        # a graph boundary on the search lane costs ~20 us (tools/timeline.py), 1.5 % of a C4 batch, once per F batches instead of
        # once per batch.  The members' results then complete together.  N > 1 keeps one graph per batch: a collective follows each.
        if search_fuse == "auto":
            search_fuse = 4 if (not self.multi and self.enc_group % 4 == 0 and self.use_graph) else 1
        self.search_fuse = max(1, int(os.environ.get("CRS_SEARCH_FUSE", search_fuse)))
        if self.multi or self.enc_group % self.search_fuse:
            self.search_fuse = 1
        self.groups: List[_Group] = []
        self.ctxs: List[_Ctx] = []
        for g0 in range(0, self.n_ctx, self.enc_group):
            grp = self._make_group(list(range(g0, g0 + self.enc_group)))
            self.groups.append(grp)
            self.ctxs.extend(self._make_ctx(grp, j) for j in range(self.enc_group))
        # segments of a batch, in order; the lane each runs on; the collective that follows it (N > 1)
        # gathered queries (sharded / per-rank encode, or its one-rank proxy): a second per-GROUP segment on the encoder lane,
        # behind the all-gather, lays the group's queries out batch by batch and writes their fp16 blocks -- two kernels per group
        # instead of two per batch in front of every scan
        self.segs = ([self._seg_encode] + ([self._seg_prep] if self.gather_q else []) + [self._seg_search] +
                     ([self._seg_merge] if self.multi else []))
        self.group_segs = 2 if self.gather_q else 1          # leading segments that run once per group (first buffer set)
        self.seg_lanes = ["E"] * self.group_segs + ["S", "S"][: len(self.segs) - self.group_segs]
        self.last_search_seg = self.group_segs
        self.exchanges = [None] * len(self.segs)
        if self.multi and self.gather_q:   # queries encoded in shards (or per-rank queries): embeddings gathered first
            self.exchanges[0] = lambda c: dist.all_gather_into_tensor(c.grp.q_gath, c.grp.q_out)
        if self.multi:                     # THE exchange of a sharded search: every rank's wire block
            self.exchanges[self.last_search_seg] = lambda c: dist.all_gather_into_tensor(c.wire.gathered, c.wire.buf)
        # (the embeddings of a group travel in one all-gather: 1 / G of a collective per batch)
        self.collectives_per_batch = round(sum((1.0 / self.enc_group if j == 0 else 1.0) for j, e in enumerate(self.exchanges) if e is not None), 4)
        if self.collectives_per_batch == int(self.collectives_per_batch):
            self.collectives_per_batch = int(self.collectives_per_batch)
        self._issued = 0
        self._warm = False
        self._seg_events = None     # measure_search_segment_ms: (start, end) events of the search segments

    # ---- buffers ---------------------------------------------------------------------------------------------------
    def _make_group(self, members) -> _Group:
        torch, v, dev = self.torch, self.view, self.dev
        g = _Group()
        n = len(members) * self.q_loc
        g.members = members
        g.ids = torch.zeros((n, self.seq), dtype=torch.int32, device=dev)
        g.lens = torch.ones(n, dtype=torch.int32, device=dev)
        g.q_out = torch.empty((n, v.dim), dtype=torch.float32, device=dev)
        g.q16 = torch.empty((n, self.pd), dtype=torch.float16, device=dev)
        g.enc_ws = (torch.empty(self.enc.workspace_bytes(n, self.seq), dtype=torch.uint8, device=dev) if self.encode else None)
        # N > 1 with gathered queries: every rank's block of the group in ONE all-gather, [world, G * q_loc, dim]
        g.q_gath = torch.empty((self.world * n, v.dim), dtype=torch.float32, device=dev) if (self.multi and self.gather_q) else None
        if self.gather_q:      # the group's queries batch by batch: [G, nq_all, dim] fp32 and the scans' fp16 blocks
            G = len(members)
            g.q_all32 = (g.q_gath if (g.q_gath is not None and G == 1) else       # (one batch per forward: the all-gather's output IS it)
                         torch.empty((G * self.nq_all, v.dim), dtype=torch.float32, device=dev))
            g.q_all16 = torch.empty((G * self.nq_all, self.pd), dtype=torch.float16, device=dev)
        g.ev_enc = torch.cuda.Event()
        g.n_enc = 0
        return g

    def _make_ctx(self, grp: _Group, slot: int) -> _Ctx:
        torch, v, dev = self.torch, self.view, self.dev
        c = _Ctx()
        c.grp, c.slot, c.n_sub = grp, slot, 0
        lo, hi = slot * self.q_loc, (slot + 1) * self.q_loc
        c.ids, c.lens, c.q_out, c.q16 = grp.ids[lo:hi], grp.lens[lo:hi], grp.q_out[lo:hi], grp.q16[lo:hi]   # views of the group's blocks
        c.ws = torch.empty(nat.scan_workspace_bytes(self.nq_all, v.dim, self.k_scan, v.n), dtype=torch.uint8, device=dev)
        c.cand_s = torch.empty((self.nq_all, self.k_scan), dtype=torch.float32, device=dev)
        c.cand_i = torch.empty((self.nq_all, self.k_scan), dtype=torch.int64, device=dev)
        c.wire = nat.WireBlock(self.nq_all, self.k, dev, self.world, gather=self.multi)   # this rank's (ids | scores) block
        c.status = torch.zeros(self.nq_all, dtype=torch.int32, device=dev)
        c.exact_ws = (torch.empty(nat.exact_workspace_bytes(self.nq_all, self.exact_cap), dtype=torch.uint8, device=dev)
                      if self.refine else None)
        c.graphs = None
        c.chunk_graph = None
        if self.multi:
            c.fin_s = torch.empty((self.nq_all, self.k), dtype=torch.float32, device=dev)
            c.fin_i = torch.empty((self.nq_all, self.k), dtype=torch.int64, device=dev)
        if self.gather_q:
            c.q_all32 = grp.q_all32[slot * self.nq_all:(slot + 1) * self.nq_all]
            c.q_all16 = grp.q_all16[slot * self.nq_all:(slot + 1) * self.nq_all]
        c.ev_done = torch.cuda.Event()
        c.ev_seg = [torch.cuda.Event() for _ in range(4)]     # a segment's output is complete (next segment on another lane)
        return c

    def set_tokens(self, ctx_index: int, ids, lens, stream=None) -> None:
        """Token ids / lengths of the NEXT batch of buffer set ctx_index (this rank's slice when the encode is sharded):
        int32 [q_loc, seq] / [q_loc], host or device.  Queued behind the previous batch of that buffer set."""
        torch, c = self.torch, self.ctxs[ctx_index]
        st = stream or self.enc_streams[ctx_index % self.n_enc]
        with torch.cuda.stream(st):
            st.wait_event(c.ev_done)
            c.ids.copy_(torch.as_tensor(ids, dtype=torch.int32), non_blocking=True)
            c.lens.copy_(torch.as_tensor(lens, dtype=torch.int32), non_blocking=True)
            c.ev_done.record(st)

    # ---- segments --------------------------------------------------------------------------------------------------
    def _seg_encode(self, c: _Ctx) -> None:      # token ids -> fp32 embeddings + the scan's fp16 query block (the whole group's)
        g = c.grp
        if self.encode:
            self.enc.forward(g.ids, g.lens, out=g.q_out, workspace=g.enc_ws, q16_out=g.q16, slab_type=self.view.slab_type,
                             small_lds=self.small_lds)
        else:
            nat.queries_to_f16(g.q_out, self.view.slab_type, out=g.q16)

    def _seg_prep(self, c: _Ctx) -> None:     # gathered queries of the group -> per-batch fp32 blocks + fp16 scan blocks
        g, v = c.grp, self.view
        G = len(g.members)
        if not self.multi:         # proxy: the local queries tiled in place of the all-gather
            g.q_all32.view(G, self.shard_w, self.q_loc, v.dim).copy_(g.q_out.view(G, 1, self.q_loc, v.dim).expand(G, self.shard_w, self.q_loc, v.dim))
        elif G > 1:                # batch j = rows j of every rank's block: global query r * q_loc + i, as ungrouped
            g.q_all32.view(G, self.world, self.q_loc, v.dim).copy_(g.q_gath.view(self.world, G, self.q_loc, v.dim).transpose(0, 1))
        nat.queries_to_f16(g.q_all32, v.slab_type, out=g.q_all16)

    def _seg_search(self, c: _Ctx) -> None:   # all queries x this rank's shard -> wire block
        self._seg_scan(c)
        if self.refine:
            self._seg_post(c)

    def _seg_scan(self, c: _Ctx) -> None:     # the sweep: k' candidates per query (no re-rank configured: the final lists)
        v = self.view
        qa16 = c.q_all16 if self.gather_q else c.q16
        if self.refine:
            nat.cosine_topk(qa16, v.slab, v.n, v.dim, self.k_scan, slab_type=v.slab_type, scales=v.scales, id_base=v.id_base,
                            workspace=c.ws, out_scores=c.cand_s, out_ids=c.cand_i)
        else:
            nat.cosine_topk(qa16, v.slab, v.n, v.dim, self.k, slab_type=v.slab_type, scales=v.scales, id_base=v.id_base,
                            workspace=c.ws, out_scores=c.wire.scores, out_ids=c.wire.ids)

    def _seg_post(self, c: _Ctx) -> None:     # fp32 re-rank + certificate + escalation -> this rank's wire block
        v = self.view
        qa32 = c.q_all32 if self.gather_q else c.q_out
        qa16 = c.q_all16 if self.gather_q else c.q16
        if self.refine:
            nat.refine_f32_cert(qa32, qa16, v.shadow, v.n, v.id_base, c.cand_i, c.cand_s, self.k, v.row_err_max, v.slab_type,
                                c.exact_ws, self.exact_cap, out_scores=c.wire.scores, out_ids=c.wire.ids, status=c.status)
            if self.exact:
                nat.escalate_exact(qa32, qa16, v.slab, v.shadow, v.n, v.id_base, self.k, c.wire.scores, c.wire.ids, c.status,
                                   c.exact_ws, self.exact_cap, scales=v.scales)

    def _seg_merge(self, c: _Ctx) -> None:       # N > 1: the gathered wire blocks -> global top-k
        nat.merge_topk_wire(c.wire.gathered, self.world, self.nq_all, self.k, self.k, out_scores=c.fin_s, out_ids=c.fin_i)

    # ---- issue -----------------------------------------------------------------------------------------------------
    def submit(self, ctx_index: int):
        """Issue one batch from buffer set ctx_index: encode on an encoder lane, search (+ exchange + merge) on a search lane.
        Returns the (scores, ids) DEVICE tensors the batch will fill (valid after ``wait(ctx_index)``)."""
        torch, c = self.torch, self.ctxs[ctx_index]
        g = c.grp
        b = self._issued
        self._issued += 1
        if c.slot == 0:
            g.n_enc += 1
        elif c.n_sub >= g.n_enc:
            raise nat.NativeError("encode groups: submit the group's first buffer set before the others (step() / search_token_batches do)")
        c.n_sub += 1
        lane = {"E": self.enc_streams[((b - c.slot) // self.enc_group) % self.n_enc], "S": self.srch_streams[b % self.n_srch]}
        prev, prev_ev = None, None
        n_seg = len(self.segs)
        for j, seg in enumerate(self.segs):
            if j < self.group_segs and c.slot != 0:     # the group's forward (+ query layout) was issued with its first buffer set
                prev_ev = g.ev_enc
                continue
            st = lane[self.seg_lanes[j]]
            with torch.cuda.stream(st):
                if j == 0:
                    for m in g.members:    # the previous users of these buffer sets are through (no-op before their first use)
                        st.wait_event(self.ctxs[m].ev_done)
                elif st is not prev:
                    st.wait_event(prev_ev)            # lane change: the previous segment's output (and its collective) is complete
                timed = self._seg_events is not None
                if timed and j == self.last_search_seg:
                    c.t0 = torch.cuda.Event(enable_timing=True)
                    c.t0.record(st)
                if c.graphs is not None:
                    c.graphs[j].replay()
                else:
                    seg(c)
                if timed and j == self.last_search_seg:
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record(st)
                    self._seg_events.append((c.t0, e1))
                if self.exchanges[j] is not None:
                    self.exchanges[j](c)
                if j == self.group_segs - 1:
                    g.ev_enc.record(st)
                    prev_ev = g.ev_enc
                elif j + 1 < n_seg and lane[self.seg_lanes[j + 1]] is not st:
                    c.ev_seg[j].record(st)
                    prev_ev = c.ev_seg[j]
                if j == n_seg - 1:
                    c.ev_done.record(st)
            prev = st
        return (c.fin_s, c.fin_i) if self.multi else (c.wire.scores, c.wire.ids)

    def wait(self, ctx_index: int) -> None:
        self.ctxs[ctx_index].ev_done.synchronize()

    def outputs(self, ctx_index: int):
        c = self.ctxs[ctx_index]
        return ((c.fin_s, c.fin_i) if self.multi else (c.wire.scores, c.wire.ids)) + (c.status,)

    def warm_up(self) -> None:
        """Two eager batches per buffer set, then capture every segment into a hipGraph (run eagerly if a capture fails)."""
        if self._warm:
            return
        torch = self.torch
        torch.cuda.synchronize()
        for g in self.groups:
            for _ in range(2):
                for i in g.members:
                    self.submit(i)
            torch.cuda.synchronize()
            if self.use_graph:
                # thread_local: with N > 1 the process group's watchdog thread polls events while we capture
                try:
                    for i in g.members:
                        c, gl = self.ctxs[i], []
                        for j, seg in enumerate(self.segs):
                            if j < self.group_segs and c.slot != 0:      # one encode (+ layout) graph per group, held by its first buffer set
                                gl.append(None)
                                continue
                            st = (self.enc_streams if self.seg_lanes[j] == "E" else self.srch_streams)[0]
                            g_ = torch.cuda.CUDAGraph()
                            with torch.cuda.graph(g_, stream=st, capture_error_mode="thread_local"):
                                seg(c)
                            gl.append(g_)
                        c.graphs = gl
                    if self.search_fuse > 1:
                        j = self.last_search_seg
                        for i0 in range(g.members[0], g.members[-1] + 1, self.search_fuse):
                            g_ = torch.cuda.CUDAGraph()
                            with torch.cuda.graph(g_, stream=self.srch_streams[0], capture_error_mode="thread_local"):
                                for i in range(i0, i0 + self.search_fuse):
                                    self.segs[j](self.ctxs[i])
                            self.ctxs[i0].chunk_graph = g_
                except Exception as exc:   # noqa: BLE001 -- report and keep going without graphs
                    print(f"[engine] hipGraph capture failed on rank {self.rank} ({exc!r}); launching eagerly", file=sys.stderr, flush=True)
                    self.use_graph = False
                    for cc in self.ctxs:
                        cc.graphs = None
                    torch.cuda.synchronize()   # (no break: every rank must still run the same warm-up collectives)
        self._warm = True

    def submit_chunk(self, i0: int) -> None:
        """Buffer sets i0 .. i0 + search_fuse - 1 (i0 a multiple of search_fuse, their tokens in place) as one unit: the group's
        forward if i0 opens its group, then ONE graph with the chunk's searches."""
        torch, F = self.torch, self.search_fuse
        c0 = self.ctxs[i0]
        if F == 1 or c0.graphs is None or getattr(c0, "chunk_graph", None) is None:
            for i in range(i0, i0 + F):
                self.submit(i)
            return
        g = c0.grp
        b = self._issued
        self._issued += F
        if c0.slot == 0:
            g.n_enc += 1
            st = self.enc_streams[(b // self.enc_group) % self.n_enc]
            with torch.cuda.stream(st):
                for m in g.members:
                    st.wait_event(self.ctxs[m].ev_done)
                for j in range(self.group_segs):
                    c0.graphs[j].replay()
                    if self.exchanges[j] is not None:
                        self.exchanges[j](c0)
                g.ev_enc.record(st)
        elif c0.n_sub >= g.n_enc:
            raise nat.NativeError("encode groups: submit the group's first buffer set before the others (step() / search_token_batches do)")
        st = self.srch_streams[(b // F) % self.n_srch]
        with torch.cuda.stream(st):
            st.wait_event(g.ev_enc)
            c0.chunk_graph.replay()
            for i in range(i0, i0 + F):
                self.ctxs[i].n_sub += 1
                self.ctxs[i].ev_done.record(st)

    def step(self, fused: bool = True) -> None:
        """One batch from every buffer set (the unit bench.py times)."""
        if fused and self.search_fuse > 1:
            for i0 in range(0, self.n_ctx, self.search_fuse):
                self.submit_chunk(i0)
            return
        for i in range(self.n_ctx):
            self.submit(i)

    def measure_search_segment_ms(self, rounds: int = 2):
        """Mean duration of the SEARCH segment (scan + merge + tile refine + certificate + escalation launches) while the
        engine runs its normal mix, hipEvent-timed on the search lane over `rounds` steps (the first is discarded).
        Single rank only; None without graphs."""
        torch = self.torch
        if self.multi or not self._warm or self.ctxs[0].graphs is None:
            return None
        torch.cuda.synchronize()
        evs = []
        for r in range(max(2, rounds)):
            self._seg_events = [] if r else None
            self.step(fused=False)
            evs += self._seg_events or []
        self._seg_events = None
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in evs) / len(evs)

    def describe_lanes(self) -> str:
        if not self.pipelined:
            return "one per batch"
        return f"{self.n_enc} encoder + {self.n_srch} search (encoder kernels <= 48 KB of LDS)"

    # ---- the product entry: many query batches through the pipeline -------------------------------------------------
    def search_token_batches(self, batches):
        """batches: iterable of (ids int32 [m, seq], lens int32 [m]) host arrays with m <= queries per batch (a short last
        batch is padded with copies of its first query).  Yields, in order, (scores [m, k], rows [m, k], status [m]) numpy
        arrays.  Up to n_ctx batches are in flight; results are read back as their buffer set comes round again."""
        import numpy as np
        if self.multi or self.gather_q:
            raise nat.NativeError("search_token_batches drives a single-rank engine")
        self.warm_up()
        pending = {}          # ctx index -> rows of that batch that are real
        order = []

        def collect(i):
            self.wait(i)
            s, r, st = self.outputs(i)
            m = pending.pop(i)
            return s[:m].cpu().numpy(), r[:m].cpu().numpy(), st[:m].cpu().numpy()

        G = self.enc_group
        nb = 0
        for ids, lens in batches:
            i = nb % self.n_ctx
            if i in pending:
                order.remove(i)
                yield collect(i)
            ids = np.asarray(ids, dtype=np.int32)
            lens = np.asarray(lens, dtype=np.int32)
            m = ids.shape[0]
            if m < self.q_loc:
                ids = np.concatenate([ids, np.repeat(ids[:1], self.q_loc - m, axis=0)])
                lens = np.concatenate([lens, np.repeat(lens[:1], self.q_loc - m)])
            self.set_tokens(i, ids, lens)
            pending[i] = m
            order.append(i)
            nb += 1
            if i % G == G - 1:            # the group's token block is complete: one forward, G searches
                for j in range(i - G + 1, i + 1, self.search_fuse):
                    self.submit_chunk(j)
        tail = nb % G
        if tail:                          # an incomplete last group: its forward runs over the whole token block (the free slices
            i0 = (nb - tail) % self.n_ctx  # hold earlier, valid tokens), only the real batches are searched
            for j in range(i0, i0 + tail):
                self.submit(j)
        for i in list(order):
            yield collect(i)
