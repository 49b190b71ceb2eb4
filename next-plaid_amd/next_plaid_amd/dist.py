"""Document-sharded search: one process per GPU, RCCL all-gather of per-shard rank keys / top-k.

The reference has no multi-GPU path; this is the north_star's "index shards by
document across the GPUs of one node with a final RCCL all-gather of per-shard top-k over xGMI",
done so that the merged result is IDENTICAL to the unsharded search (search.rs:460-515):

  phase A  every shard runs S1-S5 locally and emits its best n_sel candidates as 64-bit rank keys
           (approx score, then ascending global doc id -- the reference's stable-sort order)
  gather 1 all-gather keys  [B, n_sel] x u64 per rank   (8 MB at B=64, n_sel=1024, G=8: latency-bound)
  cut      every rank takes the global n_sel-th key per query (np_hip_select_cut)
  phase B  exact MaxSim only for local candidates with key >= cut  -> local top-k triples
  gather 2 all-gather (id, score, key, count) packed in one i64 buffer [B, 3k+1]
  merge    top-k by (exact score desc, approx rank) == the reference's final stable sort

Two small collectives per batch; no payload-sized traffic ever crosses xGMI.  With torch.distributed
backend "nccl" this IS RCCL on ROCm; the same code runs over "gloo" on CPU tensors in the tests with
an oracle-backed shard backend (tests/test_dist_cpu.py).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import api


class HipShardBackend:
    """One shard's device work through the C ABI, on torch tensors resident on the shard's GPU."""

    def __init__(self, index: "api.MmapIndex", device=None, stream=None):
        import torch
        self.torch = torch
        self.index = index
        self.device = torch.device("cuda", index.info.device) if device is None else device
        # a NON-default stream: the C ABI treats a NULL stream as "use the context's own stream",
        # which would not be ordered with torch work on the legacy default stream
        self.stream = torch.cuda.Stream(self.device) if stream is None else stream

    def stream_ctx(self):
        return self.torch.cuda.stream(self.stream)

    def n_sel(self, params):
        return int(api.lib().np_hip_n_sel(C.byref(params._c())))

    def num_partitions(self):
        return self.index.num_partitions()

    def _stream(self):
        return C.c_void_p(self.stream.cuda_stream)

    def eligible(self, d_subset):
        """This shard's eligible-centroid bitmap for a subset (search.rs:350-364), u32 words as an int32 tensor."""
        t = self.torch
        words = int(api.lib().np_hip_elig_words(self.index._h))
        bits = t.zeros(max(words, 1), dtype=t.int32, device=self.device)
        api._check(api.lib().np_hip_subset_eligible(self.index._h, C.c_void_p(d_subset.data_ptr()), d_subset.numel(),
                                                    C.c_void_p(bits.data_ptr()), self._stream()))
        return bits

    def phase_a(self, d_q, d_qoff, h_qoff, params, d_subset=None, elig=None):
        t = self.torch
        B = len(h_qoff) - 1
        ns = max(self.n_sel(params), 1)
        keys = t.zeros((B, ns), dtype=t.int64, device=self.device)
        st = C.c_void_p()
        p = params._c()
        hq = np.ascontiguousarray(h_qoff, np.int32)
        api._check(api.lib().np_hip_search_phase_a(
            self.index._h, C.c_void_p(d_q.data_ptr()), C.c_void_p(d_qoff.data_ptr()), hq.ctypes.data_as(C.c_void_p),
            B, self.index.embedding_dim(), C.byref(p),
            None if d_subset is None else C.c_void_p(d_subset.data_ptr()), -1 if d_subset is None else d_subset.numel(),
            None if elig is None else C.c_void_p(elig.data_ptr()), C.c_void_p(keys.data_ptr()), self._stream(),
            C.byref(st)))
        return keys[:, : self.n_sel(params)], (st, B, params)

    def select_cut(self, all_keys):
        t = self.torch
        G, B, ns = all_keys.shape
        cut = t.zeros(B, dtype=t.int64, device=self.device)
        all_keys = all_keys.contiguous()
        api._check(api.lib().np_hip_select_cut(self.index._h, C.c_void_p(all_keys.data_ptr()), G, B, ns,
                                               C.c_void_p(cut.data_ptr()), self._stream()))
        return cut

    def phase_b(self, state, cut):
        t = self.torch
        st, B, params = state
        k = max(params.top_k, 1)
        packed = t.zeros((B, 3 * k + 1), dtype=t.int64, device=self.device)   # ids | keys | scores(bits) | count
        ids = t.zeros((B, k), dtype=t.int64, device=self.device)
        sc = t.zeros((B, k), dtype=t.float32, device=self.device)
        keys = t.zeros((B, k), dtype=t.int64, device=self.device)
        cnt = t.zeros(B, dtype=t.int32, device=self.device)
        api._check(api.lib().np_hip_search_phase_b(
            self.index._h, st, None if cut is None else C.c_void_p(cut.data_ptr()), C.c_void_p(ids.data_ptr()),
            C.c_void_p(sc.data_ptr()), C.c_void_p(keys.data_ptr()), C.c_void_p(cnt.data_ptr()), self._stream()))
        packed[:, :k] = ids
        packed[:, k:2 * k] = keys
        packed[:, 2 * k:3 * k] = sc.view(t.int32).to(t.int64)
        packed[:, 3 * k] = cnt.to(t.int64)
        return packed

    def end(self, state):
        api.lib().np_hip_search_end(self.index._h, state[0])

    def merge(self, all_packed, top_k):
        t = self.torch
        G, B, _ = all_packed.shape
        k = max(top_k, 1)
        ids = all_packed[:, :, :k].contiguous()
        keys = all_packed[:, :, k:2 * k].contiguous()
        sc = all_packed[:, :, 2 * k:3 * k].to(t.int32).view(t.float32).contiguous()
        cnt = all_packed[:, :, 3 * k].to(t.int32).contiguous()
        o_ids = t.zeros((B, k), dtype=t.int64, device=self.device)
        o_sc = t.zeros((B, k), dtype=t.float32, device=self.device)
        o_cnt = t.zeros(B, dtype=t.int32, device=self.device)
        api._check(api.lib().np_hip_merge_topk(
            self.index._h, C.c_void_p(ids.data_ptr()), C.c_void_p(sc.data_ptr()), C.c_void_p(keys.data_ptr()),
            C.c_void_p(cnt.data_ptr()), G, B, top_k, C.c_void_p(o_ids.data_ptr()), C.c_void_p(o_sc.data_ptr()),
            C.c_void_p(o_cnt.data_ptr()), self._stream()))
        return o_ids, o_sc, o_cnt


class ShardedSearcher:
    """Runs the two-collective protocol over `backends` (the shards this process holds, normally
    one) and, if `group`/torch.distributed is initialised, over all ranks.  Rank r holds the
    shards r*len(backends) ... of world_size*len(backends) in rank order."""

    def __init__(self, backends, use_dist: bool | None = None, group=None):
        import torch
        self.torch = torch
        self.backends = list(backends)
        self.group = group
        if use_dist is None:
            use_dist = torch.distributed.is_available() and torch.distributed.is_initialized()
        self.use_dist = use_dist

    def _all_gather(self, local):  # local: [L, ...] stacked over this process's shards -> [G, ...]
        if not self.use_dist:
            return local
        import torch.distributed as dist
        ws = dist.get_world_size(self.group)
        out = [self.torch.empty_like(local) for _ in range(ws)]
        dist.all_gather(out, local.contiguous(), group=self.group)
        return self.torch.cat(out, 0)

    def search_batch_device(self, d_q, d_qoff, h_qoff, params, d_subset=None):
        """Queries already resident on every shard's device (the host broadcasts them).
        d_q / d_qoff (/ d_subset, global i64 doc ids): one tensor per local backend (or a single tensor when there is one)."""
        with self.backends[0].stream_ctx():
            return self._search(d_q, d_qoff, h_qoff, params, d_subset)

    def _global_eligible(self, subs, params):
        """OR of the shards' eligible-centroid bitmaps: one more small all-gather, only with a subset on the dense
        path (the batched path just filters candidates, search.rs:542-545).  Identical on every shard."""
        K = getattr(self.backends[0], "num_partitions", None)
        cbs = params.centroid_batch_size
        if subs is None or (K is not None and cbs > 0 and K() > cbs):
            return None
        if any(s.numel() == 0 for s in subs):
            return None
        t = self.torch
        local = t.stack([be.eligible(s).to(self.backends[0].device) for be, s in zip(self.backends, subs)], 0)
        allb = self._all_gather(local)
        glob = allb[0].clone()
        for g in range(1, allb.shape[0]):
            glob |= allb[g]
        return glob

    def _search(self, d_q, d_qoff, h_qoff, params, d_subset=None):
        t = self.torch
        single = not isinstance(d_q, (list, tuple))
        dqs = [d_q] if single else list(d_q)
        dos = [d_qoff] if single else list(d_qoff)
        subs = None if d_subset is None else ([d_subset] * len(self.backends) if not isinstance(d_subset, (list, tuple))
                                              else list(d_subset))
        glob = self._global_eligible(subs, params)
        keys, states = [], []
        packed = []
        try:
            for i, (be, q, o) in enumerate(zip(self.backends, dqs, dos)):   # inside the try: a failing shard must not strand the others' contexts
                k, st = be.phase_a(q, o, h_qoff, params, None if subs is None else subs[i],
                                   None if glob is None else glob.to(be.device))
                keys.append(k)
                states.append(st)
            all_keys = self._all_gather(t.stack([k.to(keys[0].device) for k in keys], 0))
            for be, st in zip(self.backends, states):
                cut = be.select_cut(all_keys.to(be.device))
                packed.append(be.phase_b(st, cut))
        finally:
            for be, st in zip(self.backends, states):
                be.end(st)
        all_packed = self._all_gather(t.stack([p.to(packed[0].device) for p in packed], 0))
        return self.backends[0].merge(all_packed.to(self.backends[0].device), params.top_k)

    def search_batch(self, queries, params, subset=None):
        """Host-side convenience: numpy queries in, list of QueryResult out (rank-identical)."""
        t = self.torch
        qs = [np.ascontiguousarray(q, np.float32) for q in queries]
        off = np.zeros(len(qs) + 1, np.int32)
        off[1:] = np.cumsum([q.shape[0] for q in qs])
        flat = np.concatenate(qs, 0)
        with self.backends[0].stream_ctx():
            dq = [t.from_numpy(flat).to(be.device) for be in self.backends]
            do = [t.from_numpy(off).to(be.device) for be in self.backends]
            ds = None if subset is None else [t.from_numpy(np.ascontiguousarray(subset, np.int64)).to(be.device)
                                              for be in self.backends]
            ids, sc, cnt = self._search(dq, do, off, params, ds)
            ids, sc, cnt = ids.cpu().numpy(), sc.cpu().numpy(), cnt.cpu().numpy()
        return [api.QueryResult(i, ids[i, : cnt[i]].copy(), sc[i, : cnt[i]].copy()) for i in range(len(qs))]


class ShardComm:
    """np_comm: the communicator of the C-level sharded entry point (np_hip_comm_*, np_dist.hip).

    RCCL transport (default): the 128-byte id is drawn on rank 0 and handed to the other ranks by `exchange` (a callable
    bytes -> bytes that broadcasts rank 0's value; bench.py uses a torch.distributed broadcast).  world 1 with rccl=False
    needs no RCCL at all.

    Hosted transport (`all_gather=`): a callable (send: np.ndarray[uint8], recv: np.ndarray[uint8] of world * len(send))
    that fills recv with every rank's bytes in rank order -- MPI, gloo, shared memory, anything; `gloo_all_gather(group)`
    below is one over torch.distributed.  This is how ranks that share one GPU run the shipped protocol (RCCL refuses two
    ranks on a device).  deferred_status=True keeps failure propagation on the device, as the RCCL transport does."""

    def __init__(self, index: "api.MmapIndex", rank: int = 0, world: int = 1, exchange=None, rccl: bool = True,
                 all_gather=None, deferred_status: bool = False):
        L = api.lib()
        self._h = C.c_void_p()
        self.index = index
        self._cb = None
        if all_gather is not None:
            def _cb(_ctx, send, recv, nbytes):
                try:
                    s = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(int(nbytes),))
                    r = np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_uint8)), shape=(int(nbytes) * world,))
                    all_gather(s, r)
                    return 0
                except Exception:   # never unwind through the C frames
                    import traceback
                    traceback.print_exc()
                    return 1
            self._cb = api.ALL_GATHER_HOST_FN(_cb)   # keep the trampoline alive as long as the communicator
            api._check(L.np_hip_comm_create_hosted(index._h, rank, world, self._cb, None,
                                                   api.NP_COMM_DEFERRED_STATUS if deferred_status else 0, C.byref(self._h)))
            return
        idbuf = None
        if rccl:
            idbuf = C.create_string_buffer(128)
            if rank == 0:
                api._check(L.np_hip_comm_unique_id(idbuf))
            if exchange is not None:
                idbuf = C.create_string_buffer(exchange(bytes(idbuf.raw)), 128)
        api._check(L.np_hip_comm_create(index._h, idbuf, rank, world, C.byref(self._h)))

    def status(self):
        """(failed_rank, np_status) of the batches since the last call, (-1, 0) if all were healthy.  Synchronise first."""
        r, c = C.c_int32(), C.c_int32()
        api._check(api.lib().np_hip_comm_status(self._h, C.byref(r), C.byref(c)))
        return int(r.value), int(c.value)

    def info(self):
        """np_hip_comm_info: dict(transport "local" | "rccl" | "hosted", nranks, rccl_ranks = ncclCommCount or 0)."""
        t, n, r = C.c_int32(), C.c_int32(), C.c_int32()
        api._check(api.lib().np_hip_comm_info(self._h, C.byref(t), C.byref(n), C.byref(r)))
        return dict(transport={0: "local", 1: "rccl", 2: "hosted"}.get(int(t.value), "?"), nranks=int(n.value),
                    rccl_ranks=int(r.value))

    def close(self):
        h, self._h = self._h, None
        if h:
            api.lib().np_hip_comm_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def gloo_all_gather(group=None):
    """A hosted-transport all-gather over a torch.distributed (gloo) process group, for ShardComm(all_gather=...)."""
    import torch
    import torch.distributed as dist

    def fn(send, recv):
        ws = dist.get_world_size(group)
        out = [torch.empty(send.size, dtype=torch.uint8) for _ in range(ws)]
        dist.all_gather(out, torch.from_numpy(send.copy()), group=group)
        for r, o in enumerate(out):
            recv[r * send.size:(r + 1) * send.size] = o.numpy()
    return fn


class CShardedSearcher:
    """The whole two-collective protocol through ONE C call per batch (np_hip_search_batch_sharded): what a compiled
    host uses.  Outputs are torch tensors on the shard's device; every rank gets the global top-k."""

    def __init__(self, index: "api.MmapIndex", comm: ShardComm, stream=None):
        import torch
        self.torch = torch
        self.index, self.comm = index, comm
        self.device = torch.device("cuda", index.info.device)
        self.stream = torch.cuda.Stream(self.device) if stream is None else stream

    def search_batch_device(self, d_q, d_qoff, h_qoff, params, d_subset=None, out=None):
        t = self.torch
        B, k = len(h_qoff) - 1, max(params.top_k, 1)
        if out is None:
            with t.cuda.stream(self.stream):
                out = (t.zeros((B, k), dtype=t.int64, device=self.device), t.zeros((B, k), dtype=t.float32, device=self.device),
                       t.zeros(B, dtype=t.int32, device=self.device))
        p = params._c()
        hq = np.ascontiguousarray(h_qoff, np.int32)
        rc = api.lib().np_hip_search_batch_sharded(
            self.index._h, self.comm._h, C.c_void_p(d_q.data_ptr()), C.c_void_p(d_qoff.data_ptr()),
            hq.ctypes.data_as(C.c_void_p), B, self.index.embedding_dim(), C.byref(p),
            None if d_subset is None else C.c_void_p(d_subset.data_ptr()), -1 if d_subset is None else d_subset.numel(),
            C.c_void_p(out[0].data_ptr()), C.c_void_p(out[1].data_ptr()), C.c_void_p(out[2].data_ptr()),
            C.c_void_p(self.stream.cuda_stream))
        if rc != 0:
            msg = api.last_error()
            # a failed batch must not leave its status word behind for the next healthy one: drain it before raising
            self.stream.synchronize()
            self.comm.status()
            api._check(rc, msg)
        return out

    def search_batch(self, queries, params, subset=None):
        t = self.torch
        qs = [np.ascontiguousarray(q, np.float32) for q in queries]
        off = np.zeros(len(qs) + 1, np.int32)
        off[1:] = np.cumsum([q.shape[0] for q in qs])
        with t.cuda.stream(self.stream):
            dq = t.from_numpy(np.concatenate(qs, 0)).to(self.device)
            do = t.from_numpy(off).to(self.device)
            ds = None if subset is None else t.from_numpy(np.ascontiguousarray(subset, np.int64)).to(self.device)
            ids, sc, cnt = self.search_batch_device(dq, do, off, params, ds)
            ids, sc, cnt = ids.cpu().numpy(), sc.cpu().numpy(), cnt.cpu().numpy()
        self.stream.synchronize()
        rank, code = self.comm.status()   # a peer's failure abandons the batch on every rank: counts = -1 (np_dist.hip)
        if code or (cnt < 0).any():
            raise api.SearchError(f"Search failed: shard {rank} failed with status {code}; the batch was abandoned on every rank")
        return [api.QueryResult(i, ids[i, : cnt[i]].copy(), sc[i, : cnt[i]].copy()) for i in range(len(qs))]
