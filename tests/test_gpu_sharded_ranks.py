"""The SHIPPED multi-rank code path at G >= 2 on a one-GPU box.

`bench.py --gpus N` runs np_hip_search_batch_sharded (np_dist.hip) -> np_hip_merge_packed.  RCCL refuses two ranks on one
device, so here G processes SHARE GPU 0 and the two all-gathers go through the hosted transport
(np_hip_comm_create_hosted: pinned staging + a gloo all-gather on the host) -- everything else is the code the 8-GPU run
executes: the per-rank record layout and striding, the communicator's buffer sizing at G > 1, the strided global cut,
the packed merge, the status words.  Results must equal the unsharded handle bit for bit, with and without a subset.
A rank whose local work fails must make every rank return an error instead of leaving its peers in a collective.
"""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest

from helpers import hip_index, make_arrays, synth

import next_plaid_amd as npa

pytestmark = pytest.mark.gpu

SPEC = dict(num_docs=6000, num_centroids=1024, dim=128, nbits=4, doc_len_min=5, doc_len_max=80, seed=55)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("G", [2, 3])
def test_merge_packed_equals_merge_topk_and_unsharded(G):
    """np_hip_merge_packed at G = 2, 3 on records packed from in-process shards (no transport at all): the record
    striding (ids | keys | scores | counts, one record per rank) gives the same merge as the four-array form and the
    unsharded search."""
    import torch
    from next_plaid_amd import api
    from next_plaid_amd.dist import HipShardBackend
    spec, a = make_arrays(**SPEC)
    full = hip_index(a)
    shards = [hip_index(a, shard_rank=r, shard_count=G) for r in range(G)]
    stream = torch.cuda.Stream()
    bes = [HipShardBackend(s, stream=stream) for s in shards]
    qs, _ = synth.make_queries(spec, 9, n_tokens=32, cen=a["centroids"])
    off = np.zeros(len(qs) + 1, np.int32)
    off[1:] = np.cumsum([q.shape[0] for q in qs])
    L = api.lib()
    for nfs, k in ((256, 10), (64, 7)):
        p = npa.SearchParameters(n_full_scores=nfs, top_k=k, n_ivf_probe=8)
        B = len(qs)
        with torch.cuda.stream(stream):
            dq = torch.from_numpy(np.concatenate(qs, 0)).cuda()
            do = torch.from_numpy(off).cuda()
            keys, states = zip(*[be.phase_a(dq, do, off, p) for be in bes])
            allk = torch.stack(list(keys), 0)
            # the layout np_dist.hip gathers: ids [B*k] i64 | keys [B*k] u64 | scores [B*k] f32 | counts [B] i32, 16-byte multiple
            o_keys, o_sc = B * k * 8, B * k * 16
            o_cnt = o_sc + B * k * 4
            rec = (o_cnt + B * 4 + 15) // 16 * 16 + 16      # + the status trailer of the real record (ignored here)
            records = torch.zeros(G * rec, dtype=torch.uint8, device="cuda")
            four = []
            for g, (be, st) in enumerate(zip(bes, states)):
                cut = be.select_cut(allk)
                base = records.data_ptr() + g * rec
                api._check(L.np_hip_search_phase_b(be.index._h, st[0], C.c_void_p(cut.data_ptr()), C.c_void_p(base),
                                                   C.c_void_p(base + o_sc), C.c_void_p(base + o_keys), C.c_void_p(base + o_cnt),
                                                   C.c_void_p(stream.cuda_stream)))
                be.end(st)
            stream.synchronize()
            raw = records.cpu().numpy().reshape(G, rec)
            ids = torch.from_numpy(np.stack([raw[g, :o_keys].view(np.int64) for g in range(G)])).cuda()
            kk = torch.from_numpy(np.stack([raw[g, o_keys:o_sc].view(np.int64) for g in range(G)])).cuda()
            sc = torch.from_numpy(np.stack([raw[g, o_sc:o_cnt].view(np.float32) for g in range(G)])).cuda()
            cnt = torch.from_numpy(np.stack([raw[g, o_cnt:o_cnt + B * 4].view(np.int32) for g in range(G)])).cuda()
            outs = []
            for packed in (True, False):
                oi = torch.zeros((B, k), dtype=torch.int64, device="cuda")
                os_ = torch.zeros((B, k), dtype=torch.float32, device="cuda")
                oc = torch.zeros(B, dtype=torch.int32, device="cuda")
                if packed:
                    api._check(L.np_hip_merge_packed(bes[0].index._h, C.c_void_p(records.data_ptr()), rec, o_keys, o_sc, o_cnt, G, B, k,
                                                     C.c_void_p(oi.data_ptr()), C.c_void_p(os_.data_ptr()), C.c_void_p(oc.data_ptr()),
                                                     C.c_void_p(stream.cuda_stream)))
                else:
                    api._check(L.np_hip_merge_topk(bes[0].index._h, C.c_void_p(ids.data_ptr()), C.c_void_p(sc.data_ptr()),
                                                   C.c_void_p(kk.data_ptr()), C.c_void_p(cnt.data_ptr()), G, B, k,
                                                   C.c_void_p(oi.data_ptr()), C.c_void_p(os_.data_ptr()), C.c_void_p(oc.data_ptr()),
                                                   C.c_void_p(stream.cuda_stream)))
                stream.synchronize()
                outs.append((oi.cpu().numpy(), os_.cpu().numpy(), oc.cpu().numpy()))
        ref = full.search_batch(qs, p)
        assert sum(int(c) for c in cnt.cpu().numpy().ravel()) > B * k, "every shard must contribute: the merge has real work"
        for i, r in enumerate(ref):
            for oi, os_, oc in outs:
                n = int(oc[i])
                assert n == len(r.passage_ids), (G, i, n)
                assert np.array_equal(oi[i, :n], r.passage_ids) and np.array_equal(os_[i, :n], r.scores), (G, i)


def _rank_main(rank, world, port, mode, q):
    """One rank = one process; all ranks use GPU 0.  Reports ("ok", rank) or ("fail", rank, message) through q."""
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        here = os.path.dirname(os.path.abspath(__file__))
        for p in (here, os.path.dirname(here), os.path.join(os.path.dirname(here), "next-plaid_amd")):
            if p not in sys.path:
                sys.path.insert(0, p)
        import torch
        import torch.distributed as dist
        from helpers import hip_index, make_arrays, synth
        import next_plaid_amd as npa
        from next_plaid_amd.dist import CShardedSearcher, ShardComm, gloo_all_gather
        dist.init_process_group("gloo", rank=rank, world_size=world)
        try:
            spec, a = make_arrays(**SPEC)
            full = hip_index(a)
            qs, _ = synth.make_queries(spec, 12, n_tokens=32, cen=a["centroids"])
            if mode == "parity":
                shard = hip_index(a, shard_rank=rank, shard_count=world)
                for deferred in (False, True):
                    comm = ShardComm(shard, rank, world, all_gather=gloo_all_gather(), deferred_status=deferred)
                    cs = CShardedSearcher(shard, comm)
                    subs = (None, np.arange(0, 6000, 7, dtype=np.int64), np.arange(100, 900, dtype=np.int64),
                            np.array([3, 5999, 7000, -1, 2500], np.int64))
                    for prm in (npa.SearchParameters(n_full_scores=256, top_k=10, n_ivf_probe=8),
                                npa.SearchParameters(n_full_scores=64, top_k=20, n_ivf_probe=4, centroid_score_threshold=None, precision=0),
                                npa.SearchParameters(n_full_scores=128, top_k=10, n_ivf_probe=4, centroid_score_threshold=None,
                                                     centroid_batch_size=300)):
                        for sub in subs:
                            got = cs.search_batch(qs, prm, sub)
                            ref = full.search_batch(qs, prm, subset=sub)
                            for i, (g, r) in enumerate(zip(got, ref)):
                                assert np.array_equal(g.passage_ids, r.passage_ids), (rank, deferred, i, g.passage_ids, r.passage_ids)
                                assert np.array_equal(g.scores, r.scores), (rank, deferred, i)
                    assert comm.status() == (-1, 0)
                    comm.close()
            else:   # "failure": the last rank's handle cannot take the batch in one slice (max_batch 4 < B = 12)
                bad = world - 1
                shard = hip_index(a, shard_rank=rank, shard_count=world, **({"max_batch": 4} if rank == bad else {}))
                prm = npa.SearchParameters(n_full_scores=256, top_k=10, n_ivf_probe=8)
                for deferred in (False, True):
                    comm = ShardComm(shard, rank, world, all_gather=gloo_all_gather(), deferred_status=deferred)
                    cs = CShardedSearcher(shard, comm)
                    for _ in range(2):   # the communicator stays usable: the second failing batch behaves like the first
                        with pytest.raises(npa.SearchError) as ei:
                            cs.search_batch(qs, prm)
                        msg = str(ei.value)
                        if rank == bad:
                            assert "one workspace slice" in msg, msg           # its own error, at once
                        else:
                            assert f"shard {bad} failed with status 2" in msg, msg
                    # ... and a batch every rank can take goes through afterwards on the same communicator
                    got = cs.search_batch(qs[:4], prm)
                    for g, r in zip(got, full.search_batch(qs[:4], prm)):
                        assert np.array_equal(g.passage_ids, r.passage_ids) and np.array_equal(g.scores, r.scores)
                    comm.close()
            dist.barrier()
        finally:
            dist.destroy_process_group()
        q.put(("ok", rank))
    except BaseException as e:   # noqa: BLE001 -- reported to the parent, which fails the test
        import traceback
        q.put(("fail", rank, "".join(traceback.format_exception(type(e), e, e.__traceback__))[-3000:]))


def _run_ranks(world, mode, timeout=420):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = []
    try:
        for _ in range(world):
            res.append(q.get(timeout=timeout))   # a hang (a rank stuck in a collective) fails here, not at the box's limit
    finally:
        for p in procs:
            p.join(timeout=20)
            if p.is_alive():
                p.kill()
    bad = [r for r in res if r[0] != "ok"]
    assert not bad, "\n".join(f"rank {r[1]}:\n{r[2]}" for r in bad)


@pytest.mark.parametrize("G", [2, 3])
def test_c_sharded_entry_ranks_share_one_gpu(G):
    """np_hip_search_batch_sharded from G processes (hosted all-gather): bit-equal to the unsharded handle, with and
    without a subset, dense and batched probe, host-checked and device-propagated status."""
    _run_ranks(G, "parity")


def test_failing_rank_returns_an_error_on_every_rank():
    """One rank's phase A fails (its slice is too small for the batch): it returns its own error, its peers return
    'shard r failed' -- nobody hangs, and the communicator serves the next batch."""
    _run_ranks(2, "failure")
