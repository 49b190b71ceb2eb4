"""world_size-2 gloo test of the document-sharded protocol (next_plaid_amd/dist.py) on CPU: the
two all-gathers, the global cut and the merge reproduce the UNSHARDED oracle result exactly."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import O, oracle_index, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


SPEC = dict(num_docs=1200, num_centroids=128, dim=64, nbits=4, doc_len_min=0, doc_len_max=30, seed=123)


def _worker(rank, world, port, top_k, nfs, thr):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from shard_oracle_backend import OracleShardBackend
        from next_plaid_amd.dist import ShardedSearcher
        import next_plaid_amd as npa
        spec = synth.SynthSpec(**SPEC)
        a = synth.generate_arrays(spec)
        n = spec.num_docs
        b0, b1 = n * rank // world, n * (rank + 1) // world
        be = OracleShardBackend(a, b0, b1)
        qs, _ = synth.make_queries(spec, 5, n_tokens=12, cen=a["centroids"])
        off = np.zeros(len(qs) + 1, np.int32)
        off[1:] = np.cumsum([q.shape[0] for q in qs])
        flat = torch.from_numpy(np.concatenate(qs, 0))
        p = npa.SearchParameters(n_full_scores=nfs, top_k=top_k, n_ivf_probe=4, centroid_score_threshold=thr)
        ids, sc, cnt = ShardedSearcher([be]).search_batch_device(flat, torch.from_numpy(off), off, p)
        # bench.py overlaps batches on per-stream process groups: two searchers on their own groups, interleaved,
        # must give the same answer as the default group
        g1, g2 = dist.new_group(ranks=list(range(world))), dist.new_group(ranks=list(range(world)))
        s1, s2 = ShardedSearcher([be], group=g1), ShardedSearcher([be], group=g2)
        for ss in (s1, s2, s1):
            i2, c2, n2 = ss.search_batch_device(flat, torch.from_numpy(off), off, p)
            assert torch.equal(i2, ids) and torch.equal(c2, sc) and torch.equal(n2, cnt), rank
        full = oracle_index(a)
        po = O.SearchParameters(n_full_scores=nfs, top_k=top_k, n_ivf_probe=4, centroid_score_threshold=thr)
        for i, q in enumerate(qs):
            r = full.search(q, po)
            assert cnt[i] == len(r.passage_ids), (rank, i, int(cnt[i]), len(r.passage_ids))
            assert np.array_equal(ids[i, : cnt[i]].numpy(), r.passage_ids), (rank, i)
            assert np.array_equal(sc[i, : cnt[i]].numpy(), r.scores), (rank, i)
    finally:
        dist.destroy_process_group()


def _worker_subset(rank, world, port):
    """Sharded + subset: the eligible-centroid bitmap (search.rs:350-364) is OR-ed over the shards by one more small
    all-gather, so nprobe scaling and the probe see the WHOLE index's eligibility: bit-equal to the unsharded oracle."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from shard_oracle_backend import OracleShardBackend
        from next_plaid_amd.dist import ShardedSearcher
        import next_plaid_amd as npa
        spec = synth.SynthSpec(**SPEC)
        a = synth.generate_arrays(spec)
        n = spec.num_docs
        b0, b1 = n * rank // world, n * (rank + 1) // world
        be = OracleShardBackend(a, b0, b1)
        qs, _ = synth.make_queries(spec, 5, n_tokens=12, cen=a["centroids"])
        off = np.zeros(len(qs) + 1, np.int32)
        off[1:] = np.cumsum([q.shape[0] for q in qs])
        flat = torch.from_numpy(np.concatenate(qs, 0))
        full = oracle_index(a)
        n_diff = 0
        # subsets concentrated in ONE shard make the per-shard eligibility differ most from the global one
        for subset in (np.arange(0, n, 3), np.arange(5, 200, 2), np.array([7, 900, 1100, 1199, 5000, -2]), np.arange(n // 2, n)):
            subset = subset.astype(np.int64)
            for cbs in (100_000, 50):      # dense path (eligibility + nprobe scaling) and batched path (filter only)
                p = npa.SearchParameters(n_full_scores=64, top_k=6, n_ivf_probe=3, centroid_score_threshold=None,
                                         centroid_batch_size=cbs)
                ids, sc, cnt = ShardedSearcher([be]).search_batch_device(flat, torch.from_numpy(off), off, p,
                                                                         torch.from_numpy(subset))
                po = O.SearchParameters(n_full_scores=64, top_k=6, n_ivf_probe=3, centroid_score_threshold=None,
                                        centroid_batch_size=cbs)
                for i, q in enumerate(qs):
                    r = full.search(q, po, subset)
                    assert cnt[i] == len(r.passage_ids), (rank, i, cbs, int(cnt[i]), len(r.passage_ids))
                    assert np.array_equal(ids[i, : cnt[i]].numpy(), r.passage_ids), (rank, i, cbs)
                    assert np.array_equal(sc[i, : cnt[i]].numpy(), r.scores), (rank, i, cbs)
                    assert set(r.passage_ids.tolist()) <= set(subset.tolist())
                    # the case is only meaningful if per-shard eligibility WOULD have changed something
                    loc = be.ix.search(q, po, (subset[(subset >= b0) & (subset < b1)] - b0), trace=True).trace
                    n_diff += int(cbs > 1000 and loc.cells.size != full.search(q, po, subset, trace=True).trace.cells.size)
        tot = torch.tensor([n_diff])
        dist.all_reduce(tot)
        assert int(tot) > 0, "no case separated per-shard from global eligibility"
    finally:
        dist.destroy_process_group()


def test_sharded_subset_world2_gloo():
    mp.spawn(_worker_subset, args=(2, _free_port()), nprocs=2, join=True)


@pytest.mark.parametrize("top_k,nfs,thr", [(5, 64, None), (10, 16, 0.3)])
def test_sharded_protocol_world2_gloo(top_k, nfs, thr):
    mp.spawn(_worker, args=(2, _free_port(), top_k, nfs, thr), nprocs=2, join=True)


def test_shard_range_partition():
    # contiguous doc ranges [N*r/G, N*(r+1)/G) cover [0,N) without overlap (np_open_opts.shard_*)
    for n in (0, 1, 7, 1000003):
        for g in (1, 2, 3, 8):
            edges = [n * r // g for r in range(g + 1)]
            assert edges[0] == 0 and edges[-1] == n and all(b >= a for a, b in zip(edges, edges[1:]))
