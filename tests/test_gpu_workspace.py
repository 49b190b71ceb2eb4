"""Per-context scratch under memory pressure (the default budget was fixed at open from hipMemGetInfo and
reserved lazily, so a second index on the device -- the documented "swap handles" reload pattern -- or a later allocation
made searches fail with OutOfMemory).  Needs a real MI355X."""
import numpy as np
import pytest

from helpers import hip_index, make_arrays, synth

import next_plaid_amd as npa

pytestmark = pytest.mark.gpu


def _same(a, b):
    return all(np.array_equal(x.passage_ids, y.passage_ids) and np.array_equal(x.scores, y.scores) for x, y in zip(a, b))


def test_two_indexes_on_one_device():
    """Open A, open B (the reload pattern keeps both resident), search both alternately: A's answers do not change."""
    spec, a = make_arrays(num_docs=4000, num_centroids=512, dim=128, nbits=4, doc_len_min=5, doc_len_max=60, seed=71)
    spec2, b = make_arrays(num_docs=3000, num_centroids=256, dim=64, nbits=2, doc_len_min=5, doc_len_max=40, seed=72)
    ha = hip_index(a)
    qa, _ = synth.make_queries(spec, 8, n_tokens=32, cen=a["centroids"])
    p = npa.SearchParameters(n_full_scores=128, top_k=10, n_ivf_probe=8)
    first = ha.search_batch(qa, p)
    hb = hip_index(b)
    qb, _ = synth.make_queries(spec2, 8, n_tokens=32, cen=b["centroids"])
    rb = hb.search_batch(qb, p)
    assert _same(ha.search_batch(qa, p), first)
    assert _same(hb.search_batch(qb, p), rb)
    hb.close()
    assert _same(ha.search_batch(qa, p), first)


def test_pool_shrinks_instead_of_out_of_memory():
    """2 M documents x 64 queries want a 7.7 GB candidate pool under the default budget.  The first call reserves it; then
    most of the free HBM is taken away and the NEXT call (a second context = a fresh workspace) must plan against what is
    free now: a smaller pool and more rounds, the same results -- not NP_ERR_OUT_OF_MEMORY."""
    import torch
    spec = synth.SynthSpec(num_docs=2_000_000, num_centroids=4096, dim=128, nbits=4, doc_len_min=24, doc_len_max=24, seed=73)
    cen = synth.centroids(spec)
    ix = npa.MmapIndex.synth(spec, centroids=cen, n_contexts=2, max_batch=64)
    qs, _ = synth.make_queries(spec, 64, n_tokens=32, cen=cen)
    # no threshold + a wide probe: hundreds of thousands of candidates per query, so the pool size matters
    p = npa.SearchParameters(n_full_scores=256, top_k=10, n_ivf_probe=64, centroid_score_threshold=None)
    first = ix.search_batch(qs, p)
    at_open = ix.workspace_bytes()
    free, _total = torch.cuda.mem_get_info()
    hog = torch.empty(max(free - (3 << 30), 1 << 20), dtype=torch.uint8, device="cuda")   # leave ~3 GiB
    try:
        again = ix.search_batch(qs, p)          # least-recently-used hand-out: this is the second context
        assert _same(again, first)
        assert ix.last_stats["n_rounds"] >= 1
        assert ix.workspace_bytes() < at_open   # the live budget is what np_hip_index_info reports
        third = ix.search_batch(qs, p)          # and the first context again
        assert _same(third, first)
    finally:
        del hog
        torch.cuda.empty_cache()
    # the tenant is gone: the budget grows back to its value at open instead of staying pinned to the small pool
    for _ in range(2):
        assert _same(ix.search_batch(qs, p), first)
    assert ix.workspace_bytes() == at_open
