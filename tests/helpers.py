"""Shared helpers for the parity tests (test infrastructure; may import oracle/)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "next-plaid_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from next_plaid_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")

# Stated tolerances (north_star allows 1e-3 relative; the fp32 path is far inside it).
RTOL_F32 = 2e-5   # fp32 MFMA path vs oracle: summation-order differences only
RTOL_BF16 = 1e-3  # bf16 on the residual term only (precision=1): north_star's bound
RTOL_BF16_PLAIN = 4e-3  # precision=3: BOTH operands of the whole dot product rounded to bf16 (2^-8 per product); the
                        # config-5 comparison mode, measured 1.1e-3 on one-token queries -- outside north_star's bound,
                        # which is why it is not a default


def make_arrays(**kw):
    spec = synth.SynthSpec(**kw)
    return spec, synth.generate_arrays(spec)


def oracle_index(a):
    return O.OracleIndex(a["centroids"], a["bucket_weights"], a["ivf"], a["ivf_lengths"], a["doc_lengths"],
                         a["codes"], a["residuals"], a["nbits"])


def hip_index(a, **opts):
    import next_plaid_amd as npa
    return npa.MmapIndex.from_arrays(a["centroids"], a["bucket_weights"], a["ivf"], a["ivf_lengths"],
                                     a["doc_lengths"], a["codes"], a["residuals"], a["nbits"], **opts)


def to_oracle_params(p):
    return O.SearchParameters(n_full_scores=p.n_full_scores, top_k=p.top_k, n_ivf_probe=p.n_ivf_probe,
                              centroid_batch_size=p.centroid_batch_size,
                              centroid_score_threshold=p.centroid_score_threshold)


def assert_ranking_close(ids, scores, ref_ids, ref_scores, rtol, what=""):
    """Same length; scores within rtol; ids identical except inside groups of reference scores that are
    closer than the tolerance (where summation order may legitimately swap neighbours)."""
    ids, ref_ids = np.asarray(ids), np.asarray(ref_ids)
    scores, ref_scores = np.asarray(scores, np.float64), np.asarray(ref_scores, np.float64)
    assert ids.shape == ref_ids.shape, f"{what}: count {ids.shape} vs {ref_ids.shape}"
    if ids.size == 0:
        return
    tol = rtol * np.maximum(np.abs(ref_scores), 1.0)
    assert np.all(np.abs(scores - ref_scores) <= tol), \
        f"{what}: scores differ: {scores} vs {ref_scores} (max rel {np.max(np.abs(scores-ref_scores)/np.maximum(np.abs(ref_scores),1))})"
    assert np.all(np.diff(scores) <= tol[1:] * 2), f"{what}: scores not descending: {scores}"
    bad = np.nonzero(ids != ref_ids)[0]
    for i in bad:
        j = np.nonzero(ref_ids == ids[i])[0]
        if j.size == 0:
            # not in the reference top-k at all: only legitimate at the k-th place boundary, i.e. its
            # score must tie (within tolerance) with the reference's last kept score
            assert abs(scores[i] - ref_scores[-1]) <= 2 * tol[-1], \
                f"{what}: id {ids[i]} at rank {i} (score {scores[i]}) is not in the reference top-k {ref_ids} " \
                f"and is not a boundary tie with {ref_scores[-1]}"
            continue
        # a swapped id must sit in a near-tie with its reference counterpart
        assert abs(ref_scores[j[0]] - ref_scores[i]) <= 2 * tol[i], \
            f"{what}: rank {i}: id {ids[i]} vs {ref_ids[i]} is not a near-tie ({ref_scores[j[0]]} vs {ref_scores[i]})"
