#!/usr/bin/env python3
"""Tuning sweep for one stage: builds the bench workload once, then times the stages under different
NP_* environment knobs (read by the library on every call).  usage: s4_sweep.py 'K=V,K=V' 'K=V' ..."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "next-plaid_amd"))
import next_plaid_amd as npa
from next_plaid_amd import synth


def main():
    docs = int(os.environ.get("SWEEP_DOCS", "1000000"))
    spec = synth.SynthSpec(num_docs=docs, num_centroids=65536, dim=128, nbits=4, doc_len_min=300, doc_len_max=300, seed=1236)
    cen = synth.centroids(spec)
    ix = npa.MmapIndex.synth(spec, centroids=cen, device=0, max_batch=64, n_contexts=1)
    prm = npa.SearchParameters(n_full_scores=4096, top_k=10, n_ivf_probe=32, centroid_score_threshold=0.4,
                               precision=int(os.environ.get("SWEEP_PREC", "2")))
    qs, _ = synth.make_queries(spec, 64, n_tokens=32, cen=cen)
    base = None
    for cfg in sys.argv[1:] or [""]:
        kv = dict(x.split("=") for x in cfg.split(",") if x)
        for k, v in kv.items():
            os.environ[k] = v
        best = None
        for _ in range(4):
            res = ix.search_batch(qs, prm)
            st = ix.last_stats
            if best is None or st["ms_total"] < best["ms_total"]:
                best = st
        sig = [(tuple(r.passage_ids.tolist()), tuple(r.scores.tolist())) for r in res]
        if base is None:
            base = sig
        same = sig == base
        print(cfg or "(default)", "same_as_first=%s" % same,
              {k: round(v, 3) for k, v in best.items() if k.startswith("ms_")}, flush=True)
        for k in kv:
            del os.environ[k]


if __name__ == "__main__":
    main()
