#!/usr/bin/env python3
"""Reference-side pinning recipe, part 1 (runs here; needs no GPU and no cargo).

Writes, into OUT (default tools/ref_golden/out/):
  index/        the 2 000-document golden index of tests/golden/make_golden.py as an index DIRECTORY in the crate's on-disk
                format (np_hip_index_write_dir: the file set of write_index_from_encoded_chunks, index.rs:373-528) -- what
                `MmapIndex::load` (index.rs:1026-1139) reads;
  golden.json   the six cases of make_golden.py (dense / batched probe, with and without threshold, with a subset): per case the
                SearchParameters, the subset, and per query the token matrix and the expected passage ids and scores, copied from
                tests/golden/search_2000.npz (minted by the independent numpy restatement oracle/plaid_numpy.py).

Part 2 is golden_search.rs next to this file: a Rust integration test for the next-plaid crate that loads index/, runs the cases
through the crate's own `MmapIndex::search` and compares.  It cannot run in this image (no cargo / rustc); anyone with a Rust
toolchain turns "parity unpinned by the reference at search()" into one command (README.md here).
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "next-plaid_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import make_golden as MG  # noqa: E402
from next_plaid_amd import api, synth  # noqa: E402


def main(out):
    os.makedirs(out, exist_ok=True)
    spec = synth.SynthSpec(**MG.GOLDEN_SPEC)
    a = synth.generate_arrays(spec)
    idx = os.path.join(out, "index")
    api.write_index_dir(idx, a["centroids"], a["bucket_weights"], a["doc_lengths"], a["codes"], a["residuals"], spec.nbits,
                        ivf=a["ivf"], ivf_lengths=a["ivf_lengths"], bucket_cutoffs=a["bucket_cutoffs"])
    g = np.load(os.path.join(ROOT, "tests", "golden", "search_2000.npz"))
    qs = g["queries"]
    cases = []
    for name, kw, sub in MG.CASES:
        subset = None if sub is None else list(range(0, spec.num_docs, 2))
        cases.append(dict(
            name=name,
            params=dict(batch_size=2000, n_full_scores=kw["n_full_scores"], top_k=kw["top_k"], n_ivf_probe=kw["n_ivf_probe"],
                        centroid_batch_size=kw.get("centroid_batch_size", 100_000),
                        centroid_score_threshold=kw.get("centroid_score_threshold")),
            subset=subset,
            queries=[dict(ids=g[f"{name}_q{qi}_ids"].astype(int).tolist(),
                          scores=[float(x) for x in g[f"{name}_q{qi}_scores"]]) for qi in range(qs.shape[0])]))
    doc = dict(source="tests/golden/search_2000.npz (oracle/plaid_numpy.py on tests/golden/make_golden.py's seeded index)",
               index=dict(num_documents=spec.num_docs, num_partitions=spec.num_centroids, embedding_dim=spec.dim, nbits=spec.nbits),
               rtol=2e-5,
               query_tokens=[[[float(x) for x in row] for row in q] for q in qs], cases=cases)
    with open(os.path.join(out, "golden.json"), "w") as f:
        json.dump(doc, f)
    print("wrote", idx, "(%d files)" % len(os.listdir(idx)), "and", os.path.join(out, "golden.json"))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "out"))
