"""tools/ref_golden: the reference-side pinning recipe.  The Rust half (golden_search.rs) needs cargo and
cannot run here; what CAN be checked here is that the exported bundle is what the Rust test expects: the index directory is
one this repository's loader parses (MmapIndex::load's file set), an index loaded from it answers the golden cases with the
golden ids and scores (CPU oracle), golden.json is exactly tests/golden/search_2000.npz, and the Rust source names the
crate's real API (checked against the citations, not compiled)."""
import json
import os
import subprocess
import sys

import numpy as np

from helpers import GOLDEN, O, ROOT

import next_plaid_amd as npa


def test_exported_bundle_answers_the_golden_cases(tmp_path):
    out = str(tmp_path / "out")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ref_golden", "export_golden.py"), out], capture_output=True,
                       text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    doc = json.load(open(os.path.join(out, "golden.json")))
    info = npa.probe_index_dir(os.path.join(out, "index"))          # the loader's own parse + validation, no device
    assert info.num_documents == doc["index"]["num_documents"] == 2000 and info.num_partitions == 512
    g = np.load(os.path.join(GOLDEN, "search_2000.npz"))
    assert np.array_equal(np.asarray(doc["query_tokens"], np.float32), g["queries"])     # f32 -> JSON double -> f32 is exact
    from oracle import npy_index
    a = npy_index.read_index(os.path.join(out, "index"))
    ox = O.OracleIndex(a["centroids"], a["bucket_weights"], a["ivf"], a["ivf_lengths"], a["doc_lengths"], a["codes"], a["residuals"],
                       a["nbits"])
    assert len(doc["cases"]) == 6
    for case in doc["cases"]:
        p = case["params"]
        po = O.SearchParameters(n_full_scores=p["n_full_scores"], top_k=p["top_k"], n_ivf_probe=p["n_ivf_probe"],
                                centroid_batch_size=p["centroid_batch_size"], centroid_score_threshold=p["centroid_score_threshold"])
        subset = None if case["subset"] is None else np.asarray(case["subset"], np.int64)
        for qi, want in enumerate(case["queries"]):
            assert want["ids"] == g[f"{case['name']}_q{qi}_ids"].tolist()
            res = ox.search(np.asarray(doc["query_tokens"][qi], np.float32), po, subset)
            assert res.passage_ids.tolist() == want["ids"], (case["name"], qi)
            assert np.allclose(res.scores, want["scores"], rtol=doc["rtol"], atol=0)


def test_rust_half_names_the_crates_api():
    src = open(os.path.join(ROOT, "tools", "ref_golden", "golden_search.rs")).read()
    for needle in ("use next_plaid::index::MmapIndex;", "use next_plaid::SearchParameters;", "MmapIndex::load(", ".search(q, &params, subset.as_deref())",
                   "batch_size:", "n_full_scores:", "top_k:", "n_ivf_probe:", "centroid_batch_size:", "centroid_score_threshold:",
                   "got.passage_ids", "got.scores", "num_documents()", "num_partitions()", "embedding_dim()"):
        assert needle in src, needle
