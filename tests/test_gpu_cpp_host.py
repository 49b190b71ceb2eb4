"""The C++ host mirror (next-plaid_amd/cpp/next_plaid.hpp, the stand-in for the Rust wrapper) drives
the C ABI end to end: on-disk index -> MmapIndex::load -> search_batch, equal to the Python mirror
and within tolerance of the oracle.  Needs a real MI355X."""
import json
import os
import subprocess

import numpy as np
import pytest

from helpers import ROOT, RTOL_F32, assert_ranking_close, hip_index, make_arrays, oracle_index, synth, to_oracle_params

import next_plaid_amd as npa

pytestmark = pytest.mark.gpu


def test_cpp_cli_matches_python_and_oracle(tmp_path):
    cli = os.path.join(ROOT, "next-plaid_amd", "cpp", "np_search")
    subprocess.check_call(["make", "-C", os.path.dirname(cli)], stdout=subprocess.DEVNULL)
    spec, a = make_arrays(num_docs=1500, num_centroids=256, dim=128, nbits=4, doc_len_min=10, doc_len_max=50, seed=71)
    idx = tmp_path / "index"
    synth.write_index(str(idx), a, chunk_docs=400)
    qs, _ = synth.make_queries(spec, 5, n_tokens=32, cen=a["centroids"])
    qf = tmp_path / "q.f32"
    np.concatenate(qs, 0).astype("<f4").tofile(qf)
    out = subprocess.check_output([cli, str(idx), str(qf), "5", "32", "7", "6", "128", "-1"], text=True)
    rows = [json.loads(l) for l in out.strip().splitlines()]
    p = npa.SearchParameters(top_k=7, n_ivf_probe=6, n_full_scores=128, centroid_score_threshold=None)
    py = hip_index(a).search_batch(qs, p)
    orc = oracle_index(a).search_batch(qs, to_oracle_params(p))
    assert len(rows) == 5
    for i, (r, y, o) in enumerate(zip(rows, py, orc)):
        assert r["query_id"] == i
        assert r["passage_ids"] == y.passage_ids.tolist()
        assert np.allclose(r["scores"], y.scores, rtol=1e-7)
        assert_ranking_close(r["passage_ids"], r["scores"], o.passage_ids, o.scores, RTOL_F32, f"cpp q{i}")
    # error mapping: a missing index is Error::IndexLoad (exit code 1, message on stderr)
    pr = subprocess.run([cli, str(tmp_path / "nope"), str(qf), "1", "32"], capture_output=True, text=True)
    assert pr.returncode == 1 and "Index load failed" in pr.stderr
