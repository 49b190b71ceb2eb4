"""np_hip_index_write_dir (host only): the canonical file set of write_index_from_encoded_chunks (index.rs:373-528) in the
formats of SURVEY.md Appendix A.  Checked against numpy.save byte for byte (NPY 1.0, 64-byte aligned header,
mmap.rs:1176-1250), against the Python statement of the same writer (synth.write_index), and read back through the
loader (np_hip_index_probe_dir: the parsing half of MmapIndex::load)."""
import filecmp
import json
import os

import numpy as np
import pytest

from helpers import make_arrays, synth

import next_plaid_amd as npa


def write_native(path, a, **kw):
    npa.write_index_dir(str(path), a["centroids"], a["bucket_weights"], a["doc_lengths"], a["codes"], a["residuals"],
                        a["nbits"], **kw)


@pytest.mark.parametrize("dim,nbits,K,chunk", [(128, 4, 64, 150), (48, 2, 33, 1000), (40, 1, 16, 7), (24, 8, 20, 64)])
def test_same_files_as_the_python_writer(tmp_path, dim, nbits, K, chunk):
    spec, a = make_arrays(num_docs=500, num_centroids=K, dim=dim, nbits=nbits, doc_len_min=0, doc_len_max=25, seed=dim)
    d_py, d_c = tmp_path / "py", tmp_path / "c"
    synth.write_index(str(d_py), a, chunk_docs=chunk)
    write_native(d_c, a, ivf=a["ivf"], ivf_lengths=a["ivf_lengths"], bucket_cutoffs=a["bucket_cutoffs"], chunk_docs=chunk)
    names = sorted(os.listdir(d_py))
    assert sorted(os.listdir(d_c)) == names            # no temporary file left behind, nothing missing
    for n in names:
        if n.endswith(".npy"):
            assert filecmp.cmp(d_py / n, d_c / n, shallow=False), f"{n} differs from numpy.save"
        else:
            assert json.load(open(d_py / n)) == json.load(open(d_c / n)), n
    meta = json.load(open(d_c / "metadata.json"))
    T = int(a["doc_lengths"].sum())
    assert meta == dict(num_chunks=-(-500 // chunk), nbits=nbits, num_partitions=K, num_embeddings=T, avg_doclen=T / 500,
                        num_documents=500, embedding_dim=dim, next_plaid_compatible=True)
    info = npa.probe_index_dir(str(d_c))
    assert (info.num_documents, info.num_embeddings, info.num_partitions, info.embedding_dim, info.nbits) == (500, T, K, dim, nbits)
    # embedding_offset of each chunk = tokens before it (index.rs:436-441)
    off = 0
    for i in range(meta["num_chunks"]):
        cm = json.load(open(d_c / f"{i}.metadata.json"))
        assert cm["embedding_offset"] == off and cm["num_documents"] == len(json.load(open(d_c / f"doclens.{i}.json")))
        off += cm["num_embeddings"]
    assert off == T


def test_posting_lists_built_from_the_codes(tmp_path):
    """index.rs:479-504: per centroid the ascending unique ids of the documents that use it."""
    spec, a = make_arrays(num_docs=400, num_centroids=50, dim=64, nbits=4, doc_len_min=0, doc_len_max=40, seed=3)
    write_native(tmp_path, a, chunk_docs=90)           # no ivf passed
    ivf, il = np.load(tmp_path / "ivf.npy"), np.load(tmp_path / "ivf_lengths.npy")
    assert ivf.dtype == np.int64 and il.dtype == np.int32
    ref_ivf, ref_il = synth.build_ivf(a["codes"], a["doc_lengths"], 50)
    assert np.array_equal(il, ref_il) and np.array_equal(ivf, ref_ivf)
    assert np.array_equal(ivf, a["ivf"]) and np.array_equal(il, a["ivf_lengths"])
    assert not os.path.exists(tmp_path / "bucket_cutoffs.npy")   # optional input, optional file (codec.rs:564-576)
    npa.probe_index_dir(str(tmp_path))


def test_overwrite_clears_derived_caches_and_empty_index(tmp_path):
    spec, a = make_arrays(num_docs=30, num_centroids=8, dim=32, nbits=2, doc_len_min=1, doc_len_max=5, seed=4)
    for stale in ("merged_codes.npy", "merged_residuals.npy", "merged_codes.npy.manifest.json"):
        (tmp_path / stale).write_bytes(b"stale")
    write_native(tmp_path, a)
    assert not any(n.startswith("merged_") for n in os.listdir(tmp_path))   # mmap.rs:1714-1743: derived, rebuilt on load
    assert not any(".tmp." in n for n in os.listdir(tmp_path))
    # zero documents: no chunk at all (chunks.len() / ceil(0 / batch_size) = 0, index.rs:420, 702), avg_doclen 0.0 (:387-391)
    e = tmp_path / "empty"
    npa.write_index_dir(str(e / "nested" / "dir"), a["centroids"], a["bucket_weights"], np.zeros(0, np.int64),
                        np.zeros(0, np.int64), np.zeros((0, 8), np.uint8), 2)
    m = json.load(open(e / "nested" / "dir" / "metadata.json"))
    assert m["num_documents"] == 0 and m["avg_doclen"] == 0.0 and m["num_chunks"] == 0
    assert not (e / "nested" / "dir" / "0.residuals.npy").exists()
    assert npa.probe_index_dir(str(e / "nested" / "dir")).num_documents == 0      # the loader takes it
    # a SHORTER index written over a longer one leaves no chunk of the old one behind, and no stale cutoffs
    d = tmp_path / "shrink"
    spec2, big = make_arrays(num_docs=30, num_centroids=8, dim=32, nbits=2, doc_len_min=1, doc_len_max=5, seed=4)
    npa.write_index_dir(str(d), big["centroids"], big["bucket_weights"], big["doc_lengths"], big["codes"], big["residuals"], 2,
                        bucket_cutoffs=np.zeros(3, np.float32), chunk_docs=10)
    assert (d / "2.codes.npy").exists() and (d / "bucket_cutoffs.npy").exists()
    n10 = int(big["doc_lengths"][:10].sum())
    npa.write_index_dir(str(d), big["centroids"], big["bucket_weights"], big["doc_lengths"][:10], big["codes"][:n10],
                        big["residuals"][:n10], 2, chunk_docs=10)
    assert not (d / "1.codes.npy").exists() and not (d / "doclens.2.json").exists() and not (d / "bucket_cutoffs.npy").exists()
    assert npa.probe_index_dir(str(d)).num_documents == 10


def test_writer_errors(tmp_path):
    spec, a = make_arrays(num_docs=20, num_centroids=8, dim=32, nbits=4, doc_len_min=2, doc_len_max=4, seed=5)
    bad = dict(a)
    bad["codes"] = a["codes"].copy()
    bad["codes"][3] = 8
    with pytest.raises(ValueError, match="outside"):    # NP_ERR_INVALID_ARGUMENT
        write_native(tmp_path / "x", bad)
    with pytest.raises(npa.ShapeError):
        npa.write_index_dir(str(tmp_path / "y"), a["centroids"], a["bucket_weights"], a["doc_lengths"], a["codes"][:-1],
                            a["residuals"], 4)
    with pytest.raises(npa.CodecError):
        npa.write_index_dir(str(tmp_path / "z"), a["centroids"], a["bucket_weights"][:3], a["doc_lengths"], a["codes"],
                            a["residuals"], 4)
    ivf_bad = a["ivf"].copy()
    ivf_bad[0], ivf_bad[1] = ivf_bad[1], ivf_bad[0]                # not ascending inside the first non-trivial list
    if int(a["ivf_lengths"][0]) >= 2:
        with pytest.raises(ValueError, match="ascending"):
            npa.write_index_dir(str(tmp_path / "v"), a["centroids"], a["bucket_weights"], a["doc_lengths"], a["codes"],
                                a["residuals"], 4, ivf=ivf_bad, ivf_lengths=a["ivf_lengths"])
    neg = a["ivf_lengths"].copy()
    neg[0] = -1
    with pytest.raises(ValueError, match="negative"):
        npa.write_index_dir(str(tmp_path / "w"), a["centroids"], a["bucket_weights"], a["doc_lengths"], a["codes"],
                            a["residuals"], 4, ivf=a["ivf"], ivf_lengths=neg)
    blocker = tmp_path / "file"
    blocker.write_text("x")
    with pytest.raises(npa.NextPlaidError):        # a file where the directory should go: Io, nothing half-written elsewhere
        write_native(blocker / "sub", a)


def test_cpp_mirror_writes_a_loadable_index(tmp_path):
    """next_plaid::write_index (next_plaid.hpp) -> the loader parses it; sizes and IVF as the Python statement gives them."""
    import subprocess
    from helpers import ROOT
    exe = tmp_path / "write_index"
    csrc = os.path.join(ROOT, "next-plaid_amd", "csrc")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", str(exe), os.path.join(ROOT, "tests", "cpp", "write_index.cpp"),
                           "-L", csrc, "-lnextplaid_hip", f"-Wl,-rpath,{csrc}"])
    d = tmp_path / "idx"
    out = subprocess.run([str(exe), str(d)], capture_output=True, text=True, timeout=120).stdout.split("\n")
    T = sum(i % 5 for i in range(37))
    assert out[0] == f"ok 37 {T} 12 24 2", out
    assert out[1] == "codec 4", out
    lens = np.array([i % 5 for i in range(37)], np.int64)
    codes = np.concatenate([np.load(d / f"{i}.codes.npy") for i in range(4)])
    assert np.array_equal(codes, np.arange(T) * 7 % 12)
    ivf, il = synth.build_ivf(codes, lens, 12)
    assert np.array_equal(np.load(d / "ivf.npy"), ivf) and np.array_equal(np.load(d / "ivf_lengths.npy"), il)
    assert json.load(open(d / "metadata.json"))["num_chunks"] == 4 and np.load(d / "3.residuals.npy").shape[1] == 6
