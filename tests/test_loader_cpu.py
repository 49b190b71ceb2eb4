"""On-disk format compatibility of the C-ABI loader, on the host only (np_hip_index_probe_dir runs the same
parser and checks as np_hip_index_open -- MmapIndex::load, index.rs:1026-1139 -- without a device).
Formats: SURVEY.md Appendix A; NPY v1/v2 headers mmap.rs:659-749; fast-plaid dtypes mmap.rs:1780-1808."""
import json
import os
import struct

import numpy as np
import pytest

from helpers import make_arrays, synth

import next_plaid_amd as npa


@pytest.fixture()
def index_dir(tmp_path):
    spec, a = make_arrays(num_docs=300, num_centroids=64, dim=32, nbits=4, doc_len_min=3, doc_len_max=20, seed=11)
    p = str(tmp_path / "idx")
    synth.write_index(p, a, chunk_docs=128)      # 3 chunks
    return p, a


def test_probe_reports_the_written_geometry(index_dir):
    p, a = index_dir
    info = npa.probe_index_dir(p)
    assert info.num_documents == 300 and info.num_partitions == 64 and info.embedding_dim == 32 and info.nbits == 4
    assert info.num_embeddings == int(a["doc_lengths"].sum()) == info.shard_embeddings
    assert abs(info.avg_doclen - a["doc_lengths"].mean()) < 1e-9
    assert info.device == -1 and info.shard_doc_begin == 0 and info.shard_doc_end == 300


def _rewrite_npy_v2(path):
    """Same array, NumPy format 2.0 header (u32 header length), accepted on read (mmap.rs:678-694)."""
    arr = np.load(path)
    hdr = ("{'descr': '%s', 'fortran_order': False, 'shape': %s, }" % (arr.dtype.str, repr(arr.shape))).encode()
    pad = 64 - (12 + len(hdr) + 1) % 64
    hdr = hdr + b" " * (pad % 64) + b"\n"
    with open(path, "wb") as f:
        f.write(b"\x93NUMPY\x02\x00" + struct.pack("<I", len(hdr)) + hdr + arr.tobytes())


def test_fast_plaid_dtypes_and_v2_headers_are_accepted(index_dir):
    p, a = index_dir
    np.save(os.path.join(p, "ivf_lengths.npy"), a["ivf_lengths"].astype("<i8"))        # fast-plaid writes i64
    res = np.load(os.path.join(p, "0.residuals.npy"))
    np.save(os.path.join(p, "0.residuals.npy"), res.astype("<u1"))
    _rewrite_npy_v2(os.path.join(p, "centroids.npy"))
    _rewrite_npy_v2(os.path.join(p, "1.codes.npy"))
    info = npa.probe_index_dir(p)
    assert info.num_documents == 300 and info.num_partitions == 64


def test_fast_plaid_float16_codec_files_are_accepted(index_dir):
    """fast-plaid writes centroids / bucket_weights as '<f2'; MmapIndex::load converts them to '<f4' first
    (mmap.rs:1757-1778).  The loader widens them in memory (values: test_gpu_parity.test_float16_index_files)."""
    p, a = index_dir
    np.save(os.path.join(p, "centroids.npy"), a["centroids"].astype("<f2"))
    np.save(os.path.join(p, "bucket_weights.npy"), a["bucket_weights"].astype("<f2"))
    info = npa.probe_index_dir(p)
    assert info.num_documents == 300 and info.num_partitions == 64 and info.embedding_dim == 32


def test_metadata_counts_are_inferred_when_zero(index_dir):
    p, a = index_dir
    m = json.load(open(os.path.join(p, "metadata.json")))
    m["num_documents"], m["num_embeddings"], m["avg_doclen"] = 0, 0, 0.0     # index.rs:131-155
    json.dump(m, open(os.path.join(p, "metadata.json"), "w"))
    info = npa.probe_index_dir(p)
    assert info.num_documents == 300 and info.num_embeddings == int(a["doc_lengths"].sum())
    assert abs(info.avg_doclen - a["doc_lengths"].mean()) < 1e-9


def test_probe_reports_the_abi_version(index_dir):
    """np_info of ABI v6 (np_hip_abi_version says so before any struct is handed over): abi_version 6, workspace_bytes (the live scratch budget) 0 for a host-only probe."""
    p, a = index_dir
    info = npa.probe_index_dir(p)
    assert info.abi_version == 6 == npa.api.lib().np_hip_abi_version() and info.workspace_bytes == 0 and info.device == -1


def test_merged_cache_and_extra_files_are_ignored(index_dir):
    p, a = index_dir
    np.save(os.path.join(p, "merged_codes.npy"), np.zeros(7, "<i8"))             # derived cache: never read
    open(os.path.join(p, "merged_codes.manifest.json"), "w").write("{}")
    open(os.path.join(p, "metadata.db"), "wb").write(b"sqlite")
    assert npa.probe_index_dir(p).num_documents == 300


@pytest.mark.parametrize("breakage,exc,needle", [
    ("no_metadata", npa.IndexLoadError, "metadata"),
    ("no_weights", npa.CodecError, "bucket_weights"),
    ("bad_nbits", npa.CodecError, "nbits"),
    ("f64_centroids", npa.IndexLoadError, "dtype"),
    ("res_width", npa.ShapeError, "residuals"),
    ("short_codes", npa.IndexLoadError, "doclens"),
    ("code_range", npa.IndexLoadError, "code"),
    ("ivf_range", npa.IndexLoadError, "ivf"),
    ("ivf_len_rows", npa.ShapeError, "ivf_lengths"),
    ("big_endian", npa.IndexLoadError, "dtype"),
    ("missing_chunk", npa.NextPlaidError, "2.codes.npy"),
])
def test_broken_directories_fail_with_the_reference_error_kind(index_dir, breakage, exc, needle):
    p, a = index_dir
    j = lambda n: os.path.join(p, n)
    if breakage == "no_metadata":
        os.remove(j("metadata.json"))
    elif breakage == "no_weights":
        os.remove(j("bucket_weights.npy"))                                       # codec.rs:428-431
    elif breakage == "bad_nbits":
        m = json.load(open(j("metadata.json")))
        m["nbits"] = 3
        json.dump(m, open(j("metadata.json"), "w"))
    elif breakage == "f64_centroids":
        np.save(j("centroids.npy"), a["centroids"].astype("<f8"))
    elif breakage == "res_width":
        np.save(j("0.residuals.npy"), np.load(j("0.residuals.npy"))[:, :-1].copy())
    elif breakage == "short_codes":
        np.save(j("1.codes.npy"), np.load(j("1.codes.npy"))[:-5].copy())
    elif breakage == "code_range":
        c = np.load(j("0.codes.npy"))
        c[3] = 64
        np.save(j("0.codes.npy"), c)
    elif breakage == "ivf_range":
        v = np.load(j("ivf.npy"))
        v[0] = 300
        np.save(j("ivf.npy"), v)
    elif breakage == "ivf_len_rows":
        np.save(j("ivf_lengths.npy"), a["ivf_lengths"][:-1].astype("<i4"))
    elif breakage == "big_endian":
        np.save(j("ivf.npy"), a["ivf"].astype(">i8"))
    elif breakage == "missing_chunk":
        os.remove(j("2.codes.npy"))
    with pytest.raises(exc) as e:
        npa.probe_index_dir(p)
    assert needle in str(e.value), str(e.value)


def test_probe_of_a_missing_directory():
    with pytest.raises(npa.IndexLoadError):
        npa.probe_index_dir("/nonexistent/next-plaid-index")
