"""Host-side mirror of the next-plaid crate's search API over the C ABI (include/nextplaid_hip.h).

Names, argument meaning and error behaviour follow the reference:
  MmapIndex.load / search / search_batch / accessors  -> next-plaid/src/index.rs:1026-1312
  SearchParameters / QueryResult                       -> next-plaid/src/search.rs:26-80
  error classes                                        -> next-plaid/src/error.rs:9-66
There is NO CPU fallback here: a missing library or GPU raises (DeviceUnavailableError); the
Rust wrapper is where a CPU fallback would live.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(os.path.dirname(_PKG), "csrc", "libnextplaid_hip.so")


def library_path() -> str:
    return os.environ.get("NEXTPLAID_HIP_LIB", _LIB)


# ---- errors (error.rs) ------------------------------------------------------------------------

class NextPlaidError(RuntimeError):
    pass


class IndexLoadError(NextPlaidError):
    """Error::IndexLoad"""


class SearchError(NextPlaidError):
    """Error::Search"""


class ShapeError(NextPlaidError):
    """Error::Shape"""


class CodecError(NextPlaidError):
    """Error::Codec"""


class IoError(NextPlaidError):
    """Error::Io / Error::Json"""


class DeviceUnavailableError(NextPlaidError):
    """No usable gfx950 device or the HIP library is missing."""


_ERR = {1: IndexLoadError, 2: SearchError, 3: ShapeError, 4: CodecError, 5: IoError,
        6: DeviceUnavailableError, 7: MemoryError, 8: ValueError}


# ---- C structs -----------------------------------------------------------------------------------

class np_open_opts(C.Structure):
    _fields_ = [("device", C.c_int32), ("shard_rank", C.c_int32), ("shard_count", C.c_int32),
                ("n_contexts", C.c_int32), ("max_batch", C.c_int32), ("max_query_tokens", C.c_int32),
                ("workspace_bytes", C.c_int64)]


class np_search_params(C.Structure):
    _fields_ = [("top_k", C.c_int32), ("n_full_scores", C.c_int32), ("n_ivf_probe", C.c_int32),
                ("centroid_batch_size", C.c_int32), ("centroid_score_threshold", C.c_float),
                ("has_threshold", C.c_int32), ("precision", C.c_int32)]


class np_info(C.Structure):
    _fields_ = [("num_documents", C.c_int64), ("num_embeddings", C.c_int64), ("num_partitions", C.c_int64),
                ("embedding_dim", C.c_int32), ("nbits", C.c_int32), ("avg_doclen", C.c_double),
                ("shard_doc_begin", C.c_int64), ("shard_doc_end", C.c_int64), ("shard_embeddings", C.c_int64),
                ("device_bytes", C.c_int64), ("device", C.c_int32), ("abi_version", C.c_int32),
                ("workspace_bytes", C.c_int64)]


class np_stats(C.Structure):
    _fields_ = [("ms_total", C.c_float), ("ms_centroid", C.c_float), ("ms_probe", C.c_float),
                ("ms_candidates", C.c_float), ("ms_approx", C.c_float), ("ms_select", C.c_float),
                ("ms_exact", C.c_float), ("ms_topk", C.c_float),
                ("n_cells", C.c_int64), ("n_ivf_ids", C.c_int64), ("n_candidates", C.c_int64),
                ("n_cand_tokens", C.c_int64), ("n_exact_docs", C.c_int64), ("n_exact_tokens", C.c_int64),
                ("n_cand_codes", C.c_int64), ("n_queries", C.c_int32), ("n_rounds", C.c_int32),
                ("n_survivors", C.c_int64), ("n_cand_dcodes", C.c_int64), ("n_level2", C.c_int64),
                ("ms_hot_level", C.c_float), ("reserved0", C.c_int32), ("n_level0", C.c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if not k.startswith("reserved")}


class np_index_arrays(C.Structure):
    _fields_ = [("num_documents_total", C.c_int64), ("doc_begin", C.c_int64), ("num_docs", C.c_int64),
                ("num_centroids", C.c_int64), ("dim", C.c_int32), ("nbits", C.c_int32),
                ("centroids", C.c_void_p), ("bucket_weights", C.c_void_p), ("ivf", C.c_void_p),
                ("ivf_lengths", C.c_void_p), ("doc_lengths", C.c_void_p), ("codes", C.c_void_p),
                ("residuals", C.c_void_p)]


class np_write_opts(C.Structure):
    _fields_ = [("chunk_docs", C.c_int64), ("bucket_cutoffs", C.c_void_p), ("avg_residual", C.c_void_p),
                ("cluster_threshold", C.c_float)]


class np_synth_spec(C.Structure):
    _fields_ = [("num_docs", C.c_int64), ("num_centroids", C.c_int64), ("dim", C.c_int32), ("nbits", C.c_int32),
                ("doc_len_min", C.c_int32), ("doc_len_max", C.c_int32), ("n_topics", C.c_int32),
                ("rand256", C.c_int32), ("seed", C.c_uint64), ("centroids", C.c_void_p),
                ("bucket_weights", C.c_void_p), ("len_table", C.c_void_p), ("len_table_size", C.c_int32)]


# np_all_gather_host_fn: int (*)(void* ctx, const void* send, void* recv, int64_t bytes)
ALL_GATHER_HOST_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)
NP_COMM_DEFERRED_STATUS = 1

NP_ABI_VERSION = 6     # include/nextplaid_hip.h this mirror was written against

EXPORTS = [
    "np_hip_abi_version", "np_hip_struct_size", "np_hip_comm_info",
    "np_hip_device_count", "np_hip_last_error", "np_hip_index_open", "np_hip_index_from_arrays",
    "np_hip_index_synth", "np_hip_index_export", "np_hip_index_ivf_size", "np_hip_index_tune", "np_hip_index_close",
    "np_hip_index_info", "np_hip_index_probe_dir", "np_hip_index_write_dir", "np_hip_search_batch", "np_hip_search_batch_device", "np_hip_search_phase_a",
    "np_hip_search_phase_b", "np_hip_search_end", "np_hip_n_sel", "np_hip_select_cut", "np_hip_merge_topk",
    "np_hip_merge_packed", "np_hip_elig_words", "np_hip_subset_eligible", "np_hip_or_bitmaps",
    "np_hip_comm_unique_id", "np_hip_comm_create", "np_hip_comm_create_hosted", "np_hip_comm_status", "np_hip_comm_destroy",
    "np_hip_search_batch_sharded",
    "np_hip_decompress_documents", "np_hip_encode_tokens", "np_hip_rerank_maxsim", "np_hip_debug_trace",
]

_lib = None


def _preload_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64.  Two HIP runtimes cannot share a
    process: if libnextplaid_hip.so pulls in /opt/rocm's copy first, a later `import torch` loads the bundled one
    next to it and torch's device init fails ("no ROCm-capable device is detected").  So when torch is installed
    its runtime is loaded first (without importing torch); our NEEDED libamdhip64.so.7 then binds to that copy and
    torch reuses it.  Without torch (C++/Rust hosts, plain ctypes users) the system runtime is used."""
    if os.environ.get("NEXTPLAID_HIP_NO_TORCH_RUNTIME"):
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    hip = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(hip):
        try:
            C.CDLL(hip, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """Load libnextplaid_hip.so.  Raises DeviceUnavailableError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise DeviceUnavailableError(f"{path} not built (run python -c 'import __graft_entry__ as g; g.build()')")
    _preload_hip_runtime()
    try:
        L = C.CDLL(path)
    except OSError as e:  # e.g. libamdhip64 missing
        raise DeviceUnavailableError(f"cannot load {path}: {e}") from e
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    # ABI check before any caller-allocated struct crosses the boundary (np_info / np_stats have grown between versions)
    if not hasattr(L, "np_hip_abi_version"):
        raise DeviceUnavailableError(f"{path} predates ABI v6 (no np_hip_abi_version); this mirror needs v{NP_ABI_VERSION}: rebuild it")
    L.np_hip_abi_version.restype = C.c_int
    L.np_hip_struct_size.argtypes = [i32]
    L.np_hip_struct_size.restype = i64
    ver = int(L.np_hip_abi_version())
    if ver != NP_ABI_VERSION:
        raise DeviceUnavailableError(f"{path} speaks ABI v{ver}, this mirror v{NP_ABI_VERSION}: rebuild the library")
    for which, st in ((0, np_info), (1, np_stats), (2, np_search_params), (3, np_open_opts)):
        if int(L.np_hip_struct_size(which)) != C.sizeof(st):
            raise DeviceUnavailableError(f"{path}: sizeof({st.__name__}) is {int(L.np_hip_struct_size(which))} in the library, "
                                         f"{C.sizeof(st)} in this mirror")
    L.np_hip_device_count.restype = C.c_int
    L.np_hip_last_error.restype = C.c_char_p
    L.np_hip_index_open.argtypes = [C.c_char_p, C.POINTER(np_open_opts), C.POINTER(vp)]
    L.np_hip_index_from_arrays.argtypes = [C.POINTER(np_index_arrays), C.POINTER(np_open_opts), C.POINTER(vp)]
    L.np_hip_index_synth.argtypes = [C.POINTER(np_synth_spec), C.POINTER(np_open_opts), C.POINTER(vp)]
    L.np_hip_index_export.argtypes = [vp] * 6
    L.np_hip_index_ivf_size.argtypes = [vp]
    L.np_hip_index_ivf_size.restype = i64
    L.np_hip_index_tune.argtypes = [vp, C.c_char_p, i32]
    L.np_hip_index_close.argtypes = [vp]
    L.np_hip_index_close.restype = None
    L.np_hip_index_info.argtypes = [vp, C.POINTER(np_info)]
    L.np_hip_index_probe_dir.argtypes = [C.c_char_p, C.POINTER(np_info)]
    L.np_hip_index_write_dir.argtypes = [C.c_char_p, C.POINTER(np_index_arrays), C.POINTER(np_write_opts)]
    L.np_hip_search_batch.argtypes = [vp, vp, vp, i32, i32, C.POINTER(np_search_params), vp, i64, vp, vp, vp,
                                      C.POINTER(np_stats)]
    L.np_hip_search_batch_device.argtypes = [vp, vp, vp, vp, i32, i32, C.POINTER(np_search_params), vp, i64,
                                             vp, vp, vp, vp]
    L.np_hip_search_phase_a.argtypes = [vp, vp, vp, vp, i32, i32, C.POINTER(np_search_params), vp, i64, vp, vp, vp,
                                        C.POINTER(vp)]
    L.np_hip_elig_words.argtypes = [vp]
    L.np_hip_elig_words.restype = i64
    L.np_hip_subset_eligible.argtypes = [vp, vp, i64, vp, vp]
    L.np_hip_or_bitmaps.argtypes = [vp, vp, i32, i64, vp, vp]
    L.np_hip_merge_packed.argtypes = [vp, vp, i64, i64, i64, i64, i32, i32, i32, vp, vp, vp, vp]
    L.np_hip_comm_unique_id.argtypes = [vp]
    L.np_hip_comm_create.argtypes = [vp, vp, i32, i32, C.POINTER(vp)]
    L.np_hip_comm_create_hosted.argtypes = [vp, i32, i32, ALL_GATHER_HOST_FN, vp, i32, C.POINTER(vp)]
    L.np_hip_comm_status.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.np_hip_comm_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    L.np_hip_comm_destroy.argtypes = [vp]
    L.np_hip_comm_destroy.restype = None
    L.np_hip_search_batch_sharded.argtypes = [vp, vp, vp, vp, vp, i32, i32, C.POINTER(np_search_params), vp, i64,
                                              vp, vp, vp, vp]
    L.np_hip_search_phase_b.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    L.np_hip_search_end.argtypes = [vp, vp]
    L.np_hip_search_end.restype = None
    L.np_hip_n_sel.argtypes = [C.POINTER(np_search_params)]
    L.np_hip_n_sel.restype = i32
    L.np_hip_select_cut.argtypes = [vp, vp, i32, i32, i32, vp, vp]
    L.np_hip_merge_topk.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp]
    L.np_hip_decompress_documents.argtypes = [vp, vp, i64, vp, i64, vp]
    L.np_hip_encode_tokens.argtypes = [vp, vp, i64, i32, vp, vp, vp]
    L.np_hip_rerank_maxsim.argtypes = [i32, vp, i32, i32, vp, vp, i64, vp, vp]
    L.np_hip_debug_trace.argtypes = [vp, vp, i32, i32, C.POINTER(np_search_params), vp, i64,
                                     vp, i64, vp, vp, vp, i64, vp, vp, vp, i64, vp]
    _lib = L
    return L


def last_error() -> str:
    return lib().np_hip_last_error().decode("utf-8", "replace")


def _check(rc: int, msg: str | None = None):
    if rc:
        msg = last_error() if msg is None else msg
        raise _ERR.get(rc, NextPlaidError)(msg or f"np_status {rc}")


def probe_index_dir(path: str) -> np_info:
    """Host-only parse + validation of an index directory (same checks and errors as MmapIndex.load; no GPU)."""
    info = np_info()
    _check(lib().np_hip_index_probe_dir(os.fsencode(path), C.byref(info)))
    return info


def write_index_dir(path: str, centroids, bucket_weights, doc_lengths, codes, residuals, nbits, ivf=None, ivf_lengths=None,
                    bucket_cutoffs=None, avg_residual=None, cluster_threshold: float = 0.0, chunk_docs: int = 50000):
    """write_index_from_encoded_chunks (index.rs:373-528): host arrays -> an index directory in the crate's on-disk
    format (host only).  Without ivf / ivf_lengths the posting lists are built from the codes (index.rs:479-504)."""
    cen = np.ascontiguousarray(centroids, np.float32)
    if cen.ndim != 2:
        raise ShapeError("centroids must be [K, dim]")
    w = np.ascontiguousarray(bucket_weights, np.float32)
    dl = np.ascontiguousarray(doc_lengths, np.int64)
    cd = np.ascontiguousarray(codes, np.int64)
    rs = np.ascontiguousarray(residuals, np.uint8)
    pd = cen.shape[1] * int(nbits) // 8
    if w.size != (1 << int(nbits)):
        raise CodecError(f"Codec error: bucket_weights has {w.size} entries, nbits={nbits} needs {1 << int(nbits)}")
    if cd.size != int(dl.sum()) or rs.size != cd.size * pd:
        raise ShapeError(f"Shape error: {cd.size} codes / {rs.size} residual bytes for {int(dl.sum())} tokens of {pd} bytes")
    iv = None if ivf is None else np.ascontiguousarray(ivf, np.int64)
    il = None if ivf_lengths is None else np.ascontiguousarray(ivf_lengths, np.int32)
    cut = None if bucket_cutoffs is None else np.ascontiguousarray(bucket_cutoffs, np.float32)
    avg = None if avg_residual is None else np.ascontiguousarray(avg_residual, np.float32)
    if cut is not None and cut.size != (1 << int(nbits)) - 1:
        raise CodecError(f"Codec error: bucket_cutoffs has {cut.size} entries, nbits={nbits} needs {(1 << int(nbits)) - 1}")
    if avg is not None and avg.size != cen.shape[1]:
        raise ShapeError("avg_residual must have dim entries")
    a = np_index_arrays(dl.size, 0, dl.size, cen.shape[0], cen.shape[1], int(nbits), _ptr(cen), _ptr(w),
                        None if iv is None else _ptr(iv), None if il is None else _ptr(il), _ptr(dl), _ptr(cd), _ptr(rs))
    o = np_write_opts(int(chunk_docs), None if cut is None else _ptr(cut), None if avg is None else _ptr(avg),
                      float(cluster_threshold))
    _check(lib().np_hip_index_write_dir(path.encode(), C.byref(a), C.byref(o)))


def rerank_maxsim(query, documents, device: int = 0):
    """/rerank (next-plaid-api handlers/rerank.rs:57-170): MaxSim of one query against caller-supplied document
    embeddings.  Returns (order, scores): document indices by descending score (stable) and the scores in INPUT
    order.  ValueError carries the handler's BadRequest messages."""
    q = np.ascontiguousarray(query, np.float32)
    docs = [np.ascontiguousarray(d, np.float32) for d in documents]
    if q.ndim != 2:
        raise ShapeError(f"Shape error: query has shape {q.shape}")
    for d in docs:
        if d.ndim != 2 or d.shape[1] != q.shape[1]:   # ApiError::DimensionMismatch (rerank.rs:141-146)
            raise ShapeError(f"Shape error: expected dim {q.shape[1]}, got {d.shape}")
    off = np.zeros(len(docs) + 1, np.int64)
    if docs:
        off[1:] = np.cumsum([d.shape[0] for d in docs])
    flat = np.concatenate(docs, 0) if docs and off[-1] > 0 else np.zeros((1, max(q.shape[1], 1)), np.float32)
    scores = np.zeros(max(len(docs), 1), np.float32)
    order = np.zeros(max(len(docs), 1), np.int64)
    _check(lib().np_hip_rerank_maxsim(int(device), _ptr(q), q.shape[0], q.shape[1], _ptr(flat), _ptr(off), len(docs),
                                      _ptr(scores), _ptr(order)))
    return order[: len(docs)], scores[: len(docs)]


def device_count() -> int:
    try:
        return int(lib().np_hip_device_count())
    except DeviceUnavailableError:
        return 0


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# ---- crate mirror ------------------------------------------------------------------------------------

@dataclass
class SearchParameters:
    """search.rs:26-69 (same field names and defaults) + `precision`, the arithmetic of the exact MaxSim stage
    (S1-S5 are always exact f32): 2 (default) = QC-reuse form with split-bf16 MFMA on the residual term, f32-class
    accuracy (max relative score error 5e-7 measured, the same as mode 0); 0 = exact-f32 MFMA on decompressed rows;
    1 = QC-reuse with plain bf16 on the residual term (<= 1e-3 relative); 3 = bf16 MFMA on decompressed rows."""
    batch_size: int = 2000
    n_full_scores: int = 4096
    top_k: int = 10
    n_ivf_probe: int = 8
    centroid_batch_size: int = 100_000
    centroid_score_threshold: float | None = 0.4
    precision: int = 2

    # serde (search.rs:26-48): the four counts are required fields, `centroid_batch_size` and `centroid_score_threshold`
    # carry #[serde(default = ...)] (100 000 / Some(0.4)); an explicit null threshold is None; unknown fields are ignored
    @classmethod
    def from_json(cls, text) -> "SearchParameters":
        import json
        d = json.loads(text) if isinstance(text, (str, bytes)) else dict(text)
        for k in ("batch_size", "n_full_scores", "top_k", "n_ivf_probe"):
            if k not in d:
                raise ValueError(f"missing field `{k}`")
            if isinstance(d[k], bool) or not isinstance(d[k], int) or d[k] < 0:
                raise ValueError(f"invalid type for `{k}`: expected usize")
        t = d.get("centroid_score_threshold", 0.4)
        return cls(batch_size=d["batch_size"], n_full_scores=d["n_full_scores"], top_k=d["top_k"], n_ivf_probe=d["n_ivf_probe"],
                   centroid_batch_size=int(d.get("centroid_batch_size", 100_000)),
                   centroid_score_threshold=None if t is None else float(t), precision=int(d.get("precision", 2)))

    def to_json(self) -> str:
        import json
        return json.dumps(dict(batch_size=self.batch_size, n_full_scores=self.n_full_scores, top_k=self.top_k,
                               n_ivf_probe=self.n_ivf_probe, centroid_batch_size=self.centroid_batch_size,
                               centroid_score_threshold=self.centroid_score_threshold))

    def _c(self) -> np_search_params:
        t = self.centroid_score_threshold
        return np_search_params(self.top_k, self.n_full_scores, self.n_ivf_probe, self.centroid_batch_size,
                                0.0 if t is None else float(t), 0 if t is None else 1, self.precision)


@dataclass
class QueryResult:
    """search.rs:71-80"""
    query_id: int
    passage_ids: np.ndarray  # i64
    scores: np.ndarray       # f32


def _opts(device=0, shard_rank=0, shard_count=1, n_contexts=2, max_batch=64, max_query_tokens=64,
          workspace_bytes=0):
    return np_open_opts(device, shard_rank, shard_count, n_contexts, max_batch, max_query_tokens, workspace_bytes)


class MmapIndex:
    """Device-resident PLAID index; mirror of next_plaid::MmapIndex (index.rs:995-1312)."""

    def __init__(self, handle, path="", open_opts=None):
        self._h = handle
        self.path = path
        self._open_opts = dict(open_opts or {})
        self._info = np_info()
        _check(lib().np_hip_index_info(self._h, C.byref(self._info)))
        self.last_stats: dict | None = None

    # -- constructors ---------------------------------------------------------------------------------
    @classmethod
    def load(cls, index_path: str, **opts) -> "MmapIndex":
        """MmapIndex::load (index.rs:1026).  opts: device, shard_rank, shard_count, n_contexts, max_batch."""
        h = C.c_void_p()
        o = _opts(**opts)
        _check(lib().np_hip_index_open(os.fsencode(index_path), C.byref(o), C.byref(h)))
        return cls(h, index_path, opts)

    def reload(self):
        """MmapIndex::reload (index.rs:1767-1775): refresh the resident index from its directory after the files changed
        (delete / update write new chunk files).  Like the crate -- which releases its maps first -- the old device copy is
        dropped BEFORE the new one is read: two 200 GB copies do not fit one GPU.  Exclusive access, as `&mut self` there;
        a service swaps handles instead.  If the directory no longer loads, the error is raised
        and the handle stays closed."""
        if not self.path or self.path.startswith("<"):
            raise IndexLoadError("Index load failed: reload() needs an index opened from a directory")
        self.close()
        h = C.c_void_p()
        o = _opts(**self._open_opts)
        _check(lib().np_hip_index_open(os.fsencode(self.path), C.byref(o), C.byref(h)))
        self._h = h
        _check(lib().np_hip_index_info(self._h, C.byref(self._info)))
        self.last_stats = None

    @classmethod
    def from_arrays(cls, centroids, bucket_weights, ivf, ivf_lengths, doc_lengths, codes, residuals, nbits,
                    num_documents_total=None, doc_begin=0, **opts) -> "MmapIndex":
        cen = np.ascontiguousarray(centroids, np.float32)
        w = np.ascontiguousarray(bucket_weights, np.float32)
        ivf = np.ascontiguousarray(ivf, np.int64)
        il = np.ascontiguousarray(ivf_lengths, np.int32)
        dl = np.ascontiguousarray(doc_lengths, np.int64)
        cd = np.ascontiguousarray(codes, np.int64)
        rs = np.ascontiguousarray(residuals, np.uint8)
        if cen.ndim != 2:
            raise ShapeError("centroids must be [K, dim]")
        a = np_index_arrays(dl.size if num_documents_total is None else num_documents_total, doc_begin, dl.size,
                            cen.shape[0], cen.shape[1], int(nbits), _ptr(cen), _ptr(w), _ptr(ivf), _ptr(il),
                            _ptr(dl), _ptr(cd), _ptr(rs))
        h = C.c_void_p()
        o = _opts(**opts)
        _check(lib().np_hip_index_from_arrays(C.byref(a), C.byref(o), C.byref(h)))
        return cls(h, "<arrays>")

    @classmethod
    def synth(cls, spec, centroids=None, **opts) -> "MmapIndex":
        """Seeded synthetic corpus generated in HBM (spec: next_plaid_amd.synth.SynthSpec)."""
        from . import synth as S
        cen = np.ascontiguousarray(S.centroids(spec) if centroids is None else centroids, np.float32)
        _, w = S.bucket_tables(spec)
        w = np.ascontiguousarray(w, np.float32)
        lt = None if spec.len_table is None else np.ascontiguousarray(spec.len_table, np.int32)
        s = np_synth_spec(spec.num_docs, spec.num_centroids, spec.dim, spec.nbits, spec.doc_len_min,
                          spec.doc_len_max, spec.n_topics, spec.rand256, spec.seed, _ptr(cen), _ptr(w),
                          None if lt is None else _ptr(lt), 0 if lt is None else int(lt.size))
        h = C.c_void_p()
        o = _opts(**opts)
        _check(lib().np_hip_index_synth(C.byref(s), C.byref(o), C.byref(h)))
        return cls(h, "<synth>")

    def tune(self, name: str, value: int):
        """Kernel-selection knob on a live handle (np_hip_index_tune): sweep tools / variant parity tests."""
        _check(lib().np_hip_index_tune(self._h, name.encode(), int(value)))

    def close(self):
        h, self._h = self._h, None
        if h:
            lib().np_hip_index_close(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- accessors (index.rs:1290-1312) ---------------------------------------------------------------
    def num_documents(self):
        return int(self._info.num_documents)

    def num_embeddings(self):
        return int(self._info.num_embeddings)

    def num_partitions(self):
        return int(self._info.num_partitions)

    def avg_doclen(self):
        return float(self._info.avg_doclen)

    def embedding_dim(self):
        return int(self._info.embedding_dim)

    @property
    def info(self) -> np_info:
        return self._info

    def workspace_bytes(self) -> int:
        """The LIVE per-context scratch budget (np_info.workspace_bytes, ABI v5): the default one shrinks when another
        tenant of the device leaves less room than at open and grows back towards its open value afterwards."""
        live = np_info()
        _check(lib().np_hip_index_info(self._h, C.byref(live)))
        return int(live.workspace_bytes)

    # -- search ------------------------------------------------------------------------------------------
    def _pack(self, queries):
        qs = [np.ascontiguousarray(q, np.float32) for q in queries]
        d = self.embedding_dim()
        for q in qs:
            if q.ndim != 2 or q.shape[1] != d:
                raise ShapeError(f"Shape error: query has shape {q.shape}, index dim is {d}")
        off = np.zeros(len(qs) + 1, np.int32)
        if qs:
            off[1:] = np.cumsum([q.shape[0] for q in qs])
        flat = np.concatenate(qs, 0) if qs else np.zeros((0, d), np.float32)
        return np.ascontiguousarray(flat, np.float32), off

    def search(self, query, params: SearchParameters, subset=None) -> QueryResult:
        """MmapIndex::search (index.rs:1258-1265); query_id is 0 (search.rs:511-515)."""
        r = self.search_batch([query], params, parallel=False, subset=subset)[0]
        r.query_id = 0
        return r

    def search_batch(self, queries, params: SearchParameters, parallel: bool = True, subset=None):
        """MmapIndex::search_batch (index.rs:1279-1287).  `parallel` only selects the reference's
        error policy (search.rs:650-674): the GPU path always runs the batch as one pipeline pass."""
        flat, off = self._pack(queries)
        B = len(queries)
        k = max(int(params.top_k), 0)
        ids = np.zeros(max(B * k, 1), np.int64)
        sc = np.zeros(max(B * k, 1), np.float32)
        cnt = np.zeros(max(B, 1), np.int32)
        p = params._c()
        sub = None if subset is None else np.ascontiguousarray(subset, np.int64)
        st = np_stats()
        rc = lib().np_hip_search_batch(self._h, _ptr(flat), _ptr(off), B, self.embedding_dim(), C.byref(p),
                                       _ptr(sub), -1 if sub is None else sub.size, _ptr(ids), _ptr(sc), _ptr(cnt),
                                       C.byref(st))
        if rc:
            if parallel and rc == 2:  # search.rs:656-660: a failed query yields an empty result
                return [QueryResult(i, np.zeros(0, np.int64), np.zeros(0, np.float32)) for i in range(B)]
            _check(rc)
        self.last_stats = st.as_dict()
        return [QueryResult(i, ids[i * k: i * k + cnt[i]].copy(), sc[i * k: i * k + cnt[i]].copy())
                for i in range(B)]

    # -- adjacent rows ---------------------------------------------------------------------------------------
    def get_document_embeddings(self, doc_id: int) -> np.ndarray:
        """index.rs:1159-1179"""
        embs, lens = self.decompress_documents([doc_id])
        if doc_id < 0 or doc_id >= self.num_documents():
            raise SearchError(f"Search failed: Invalid document ID: {doc_id}")
        return embs

    def decompress_documents(self, doc_ids):
        """index.rs:1197-1245: (embeddings [sum len, dim], lengths)."""
        ids = np.ascontiguousarray(doc_ids, np.int64)
        lens = np.zeros(max(ids.size, 1), np.int64)
        _check(lib().np_hip_decompress_documents(self._h, _ptr(ids), ids.size, None, 0, _ptr(lens)))
        lens = lens[: ids.size]
        total = int(lens.sum())
        out = np.zeros((max(total, 1), self.embedding_dim()), np.float32)
        _check(lib().np_hip_decompress_documents(self._h, _ptr(ids), ids.size, _ptr(out), total, _ptr(lens)))
        return out[:total], lens

    def encode_tokens(self, embeddings, bucket_cutoffs):
        """Index-time encode of a flat [n, dim] batch against this index's codec (codec.rs:297-411,
        index.rs:289-371): (codes i64 [n], packed residuals u8 [n, dim*nbits/8])."""
        x = np.ascontiguousarray(embeddings, np.float32)
        if x.ndim != 2 or x.shape[1] != self.embedding_dim():
            raise ShapeError(f"Shape error: embeddings have shape {x.shape}, index dim is {self.embedding_dim()}")
        nbits = int(self.info.nbits)
        cut = np.ascontiguousarray(bucket_cutoffs, np.float32)
        if cut.size != (1 << nbits) - 1:
            raise CodecError(f"Codec error: bucket_cutoffs has {cut.size} entries, nbits={nbits} needs {(1 << nbits) - 1}")
        n = x.shape[0]
        codes = np.zeros(max(n, 1), np.int64)
        packed = np.zeros((max(n, 1), x.shape[1] * nbits // 8), np.uint8)
        _check(lib().np_hip_encode_tokens(self._h, _ptr(x), n, x.shape[1], _ptr(cut), _ptr(codes), _ptr(packed)))
        return codes[:n], packed[:n]

    def debug_trace(self, query, params: SearchParameters, subset=None) -> dict:
        q = np.ascontiguousarray(query, np.float32)
        if q.ndim != 2 or q.shape[1] != self.embedding_dim():
            raise ShapeError(f"Shape error: query has shape {q.shape}")
        K = self.num_partitions()
        n_loc = int(self._info.shard_doc_end - self._info.shard_doc_begin)
        nsel = max(int(lib().np_hip_n_sel(C.byref(params._c()))), 1)
        cells = np.zeros(max(K, 1), np.int64)
        cand = np.zeros(max(n_loc, 1), np.int64)
        approx = np.zeros(max(n_loc, 1), np.float32)
        sel = np.zeros(nsel, np.int64)
        sel_exact = np.zeros(nsel, np.float32)
        nc, nd, ns = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        p = params._c()
        sub = None if subset is None else np.ascontiguousarray(subset, np.int64)
        _check(lib().np_hip_debug_trace(self._h, _ptr(q), q.shape[0], q.shape[1], C.byref(p), _ptr(sub),
                                        -1 if sub is None else sub.size, _ptr(cells), cells.size, C.byref(nc),
                                        _ptr(cand), _ptr(approx), cand.size, C.byref(nd),
                                        _ptr(sel), _ptr(sel_exact), sel.size, C.byref(ns)))
        return dict(cells=cells[: nc.value].copy(), cand=cand[: nd.value].copy(), approx=approx[: nd.value].copy(),
                    sel=sel[: ns.value].copy(), sel_exact=sel_exact[: ns.value].copy())

    def export(self) -> dict:
        """Shard arrays back on the host in the on-disk dtypes (bench cpu_baseline, generator tests)."""
        n_loc = int(self._info.shard_doc_end - self._info.shard_doc_begin)
        T = int(self._info.shard_embeddings)
        pd = self.embedding_dim() * int(self._info.nbits) // 8
        K = self.num_partitions()
        dl = np.zeros(max(n_loc, 1), np.int64)
        cd = np.zeros(max(T, 1), np.int64)
        rs = np.zeros((max(T, 1), pd), np.uint8)
        ivf = np.zeros(max(int(lib().np_hip_index_ivf_size(self._h)), 1), np.int64)
        il = np.zeros(max(K, 1), np.int32)
        _check(lib().np_hip_index_export(self._h, _ptr(dl), _ptr(cd), _ptr(rs), _ptr(ivf), _ptr(il)))
        return dict(doc_lengths=dl[:n_loc], codes=cd[:T], residuals=rs[:T], ivf=ivf[: int(il[:K].sum())],
                    ivf_lengths=il[:K], nbits=int(self._info.nbits))
