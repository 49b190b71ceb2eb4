"""ctypes binding for oracle/libplaid_oracle.so (the C restatement of the reference CPU path).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under next-plaid_amd/ may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libplaid_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "plaid_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class _Params(C.Structure):
    _fields_ = [
        ("top_k", C.c_int32),
        ("n_full_scores", C.c_int32),
        ("n_ivf_probe", C.c_int32),
        ("centroid_batch_size", C.c_int32),
        ("centroid_score_threshold", C.c_float),
        ("has_threshold", C.c_int32),
    ]


class _Trace(C.Structure):
    _fields_ = [
        ("n_cells", C.c_int64), ("cells", C.c_void_p),
        ("n_cand", C.c_int64), ("cand", C.c_void_p),
        ("approx", C.c_void_p),
        ("n_sel", C.c_int64), ("sel", C.c_void_p),
        ("sel_exact", C.c_void_p),
        ("n_ivf_ids", C.c_int64),
        ("n_cand_tokens", C.c_int64),
        ("n_exact_tokens", C.c_int64),
        ("used_batched", C.c_int32),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.po_cmp_score_ascending.argtypes = [C.c_float, C.c_float]
        L.po_cmp_score_ascending.restype = C.c_int
        L.po_is_score_better.argtypes = [C.c_float, C.c_float]
        L.po_is_score_better.restype = C.c_int
        L.po_max_score.argtypes = [C.c_float, C.c_float]
        L.po_max_score.restype = C.c_float
        L.po_unrolled_dot.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.po_unrolled_dot.restype = C.c_float
        L.po_simd_max.argtypes = [C.c_void_p, C.c_int64]
        L.po_simd_max.restype = C.c_float
        L.po_maxsim_score.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64]
        L.po_maxsim_score.restype = C.c_float
        L.po_rerank_maxsim.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
        L.po_rerank_maxsim.restype = C.c_int
        L.po_byte_reversed_bits_map.argtypes = [C.c_int, C.c_void_p]
        L.po_bucket_weight_indices_lookup.argtypes = [C.c_int, C.c_void_p]
        L.po_packbits.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        L.po_quantize_residuals.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]
        L.po_compress_into_codes.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
        L.po_encode_tokens.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_int64,
                                       C.c_void_p, C.c_void_p]
        L.po_decompress.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.po_index_create.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_int32] + [C.c_void_p] * 7
        L.po_index_create.restype = C.c_void_p
        L.po_index_destroy.argtypes = [C.c_void_p]
        L.po_search_one.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(_Params), C.c_void_p, C.c_int64,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(_Trace)]
        L.po_search_one.restype = C.c_int
        L.po_search_one_shard.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(_Params), C.c_void_p, C.c_int64,
                                          C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.POINTER(_Trace)]
        L.po_search_one_shard.restype = C.c_int
        L.po_search_many.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(_Params), C.c_int,
                                     C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.po_search_many.restype = C.c_int
        L.po_num_threads.restype = C.c_int
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


@dataclass
class SearchParameters:
    """Mirror of next-plaid SearchParameters (search.rs:26-69)."""
    batch_size: int = 2000
    n_full_scores: int = 4096
    top_k: int = 10
    n_ivf_probe: int = 8
    centroid_batch_size: int = 100_000
    centroid_score_threshold: float | None = 0.4

    def _c(self) -> _Params:
        t = self.centroid_score_threshold
        return _Params(self.top_k, self.n_full_scores, self.n_ivf_probe, self.centroid_batch_size,
                       0.0 if t is None else float(t), 0 if t is None else 1)


@dataclass
class Trace:
    cells: np.ndarray
    cand: np.ndarray
    approx: np.ndarray
    sel: np.ndarray
    sel_exact: np.ndarray
    n_ivf_ids: int
    n_cand_tokens: int
    n_exact_tokens: int
    used_batched: bool


@dataclass
class QueryResult:
    query_id: int
    passage_ids: np.ndarray
    scores: np.ndarray
    trace: Trace | None = field(default=None, repr=False)


# ---- leaf functions -------------------------------------------------------

def cmp_score_ascending(a, b):
    return lib().po_cmp_score_ascending(a, b)


def is_score_better(a, b):
    return bool(lib().po_is_score_better(a, b))


def max_score(a, b):
    return lib().po_max_score(a, b)


def simd_max(x):
    x = _f32(x)
    return lib().po_simd_max(_ptr(x), x.size)


def maxsim_score(q, d):
    q, d = _f32(q), _f32(d)
    assert q.shape[1] == d.shape[1]
    return lib().po_maxsim_score(_ptr(q), q.shape[0], _ptr(d), d.shape[0], q.shape[1])


def rerank_maxsim(q, d):
    q, d = _f32(q), _f32(d)
    out = np.zeros(1, np.float32)
    rc = lib().po_rerank_maxsim(_ptr(q), q.shape[0], _ptr(d), d.shape[0], q.shape[1], _ptr(out))
    if rc:
        raise ValueError("Rerank score contains non-finite value")
    return float(out[0])


def byte_reversed_bits_map(nbits):
    out = np.zeros(256, np.uint8)
    lib().po_byte_reversed_bits_map(nbits, _ptr(out))
    return out


def bucket_weight_indices_lookup(nbits):
    out = np.zeros((256, 8 // nbits), np.int32)
    lib().po_bucket_weight_indices_lookup(nbits, _ptr(out))
    return out


def packbits(bits):
    bits = np.ascontiguousarray(bits, np.uint8)
    out = np.zeros((bits.size + 7) // 8, np.uint8)
    lib().po_packbits(_ptr(bits), bits.size, _ptr(out))
    return out


def quantize_residuals(residuals, nbits, cutoffs):
    r, c = _f32(residuals), _f32(cutoffs)
    n, dim = r.shape
    out = np.zeros((n, dim * nbits // 8), np.uint8)
    lib().po_quantize_residuals(_ptr(r), n, dim, nbits, _ptr(c), c.size, _ptr(out))
    return out


def compress_into_codes(embeddings, centroids):
    """codec.rs:297-345: nearest centroid by dot product, last index among equal maxima."""
    x, c = _f32(embeddings), _f32(centroids)
    codes = np.zeros(x.shape[0], np.int64)
    lib().po_compress_into_codes(_ptr(x), x.shape[0], _ptr(c), c.shape[0], c.shape[1], _ptr(codes))
    return codes


def encode_tokens(embeddings, centroids, nbits, cutoffs):
    """index.rs:289-371 encode_index_chunk for one flat batch of tokens: (codes i64 [n], packed u8 [n, pd])."""
    x, c, cut = _f32(embeddings), _f32(centroids), _f32(cutoffs)
    n, dim = x.shape
    codes = np.zeros(n, np.int64)
    packed = np.zeros((n, dim * nbits // 8), np.uint8)
    lib().po_encode_tokens(_ptr(x), n, _ptr(c), c.shape[0], dim, nbits, _ptr(cut), cut.size, _ptr(codes), _ptr(packed))
    return codes, packed


def decompress(packed, codes, centroids, bucket_weights, nbits):
    packed = np.ascontiguousarray(packed, np.uint8)
    codes = np.ascontiguousarray(codes, np.int64)
    cen, w = _f32(centroids), _f32(bucket_weights)
    n, dim = codes.size, cen.shape[1]
    out = np.zeros((n, dim), np.float32)
    lib().po_decompress(_ptr(packed), _ptr(codes), n, dim, nbits, _ptr(cen), _ptr(w), _ptr(out))
    return out


# ---- index + search -------------------------------------------------------

class OracleIndex:
    """In-memory view of a next-plaid index (index.rs:995-1016) for the C oracle.

    Arrays follow the on-disk dtypes: centroids f32[K,d], bucket_weights f32[2^nbits],
    ivf i64, ivf_lengths i32[K], doc_lengths i64[N], codes i64[T], residuals u8[T,pd].
    """

    def __init__(self, centroids, bucket_weights, ivf, ivf_lengths, doc_lengths, codes, residuals, nbits):
        self.centroids = _f32(centroids)
        self.bucket_weights = _f32(bucket_weights)
        self.ivf = np.ascontiguousarray(ivf, np.int64)
        self.ivf_lengths = np.ascontiguousarray(ivf_lengths, np.int32)
        self.doc_lengths = np.ascontiguousarray(doc_lengths, np.int64)
        self.codes = np.ascontiguousarray(codes, np.int64)
        self.residuals = np.ascontiguousarray(residuals, np.uint8)
        self.nbits = int(nbits)
        self.K, self.d = self.centroids.shape
        self.N = self.doc_lengths.size
        assert self.residuals.ndim == 2 and self.residuals.shape[1] == self.d * self.nbits // 8
        self._h = lib().po_index_create(self.K, self.d, self.N, self.nbits, _ptr(self.centroids),
                                        _ptr(self.bucket_weights), _ptr(self.ivf), _ptr(self.ivf_lengths),
                                        _ptr(self.doc_lengths), _ptr(self.codes), _ptr(self.residuals))
        self.doc_offsets = np.concatenate([[0], np.cumsum(self.doc_lengths)]).astype(np.int64)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.po_index_destroy(h)

    @classmethod
    def load(cls, path):
        from . import npy_index
        a = npy_index.read_index(path)
        return cls(a["centroids"], a["bucket_weights"], a["ivf"], a["ivf_lengths"], a["doc_lengths"],
                   a["codes"], a["residuals"], a["nbits"])

    def get_document_embeddings(self, doc_id):
        s, e = self.doc_offsets[doc_id], self.doc_offsets[doc_id + 1]
        return decompress(self.residuals[s:e], self.codes[s:e], self.centroids, self.bucket_weights, self.nbits)

    def search(self, query, params: SearchParameters, subset=None, trace=False, shard_view=None) -> QueryResult:
        """shard_view = (eligible u8[K], n_total, subset_len_total): this index is ONE document shard and `subset`
        holds its local ids (tests/shard_oracle_backend.py only)."""
        q = _f32(query)
        if q.ndim != 2 or q.shape[1] != self.d:
            raise ValueError(f"Shape error: query {q.shape} vs dim {self.d}")
        p = params._c()
        k = max(params.top_k, 1)
        ids = np.zeros(k, np.int64)
        sc = np.zeros(k, np.float32)
        cnt = C.c_int32(0)
        sub = None if subset is None else np.ascontiguousarray(subset, np.int64)
        tr = None
        bufs = None
        if trace:
            nfs = max(params.n_full_scores, params.top_k, 1)
            bufs = dict(cells=np.zeros(max(self.K, 1), np.int64), cand=np.zeros(max(self.N, 1), np.int64),
                        approx=np.zeros(max(self.N, 1), np.float32), sel=np.zeros(nfs, np.int64),
                        sel_exact=np.zeros(nfs, np.float32))
            tr = _Trace(0, _ptr(bufs["cells"]), 0, _ptr(bufs["cand"]), _ptr(bufs["approx"]), 0,
                        _ptr(bufs["sel"]), _ptr(bufs["sel_exact"]), 0, 0, 0, 0)
        if shard_view is not None:
            el = np.ascontiguousarray(shard_view[0], np.uint8)
            rc = lib().po_search_one_shard(self._h, _ptr(q), q.shape[0], C.byref(p), _ptr(sub),
                                           -1 if sub is None else sub.size, _ptr(el), int(shard_view[1]),
                                           int(shard_view[2]), _ptr(ids), _ptr(sc), C.byref(cnt),
                                           C.byref(tr) if tr is not None else None)
        else:
            rc = lib().po_search_one(self._h, _ptr(q), q.shape[0], C.byref(p), _ptr(sub),
                                     -1 if sub is None else sub.size, _ptr(ids), _ptr(sc), C.byref(cnt),
                                     C.byref(tr) if tr is not None else None)
        if rc:
            raise RuntimeError(f"Search failed: invalid parameters (rc={rc})")
        n = cnt.value
        t = None
        if trace:
            t = Trace(bufs["cells"][: tr.n_cells].copy(), bufs["cand"][: tr.n_cand].copy(),
                      bufs["approx"][: tr.n_cand].copy(), bufs["sel"][: tr.n_sel].copy(),
                      bufs["sel_exact"][: tr.n_sel].copy(), tr.n_ivf_ids, tr.n_cand_tokens,
                      tr.n_exact_tokens, bool(tr.used_batched))
        return QueryResult(0, ids[:n].copy(), sc[:n].copy(), t)

    def search_batch(self, queries, params: SearchParameters, parallel=True, subset=None):
        qs = [_f32(q) for q in queries]
        for q in qs:
            if q.ndim != 2 or q.shape[1] != self.d:
                raise ValueError(f"Shape error: query {q.shape} vs dim {self.d}")
        B = len(qs)
        off = np.zeros(B + 1, np.int32)
        off[1:] = np.cumsum([q.shape[0] for q in qs])
        flat = np.concatenate(qs, 0) if B else np.zeros((0, self.d), np.float32)
        flat = _f32(flat)
        k = max(params.top_k, 1)
        ids = np.zeros((max(B, 1), k), np.int64)
        sc = np.zeros((max(B, 1), k), np.float32)
        cnt = np.zeros(max(B, 1), np.int32)
        p = params._c()
        sub = None if subset is None else np.ascontiguousarray(subset, np.int64)
        # stride between queries in the C call is top_k
        ids_c = np.zeros(max(B, 1) * max(params.top_k, 1), np.int64)
        sc_c = np.zeros(max(B, 1) * max(params.top_k, 1), np.float32)
        rc = lib().po_search_many(self._h, _ptr(flat), _ptr(off), B, C.byref(p), 1 if parallel else 0,
                                  _ptr(sub), -1 if sub is None else sub.size, _ptr(ids_c), _ptr(sc_c), _ptr(cnt))
        if rc:
            raise RuntimeError(f"Search failed (rc={rc})")
        ids = ids_c.reshape(max(B, 1), -1)
        sc = sc_c.reshape(max(B, 1), -1)
        return [QueryResult(i, ids[i, : cnt[i]].copy(), sc[i, : cnt[i]].copy()) for i in range(B)]


def num_threads():
    return lib().po_num_threads()
