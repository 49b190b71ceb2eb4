"""Seeded synthetic PLAID corpora in the next-plaid on-disk / in-memory layout.

This is the host (numpy) statement of the generator spec; next-plaid_amd/csrc/synth.hip is the
device statement of the SAME integer hash spec (bit-identical codes / residual bytes / doc
lengths), used by bench.py to build BASELINE.json's config-2 index directly in HBM.

Spec (integer-hash form so host and device agree bit for bit):
  mix64 = splitmix64 finaliser; rnd(stream, i) = mix64(mix64(seed + stream) + i)  (u64 wrap)
  doc length  : Lmin + rnd(LEN, doc) % (Lmax - Lmin + 1), or with a quantile table (ragged corpora, e.g. the clipped
                LogNormal of MS MARCO passages): len_table[rnd(LEN, doc) % len(len_table)]
  doc topics  : topic(doc, s) = ((r & 0xffffffff) % K) >> ((r >> 32) & 3),  r = rnd(TOPIC, doc*T + s)
                (mixture of uniforms over [0,K), [0,K/2), [0,K/4), [0,K/8): a skewed, Zipf-like
                popularity without floating point)
  token code  : r = rnd(TOK, doc*65536 + t);  (r & 0xff) < rand256 -> ((r >> 8) & 0xffffffff) % K
                else topic(doc, (r >> 40) % T)
  residual    : byte j of token (doc,t) = byte (j % 8) (little endian) of
                rnd(RES, (doc*65536 + t)*16 + j // 8)   -> uniform bucket ids, i.e. residuals
                distributed like the quantile buckets of N(0, sigma_r^2)
  centroids   : K unit vectors, numpy PCG64(seed) normals, float32 (host-generated, uploaded)
  bucket_weights / cutoffs : N(0, sigma_r^2) quantiles at (i+0.5)/2^nbits and i/2^nbits
                (what index.rs:260-270 computes from a Gaussian residual sample)
Doc ids are GLOBAL, so any doc range (shard) can be generated independently.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass
from statistics import NormalDist

import numpy as np

S_LEN, S_TOPIC, S_TOK, S_RES, S_QRY = 1, 2, 3, 4, 5
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_G = np.uint64(0x9E3779B97F4A7C15)


def mix64(x):
    with np.errstate(over="ignore"):
        z = np.asarray(x, np.uint64) + _G
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def rnd(seed, stream, i):
    with np.errstate(over="ignore"):
        base = mix64(np.uint64(seed) + np.uint64(stream))
        return mix64(base + np.asarray(i, np.uint64))


@dataclass
class SynthSpec:
    num_docs: int
    num_centroids: int
    dim: int = 128
    nbits: int = 4
    doc_len_min: int = 300
    doc_len_max: int = 300
    n_topics: int = 8
    rand256: int = 51            # ~20 % of tokens get a uniformly random code
    sigma_r: float = 0.044       # per-dim residual std  (|r| ~ 0.5 at d=128)
    seed: int = 1236             # 1234 + config number (config 2)
    len_table: object = None     # int32 quantile table of document lengths (see lognormal_len_table); None = uniform

    @property
    def packed_dim(self):
        return self.dim * self.nbits // 8


def lognormal_len_table(mean: float = 73.0, sigma: float = 0.45, max_len: int = 180, min_len: int = 1,
                        size: int = 1024) -> np.ndarray:
    """Quantile table of a clipped LogNormal with the given mean (before clipping): MS MARCO passages under the
    ColBERTv2 tokenizer are ~73 tokens on average, 180 at most."""
    mu = float(np.log(mean)) - 0.5 * sigma * sigma
    nd = NormalDist(0.0, 1.0)
    q = np.array([nd.inv_cdf((i + 0.5) / size) for i in range(size)])
    return np.clip(np.rint(np.exp(mu + sigma * q)), min_len, max_len).astype(np.int32)


def doc_lengths(spec: SynthSpec, d0: int, d1: int) -> np.ndarray:
    docs = np.arange(d0, d1, dtype=np.uint64)
    if spec.len_table is not None:
        tab = np.asarray(spec.len_table, np.int64)
        return tab[(rnd(spec.seed, S_LEN, docs) % np.uint64(tab.size)).astype(np.int64)]
    span = spec.doc_len_max - spec.doc_len_min + 1
    return (spec.doc_len_min + (rnd(spec.seed, S_LEN, docs) % np.uint64(span))).astype(np.int64)


def _topic(spec, docs, s):
    r = rnd(spec.seed, S_TOPIC, docs * np.uint64(spec.n_topics) + s)
    c = (r & np.uint64(0xFFFFFFFF)) % np.uint64(spec.num_centroids)
    return c >> ((r >> np.uint64(32)) & np.uint64(3))


def doc_tokens(spec: SynthSpec, d0: int, d1: int):
    """codes i64[T], residuals u8[T,pd], lengths i64[n] for global docs [d0,d1)."""
    lens = doc_lengths(spec, d0, d1)
    docs = np.repeat(np.arange(d0, d1, dtype=np.uint64), lens)
    offs = np.concatenate([[0], np.cumsum(lens)])
    t = (np.arange(docs.size, dtype=np.int64) - np.repeat(offs[:-1], lens)).astype(np.uint64)
    tok = docs * np.uint64(65536) + t
    r = rnd(spec.seed, S_TOK, tok)
    is_rand = (r & np.uint64(0xFF)) < np.uint64(spec.rand256)
    rand_code = ((r >> np.uint64(8)) & np.uint64(0xFFFFFFFF)) % np.uint64(spec.num_centroids)
    top_code = _topic(spec, docs, (r >> np.uint64(40)) % np.uint64(spec.n_topics))
    codes = np.where(is_rand, rand_code, top_code).astype(np.int64)
    pd = spec.packed_dim
    nw = (pd + 7) // 8
    w = rnd(spec.seed, S_RES, (tok * np.uint64(16))[:, None] + np.arange(nw, dtype=np.uint64)[None, :])
    res = np.ascontiguousarray(w.astype("<u8")).view(np.uint8).reshape(docs.size, nw * 8)[:, :pd]
    return codes, np.ascontiguousarray(res), lens


def centroids(spec: SynthSpec) -> np.ndarray:
    g = np.random.Generator(np.random.PCG64(spec.seed))
    c = g.standard_normal((spec.num_centroids, spec.dim), dtype=np.float32)
    c /= np.maximum(np.linalg.norm(c, axis=1, keepdims=True), 1e-12)
    return c.astype(np.float32)


def bucket_tables(spec: SynthSpec):
    nd, n = NormalDist(0.0, spec.sigma_r), 1 << spec.nbits
    cut = np.array([nd.inv_cdf(i / n) for i in range(1, n)], np.float32)
    wts = np.array([nd.inv_cdf((i + 0.5) / n) for i in range(n)], np.float32)
    return cut, wts


def build_ivf(codes: np.ndarray, lens: np.ndarray, K: int):
    """index.rs:479-499: per centroid the ascending unique (shard-local) doc ids."""
    n = max(int(lens.size), 1)
    doc_of_tok = np.repeat(np.arange(lens.size, dtype=np.int64), lens)
    key = np.unique(codes.astype(np.int64) * n + doc_of_tok)
    return (key % n).astype(np.int64), np.bincount(key // n, minlength=K).astype(np.int32)


def unpack_buckets(packed: np.ndarray, nbits: int) -> np.ndarray:
    bits = np.unpackbits(np.ascontiguousarray(packed, np.uint8), axis=1, bitorder="big")
    bits = bits.reshape(bits.shape[0], -1, nbits).astype(np.int64)
    return (bits << np.arange(nbits, dtype=np.int64)).sum(-1)


def reconstruct(codes, packed, cen, wts, nbits):
    v = cen[codes] + wts[unpack_buckets(packed, nbits)[:, : cen.shape[1]]]
    return (v / np.maximum(np.linalg.norm(v, axis=1, keepdims=True), 1e-12)).astype(np.float32)


def make_queries(spec: SynthSpec, n_queries: int, n_tokens: int = 32, sigma_q: float = 0.5,
                 cen: np.ndarray | None = None, first_query: int = 0):
    """Each query = n_tokens tokens of one (hash-chosen) document, noised and re-normalised.
    Returns (list of f32[n_tokens, d], source doc ids)."""
    cen = centroids(spec) if cen is None else cen
    _, wts = bucket_tables(spec)
    g = np.random.Generator(np.random.PCG64(spec.seed + 7919))
    qs, src = [], []
    for i in range(first_query, first_query + n_queries):
        doc = int(rnd(spec.seed, S_QRY, np.uint64(i)) % np.uint64(spec.num_docs))
        codes, res, lens = doc_tokens(spec, doc, doc + 1)
        if lens[0] == 0:
            v = g.standard_normal((n_tokens, spec.dim), dtype=np.float32)
        else:
            pick = (rnd(spec.seed, S_QRY, np.uint64(1 << 40) + np.uint64(i) * np.uint64(4096)
                        + np.arange(n_tokens, dtype=np.uint64)) % np.uint64(lens[0])).astype(np.int64)
            v = reconstruct(codes[pick], res[pick], cen, wts, spec.nbits)
            # isotropic noise of expected norm sigma_q
            v = v + np.float32(sigma_q / np.sqrt(spec.dim)) * g.standard_normal(v.shape, dtype=np.float32)
        v = v / np.maximum(np.linalg.norm(v, axis=1, keepdims=True), 1e-12)
        qs.append(np.ascontiguousarray(v, np.float32))
        src.append(doc)
    return qs, np.asarray(src, np.int64)


def generate_arrays(spec: SynthSpec, d0: int = 0, d1: int | None = None) -> dict:
    """Everything MmapIndex needs for docs [d0,d1) as host arrays (on-disk dtypes)."""
    d1 = spec.num_docs if d1 is None else d1
    codes, res, lens = doc_tokens(spec, d0, d1)
    ivf, ivf_lengths = build_ivf(codes, lens, spec.num_centroids)
    cut, wts = bucket_tables(spec)
    return dict(nbits=spec.nbits, centroids=centroids(spec), bucket_cutoffs=cut, bucket_weights=wts,
                ivf=ivf, ivf_lengths=ivf_lengths, doc_lengths=lens, codes=codes, residuals=res)


def write_index(path: str, a: dict, chunk_docs: int = 50_000) -> None:
    """Write host arrays as a next-plaid index directory: the file set of
    write_index_from_encoded_chunks (index.rs:373-528); chunks of <= chunk_docs documents
    (IndexConfig.batch_size = 50 000, index.rs:92)."""
    os.makedirs(path, exist_ok=True)
    cen = np.ascontiguousarray(a["centroids"], "<f4")
    K, d = cen.shape
    nbits = int(a["nbits"])
    lens = np.asarray(a["doc_lengths"], np.int64)
    N, T = int(lens.size), int(lens.sum())
    np.save(os.path.join(path, "centroids.npy"), cen)
    np.save(os.path.join(path, "bucket_weights.npy"), np.ascontiguousarray(a["bucket_weights"], "<f4"))
    if a.get("bucket_cutoffs") is not None:
        np.save(os.path.join(path, "bucket_cutoffs.npy"), np.ascontiguousarray(a["bucket_cutoffs"], "<f4"))
    np.save(os.path.join(path, "avg_residual.npy"), np.zeros(d, "<f4"))
    np.save(os.path.join(path, "cluster_threshold.npy"), np.zeros(1, "<f4"))
    np.save(os.path.join(path, "ivf.npy"), np.ascontiguousarray(a["ivf"], "<i8"))
    np.save(os.path.join(path, "ivf_lengths.npy"), np.ascontiguousarray(a["ivf_lengths"], "<i4"))
    offs = np.concatenate([[0], np.cumsum(lens)])
    n_chunks = max(1, -(-N // chunk_docs))
    for i in range(n_chunks):
        a0, a1 = i * chunk_docs, min(N, (i + 1) * chunk_docs)
        t0, t1 = int(offs[a0]), int(offs[a1])
        np.save(os.path.join(path, f"{i}.codes.npy"), np.ascontiguousarray(a["codes"][t0:t1], "<i8"))
        np.save(os.path.join(path, f"{i}.residuals.npy"), np.ascontiguousarray(a["residuals"][t0:t1], "|u1"))
        with open(os.path.join(path, f"doclens.{i}.json"), "w") as f:
            json.dump([int(x) for x in lens[a0:a1]], f)
        with open(os.path.join(path, f"{i}.metadata.json"), "w") as f:
            json.dump(dict(num_documents=a1 - a0, num_embeddings=t1 - t0, embedding_offset=t0), f)
    with open(os.path.join(path, "plan.json"), "w") as f:
        json.dump(dict(nbits=nbits, num_chunks=n_chunks), f)
    with open(os.path.join(path, "metadata.json"), "w") as f:
        json.dump(dict(num_chunks=n_chunks, nbits=nbits, num_partitions=K, num_embeddings=T,
                       avg_doclen=(T / N if N else 0.0), num_documents=N, embedding_dim=d,
                       next_plaid_compatible=True), f, indent=2)
